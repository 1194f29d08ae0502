#!/usr/bin/env python3
"""Generate tests/golden/trec_case.npz by RUNNING THE REFERENCE's output_test_res
(/root/reference/src/test_HAConvDR_qrecc.py:222-286, print_trec_res stubbed out because pytrec_eval is
absent) on a small synthetic retrieval result with duplicate passage ids and a repeated query id.
Run:  python tests/golden/make_golden_trec.py"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden.make_golden_search import import_reference_search  # noqa: E402


def main():
    ref = import_reference_search()
    ref.print_trec_res = lambda *a, **k: {"stub": True}
    rng = np.random.RandomState(7)
    nq, topN, n_off = 5, 10, 40
    pid_mat = rng.randint(0, n_off, size=(nq, 2 * topN)).astype(np.int64)     # the reference hands 2*topN columns
    score_mat = np.sort(rng.rand(nq, 2 * topN) * 50, axis=1)[:, ::-1].astype(np.float32).astype(np.float64)
    offset2pid = [int(x) for x in rng.randint(100, 120, size=n_off)]          # few distinct pids -> de-dup kicks in
    qids = ["q1_1", "q1_2", "q2_1", "q1_2", "q3_1"]                            # "q1_2" twice
    with tempfile.TemporaryDirectory() as tmp:
        tf = os.path.join(tmp, "test.json")
        with open(tf, "w") as f:
            for q in qids:
                f.write(json.dumps({"sample_id": q}) + "\n")
        args = argparse.Namespace(top_k=topN, test_file_path=tf, qrel_output_path=tmp, output_trec_file="run.trec",
                                  trec_gold_qrel_file_path="unused", rel_threshold=1)
        ref.output_test_res(qids, score_mat, pid_mat, offset2pid, args)
        text = open(os.path.join(tmp, "run.trec")).read()
    np.savez_compressed(os.path.join(HERE, "trec_case.npz"), pid_mat=pid_mat, score_mat=score_mat, offset2pid=np.array(offset2pid),
                        qids=np.array(qids), topN=topN, trec_text=np.array(text))
    print(text[:300])


if __name__ == "__main__":
    main()
