#!/usr/bin/env python3
"""Generate tests/golden/queries/*: synthetic TopiOCQA / QReCC test files and the examples the REFERENCE's datasets
(/root/reference/src/data.py: Retrieval_topiocqa :25-251, Retrieval_qrecc :381-506, padding_seq_to_same_length :8-23)
build from them with the stub tokenizer, for several argument sets (use_PRL on/off, length budgets that force every
truncation branch).  The fixtures are data: the input files and the expected token-id lists / collated batches.
Run:  python tests/golden/make_golden_queries.py"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden.stub_tokenizer import StubTokenizer  # noqa: E402

OUT = os.path.join(HERE, "queries")

ARGSETS = {   # name -> (dataset, args)
    "topiocqa_prl": ("topiocqa", dict(use_PRL=True, max_query_length=32, max_doc_length=384, max_response_length=32, max_concat_length=512)),
    "topiocqa_all_turns": ("topiocqa", dict(use_PRL=False, max_query_length=32, max_doc_length=384, max_response_length=32, max_concat_length=512)),
    "topiocqa_tight": ("topiocqa", dict(use_PRL=True, max_query_length=8, max_doc_length=20, max_response_length=6, max_concat_length=48)),
    "topiocqa_tight_all": ("topiocqa", dict(use_PRL=False, max_query_length=8, max_doc_length=20, max_response_length=6, max_concat_length=40)),
    "qrecc_history": ("qrecc", dict(use_PRL=False, max_query_length=32, max_doc_length=256, max_response_length=64, max_concat_length=256)),
    "qrecc_prl": ("qrecc", dict(use_PRL=True, max_query_length=32, max_doc_length=256, max_response_length=64, max_concat_length=256)),
    "qrecc_tight": ("qrecc", dict(use_PRL=False, max_query_length=6, max_doc_length=16, max_response_length=9, max_concat_length=30)),
    "qrecc_tight_prl": ("qrecc", dict(use_PRL=True, max_query_length=6, max_doc_length=16, max_response_length=9, max_concat_length=24)),
}


def words(rng, n, tag):
    return " ".join(f"{tag}{int(rng.randint(0, 5000))}" for _ in range(n))


def make_files(rng):
    """Conversations of 1..7 turns; record i of a conversation refers back to records i-1, i-2, ... through rel_label
    (one label per earlier turn), exactly as the reference indexes data[i - (len(rel_label) - index)]."""
    topi, qrecc = [], []
    for conv in range(9):
        n_turns = int(rng.randint(1, 8))
        qs = [words(rng, int(rng.randint(3, 14)), "q") for _ in range(n_turns)]
        ans = [words(rng, int(rng.randint(0, 40)), "a") for _ in range(n_turns)]     # an answer may be empty
        psg = [words(rng, int(rng.randint(30, 500)), "p") for _ in range(n_turns)]
        for t in range(n_turns):
            labels = [int(rng.randint(0, 2)) for _ in range(t)]
            hist = []
            for u in range(t):
                hist += [qs[u], ans[u] if ans[u] else "none"]
            topi.append({"sample_id": f"{conv}_{t + 1}", "cur_utt_text": " [SEP] ".join(hist + [qs[t]]) + " ", "last_response": hist[-1] if hist else "",
                         "pos_docs": [psg[t]], "pos_docs_pids": [int(rng.randint(0, 10 ** 6))], "rel_label": labels})
            qrecc.append({"sample_id": f"{conv}_{t + 1}", "ctx_utts_text": hist, "cur_utt_text": qs[t], "cur_response_text": ans[t],
                          "pos_docs_text": [] if (conv == 4 and t == 1) else [psg[t]], "rel_label": labels})
    return topi, qrecc


def main():
    sys.path[:0] = ["/root/reference/src"]
    import data as ref_data                      # the reference (torch, tqdm, json only)
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.RandomState(20240)
    topi, qrecc = make_files(rng)
    files = {"topiocqa": os.path.join(OUT, "topiocqa_test.jsonl"), "qrecc": os.path.join(OUT, "qrecc_test.jsonl")}
    for name, recs in (("topiocqa", topi), ("qrecc", qrecc)):
        with open(files[name], "w", encoding="utf-8") as f:
            for r in recs:
                f.write(json.dumps(r) + "\n")
    golden = {}
    for name, (ds, kw) in ARGSETS.items():
        args = argparse.Namespace(is_train=False, is_PRF=False, PRF_top=3, is_pseudo_prepos=False, hard_neg_type="bm25", **kw)
        cls = ref_data.Retrieval_topiocqa if ds == "topiocqa" else ref_data.Retrieval_qrecc
        dataset = cls(args, StubTokenizer(), files[ds])
        batch = dataset.get_collate_fn(args)([dataset[i] for i in range(min(5, len(dataset)))])
        golden[name] = {"dataset": ds, "args": kw, "examples": dataset.examples,
                        "first_batch": {k: (v if isinstance(v, list) else {"shape": list(v.shape), "dtype": str(v.dtype), "values": v.tolist()})
                                        for k, v in batch.items()}}
        lens = [sum(e[4] if ds == "topiocqa" else e[2]) for e in dataset.examples]
        print(f"{name}: {len(dataset)} examples, concat lengths min {min(lens)} max {max(lens)} (limit {kw['max_concat_length']})")
    pad = [ref_data.padding_seq_to_same_length(list(range(1, n + 1)), 6) for n in (0, 3, 6, 9)]
    golden["padding_seq_to_same_length"] = {"max_pad_length": 6, "inputs": [list(range(1, n + 1)) for n in (0, 3, 6, 9)], "outputs": [list(p) for p in pad]}
    with open(os.path.join(OUT, "queries_golden.json"), "w") as f:
        json.dump(golden, f)
    print("wrote", os.path.join(OUT, "queries_golden.json"), os.path.getsize(os.path.join(OUT, "queries_golden.json")), "bytes")


if __name__ == "__main__":
    main()
