#!/usr/bin/env python3
"""Development: the three parity figures of tests/parity.py for every encoder fixture through every GEMM family, printed
(no asserts) -- the numbers the per-fixture bounds in the tests are set from.   python tools/parity_survey.py"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from haconvdr_amd import synth  # noqa: E402
from haconvdr_amd.encoder import ANCEEncoder  # noqa: E402
from tests import parity  # noqa: E402
from tests.golden.make_golden_encoder import load_case  # noqa: E402


def main():
    encs = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "encoder_*.npz"))):
        ids, mask, ref, nl, mstd = load_case(path)
        key = (nl, mstd)
        if key not in encs:
            encs[key] = ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, nl, layer_matrix_std=mstd))
        enc = encs[key]
        for gemm in ("classic", "8phase", "auto"):
            enc.set_option("gemm", gemm)
            out = enc(ids, mask)
            m = parity.measure(out, ref)
            neg = parity.embeddings_match(np.roll(out, 1, axis=0), ref)
            print(os.path.basename(path)[8:-4], gemm, enc.last_plan().split()[0],
                  json.dumps({k: (round(v, 7) if isinstance(v, float) else v) for k, v in m.items()}),
                  "match", parity.embeddings_match(out, ref), "rolled-match", neg, flush=True)
        enc.set_option("gemm", "auto")


if __name__ == "__main__":
    main()
