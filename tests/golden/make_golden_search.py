#!/usr/bin/env python3
"""Generate tests/golden/search_*.npz by RUNNING THE REFERENCE's own
``search_one_by_one_with_faiss`` (/root/reference/src/test_HAConvDR_qrecc.py:74-162)
in the authoring container, with an injected exact inner-product index.

faiss itself is not installable here, so the object handed to the
reference in place of ``faiss.IndexFlatIP(768)`` is ``NumpyFmafIndex`` below: an
independent numpy implementation (not the C oracle) of the canonical score — the
k-ordered fp32 fma chain — and the canonical order (score desc, row asc).  For the
``grid`` cases every vector entry is a multiple of 1/8 in [-2, 2], so every partial
sum is exactly representable in fp32 and the scores are independent of summation
order: those cases hold for ANY correct IndexFlatIP, real faiss included.

Fixtures are data only: inputs (or the seeds that regenerate them through
haconvdr_amd.synth) and the reference's outputs.  Run:  python tests/golden/make_golden_search.py
"""
import argparse
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import cases  # noqa: E402  (build-owned deterministic input generators)

REF = "/root/reference"


def import_reference_search():
    """Import the reference module with stubs for packages that are absent here."""
    for name in ("faiss", "pytrec_eval", "toml", "IPython"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "IPython":
                m.embed = lambda *a, **k: None
            sys.modules[name] = m
    sys.path[:0] = [REF, os.path.join(REF, "src")]
    import torch
    import models  # noqa: F401  reference src/models.py (transformers swaps its lazy module here)
    sys.modules["transformers"].AdamW = torch.optim.AdamW  # src/utils.py:11 imports it
    import test_HAConvDR_qrecc as ref
    return ref


def fma32(a, b, c):
    """fp32 fma(a, b, c) for fp32 arrays via float64: a*b is exact in float64; the
    sum is rounded to 53 bits and then to 24.  Double rounding can only bite when the
    float64 sum sits exactly on an fp32 rounding midpoint; those entries are redone
    with exact rational arithmetic."""
    from fractions import Fraction
    p = a.astype(np.float64) * b.astype(np.float64)
    s = p + c.astype(np.float64)
    r = s.astype(np.float32)
    # midpoint detection: s is halfway between two adjacent fp32 values
    lo = np.nextafter(r, np.float32(-np.inf)).astype(np.float64)
    hi = np.nextafter(r, np.float32(np.inf)).astype(np.float64)
    r64 = r.astype(np.float64)
    mid = (s == (r64 + lo) / 2) | (s == (r64 + hi) / 2)
    if mid.any():
        for idx in zip(*np.nonzero(mid)):
            exact = Fraction(float(a[idx])) * Fraction(float(b[idx])) + Fraction(float(c[idx]))
            cands = [float(lo[idx]), float(r64[idx]), float(hi[idx])]
            best = min(cands, key=lambda v: (abs(Fraction(v) - exact), int(np.float32(v).view(np.uint32)) & 1))
            r[idx] = np.float32(best)
    return r


class NumpyFmafIndex:
    """Stand-in for faiss.IndexFlatIP(768): add / search / reset (call sites :98,:102,:122)."""

    def __init__(self, d):
        self.d = d
        self.x = np.zeros((0, d), np.float32)

    def add(self, x):
        assert x.dtype == np.float32 and x.shape[1] == self.d
        self.x = np.concatenate([self.x, x], 0)

    def reset(self):
        self.x = np.zeros((0, self.d), np.float32)

    def search(self, q, k):
        nq, n = q.shape[0], self.x.shape[0]
        acc = np.zeros((nq, n), np.float32)
        for kk in range(self.d):
            acc = fma32(np.broadcast_to(self.x[None, :, kk], (nq, n)), np.broadcast_to(q[:, None, kk], (nq, n)), acc)
        D = np.full((nq, k), -np.finfo(np.float32).max, np.float32)
        I = np.full((nq, k), -1, np.int64)
        for i in range(nq):
            order = np.lexsort((np.arange(n), -acc[i].astype(np.float64)))[:k]
            D[i, :len(order)] = acc[i, order]
            I[i, :len(order)] = order
        return D, I


CASES = [
    # name, kind, n_total, nq, topN, nblocks, passage_block_num
    ("gauss_1blk_k10", "gauss", 600, 5, 10, 1, 1),
    ("gauss_2blk_k100", "gauss", 700, 4, 100, 2, 2),
    ("gauss_4blk_k10", "gauss", 1000, 6, 10, 4, 4),
    ("gauss_8blk_k100", "gauss", 1600, 3, 100, 8, 8),
    ("gauss_missing_blocks", "gauss", 500, 3, 10, 2, 5),   # passage_block_num > files: bare except/break (:94-95)
    ("grid_1blk_k10", "grid", 900, 7, 10, 1, 1),
    ("grid_3blk_k100", "grid", 1200, 5, 100, 3, 3),
    ("dup_rows_across_blocks", "dup", 400, 4, 10, 2, 2),   # identical rows in two blocks: tie rule (:138)
    ("short_block_k100", "gauss", 150, 3, 100, 2, 2),      # blocks shorter than topN -> -1 / -FLT_MAX padding
    ("tiny_total_k100", "gauss", 60, 2, 100, 2, 2),        # fewer than topN rows in total: pads reach the output
]


def make_case(ref, name, kind, n, nq, topN, nblocks, block_num, d=768):
    seed = int.from_bytes(name.encode()[:4], "little")
    x, q, ids = cases.search_case_inputs(kind, seed, n, nq, d)
    bounds = np.linspace(0, n, nblocks + 1).astype(int)
    with tempfile.TemporaryDirectory() as tmp:
        for b in range(nblocks):
            lo, hi = bounds[b], bounds[b + 1]
            with open(os.path.join(tmp, f"passage_emb_block_{b}.pb"), "wb") as f:
                pickle.dump(x[lo:hi], f, protocol=4)          # gen_doc_embeddings.py:131-132
            with open(os.path.join(tmp, f"passage_embid_block_{b}.pb"), "wb") as f:
                pickle.dump(ids[lo:hi], f, protocol=4)        # gen_doc_embeddings.py:134-135
        args = argparse.Namespace(passage_block_num=block_num)
        mD, mI = ref.search_one_by_one_with_faiss(args, tmp, NumpyFmafIndex(d), q, topN)
    mD = np.asarray(mD)
    mI = np.asarray(mI)
    out = dict(kind=kind, seed=seed, n=n, nq=nq, topN=topN, nblocks=nblocks, bounds=bounds, ids=ids,
               ref_D=mD, ref_I=mI, ref_shape=np.array(mD.shape), ref_D_dtype=str(mD.dtype), ref_I_dtype=str(mI.dtype))
    np.savez_compressed(os.path.join(HERE, f"search_{name}.npz"), **out)
    print(f"{name}: D{mD.shape} {mD.dtype}  I{mI.shape} {mI.dtype}")


def main():
    ref = import_reference_search()
    for case in CASES:
        make_case(ref, *case)


if __name__ == "__main__":
    main()
