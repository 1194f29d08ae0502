"""ORACLE — TEST INFRASTRUCTURE ONLY (ctypes front-end of oracle/liboracle.so).

CPU restatement of the reference's search path; see the header of
``oracle/flat_ip_oracle.c`` for the file:line map and the parity statement
("parity vs real faiss: unpinned"; parity vs the reference's own merge code is
pinned by tests/golden/search_*.npz).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package ``haconvdr_amd`` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/liboracle.so with gcc (building the checker is not using it)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.oracle_ip_scores.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int64, ctypes.c_int, f32p]
        L.oracle_ip_scores.restype = None
        L.oracle_flat_ip_search.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, f32p, i64p]
        L.oracle_flat_ip_search.restype = None
        L.oracle_merge_step.argtypes = [f64p, i64p, f64p, i64p, ctypes.c_int64, ctypes.c_int, f64p, i64p]
        L.oracle_merge_step.restype = None
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_set_num_threads.argtypes = [ctypes.c_int]
        L.oracle_set_num_threads.restype = None
        _LIB = L
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def ip_scores(x, q):
    """Canonical scores [nq, n]: k-ordered fp32 fmaf chain."""
    x = np.ascontiguousarray(x, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    out = np.empty((q.shape[0], x.shape[0]), np.float32)
    lib().oracle_ip_scores(_p(x, ctypes.c_float), x.shape[0], _p(q, ctypes.c_float), q.shape[0],
                           x.shape[1], _p(out, ctypes.c_float))
    return out


def flat_ip_search(x, q, k):
    """IndexFlatIP.search restated -> (D float32 [nq,k], I int64 [nq,k])."""
    x = np.ascontiguousarray(x, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    d = q.shape[1]
    x = x.reshape(-1, d)
    D = np.empty((q.shape[0], k), np.float32)
    I = np.empty((q.shape[0], k), np.int64)
    lib().oracle_flat_ip_search(_p(x, ctypes.c_float), x.shape[0], _p(q, ctypes.c_float), q.shape[0],
                                d, k, _p(D, ctypes.c_float), _p(I, ctypes.c_int64))
    return D, I


class OracleIndex:
    """faiss.IndexFlatIP-shaped object over the oracle (add / search / reset / ntotal)."""

    def __init__(self, d=768):
        self.d = d
        self._rows = []

    @property
    def ntotal(self):
        return sum(r.shape[0] for r in self._rows)

    def add(self, x):
        x = np.array(x, np.float32, copy=True)
        assert x.ndim == 2 and x.shape[1] == self.d
        self._rows.append(x)

    def search(self, q, k):
        x = np.concatenate(self._rows, 0) if self._rows else np.zeros((0, self.d), np.float32)
        return flat_ip_search(x, q, k)

    def reset(self):
        self._rows = []


def search_one_by_one(blocks, q, topN, reference_shape=False):
    """The reference's search_one_by_one_with_faiss (src/test_HAConvDR_qrecc.py:74-162) restated.

    blocks: iterable of (emb float32 [n_b,768], ids int64 [n_b]) in block order.
    Returns (merged_D float64 [nq, topN], merged_I int64 [nq, topN]) — the first
    topN columns of the reference's output, which is all its consumers read
    (:238-239); the reference itself returns 2*topN columns when >=2 blocks load:
    reference_shape=True restates that literally (plain Python loops, small cases only).
    """
    if reference_shape:
        return _search_one_by_one_literal(blocks, q, topN)
    index = OracleIndex(q.shape[1])
    mD = mI = None
    for emb, ids in blocks:
        index.add(emb)                                   # :98
        D, I = index.search(q, topN)                     # :102
        cI = np.asarray(ids)[I]                          # :110  (I == -1 indexes the last id, as numpy does there)
        cD = D.astype(np.float64)                        # :111  .tolist() widens to python float
        index.reset()                                    # :122
        if mD is None:
            mD, mI = cD, cI.astype(np.int64)
            continue
        oD = np.empty_like(mD)
        oI = np.empty_like(mI)
        cI = np.ascontiguousarray(cI, np.int64)
        lib().oracle_merge_step(_p(mD, ctypes.c_double), _p(mI, ctypes.c_int64), _p(cD, ctypes.c_double),
                                _p(cI, ctypes.c_int64), q.shape[0], topN, _p(oD, ctypes.c_double),
                                _p(oI, ctypes.c_int64))
        mD, mI = oD, oI
    return mD, mI


def _search_one_by_one_literal(blocks, q, topN):
    """:74-162 line by line: lists of (score, passage) tuples, the two-pointer merge that stops consuming either
    list at topN (:137-149) and therefore leaves 2*topN entries behind from the second block on."""
    index = OracleIndex(q.shape[1])
    merged = None
    for emb, ids in blocks:
        index.add(emb)                                   # :98
        D, I = index.search(q, topN)                     # :102
        cand_ids = np.asarray(ids)[I].tolist()           # :110
        D = D.tolist()                                   # :111
        cand = [list(zip(sl, pl)) for sl, pl in zip(D, cand_ids)]   # :115-120
        index.reset()                                    # :122
        if merged is None:                               # :126-128
            merged = cand
            continue
        prev, merged = merged, []
        for merged_list, cur_list in zip(prev, cand):    # :133-149
            p1 = p2 = 0
            out = []
            while p1 < topN and p2 < topN:
                if merged_list[p1][0] >= cur_list[p2][0]:
                    out.append(merged_list[p1])
                    p1 += 1
                else:
                    out.append(cur_list[p2])
                    p2 += 1
            out += merged_list[p1:topN] + cur_list[p2:topN]
            merged.append(out)
    return (np.array([[c[0] for c in row] for row in merged]),       # :151-159
            np.array([[c[1] for c in row] for row in merged]))


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    """OpenMP team size of later searches (setting OMP_NUM_THREADS after libgomp is up does nothing)."""
    lib().oracle_set_num_threads(int(n))
