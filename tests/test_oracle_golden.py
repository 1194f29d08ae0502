"""The CPU oracle against the golden vectors produced by the REFERENCE's own
search_one_by_one_with_faiss (tests/golden/make_golden_search.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from tests.golden import cases

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "search_*.npz")))


def load_case(path):
    g = np.load(path)
    x, q, ids = cases.search_case_inputs(str(g["kind"]), int(g["seed"]), int(g["n"]), int(g["nq"]))
    assert np.array_equal(ids, g["ids"])
    bounds = g["bounds"]
    blocks = [(x[bounds[b]:bounds[b + 1]], ids[bounds[b]:bounds[b + 1]]) for b in range(int(g["nblocks"]))]
    return g, x, q, ids, blocks


def test_golden_present():
    assert len(GOLD) >= 9


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[7:-4] for p in GOLD])
def test_oracle_matches_reference_merge(path, oracle):
    g, x, q, ids, blocks = load_case(path)
    topN = int(g["topN"])
    mD, mI = oracle.search_one_by_one(blocks, q, topN)
    # reference returns float64 / int64 (SURVEY §3.1 [probed]); first topN columns are the contract
    assert str(g["ref_D_dtype"]) == "float64" and str(g["ref_I_dtype"]) == "int64"
    assert mD.dtype == np.float64 and mI.dtype == np.int64
    assert g["ref_D"].shape[1] == (topN if int(g["nblocks"]) == 1 else 2 * topN)
    np.testing.assert_array_equal(mI, g["ref_I"][:, :topN])
    np.testing.assert_array_equal(mD, g["ref_D"][:, :topN])       # bit-exact scores


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[7:-4] for p in GOLD])
def test_oracle_reference_shape_equals_the_reference_output_in_full(path, oracle):
    """reference_shape=True: every column of the reference's (nq, 2*topN) matrices (test_HAConvDR_qrecc.py:144-162)."""
    g, x, q, ids, blocks = load_case(path)
    mD, mI = oracle.search_one_by_one(blocks, q, int(g["topN"]), reference_shape=True)
    assert mD.shape == tuple(g["ref_shape"]) and mD.dtype == np.float64 and mI.dtype == np.int64
    np.testing.assert_array_equal(mI, g["ref_I"])
    np.testing.assert_array_equal(mD, g["ref_D"])


def test_scalar_and_blocked_scores_agree(oracle):
    x, q, _ = cases.search_case_inputs("gauss", 77, 300, 11)
    s = oracle.ip_scores(x, q)
    D, I = oracle.flat_ip_search(x, q, 300)
    for i in range(q.shape[0]):
        np.testing.assert_array_equal(D[i], s[i, I[i]])
        order = np.lexsort((np.arange(300), -s[i].astype(np.float64)))
        np.testing.assert_array_equal(I[i], order)


def test_tie_order_row_ascending(oracle):
    x = np.zeros((50, 768), np.float32)
    x[:, 0] = 1.0
    x[10:20, 1] = 1.0
    q = np.zeros((1, 768), np.float32)
    q[0, 0] = 1.0
    q[0, 1] = 0.5
    D, I = oracle.flat_ip_search(x, q, 15)
    assert list(I[0]) == list(range(10, 20)) + [0, 1, 2, 3, 4]
    assert np.all(D[0, :10] == 1.5) and np.all(D[0, 10:] == 1.0)


def test_short_and_empty(oracle):
    x, q, _ = cases.search_case_inputs("gauss", 5, 7, 2)
    D, I = oracle.flat_ip_search(x, q, 10)
    assert np.all(I[:, 7:] == -1) and np.all(D[:, 7:] == -np.finfo(np.float32).max)
    D, I = oracle.flat_ip_search(x[:0], q, 3)
    assert np.all(I == -1)
