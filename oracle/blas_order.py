"""TEST INFRASTRUCTURE (tests/, bench.py's cpu_baseline leg): what the canonical score definition costs a user who compares
with a BLAS-order IndexFlatIP.

The reference's index is faiss.IndexFlatIP (src/test_HAConvDR_topiocqa.py:52, searched at :102; faiss-gpu 1.7.2, README.md:12).
faiss is not installable here, and its summation order is the BLAS library's: for nq >= 20 it computes
S = Q . X_blk^T with sgemm over database blocks of 1024 rows and keeps a running top-k.  This package DEFINES the score as the
k-ordered fp32 fma chain (oracle/flat_ip_oracle.c, the arithmetic of the fp32 MFMA).  Both are correctly rounded-ish fp32 dot
products of the same numbers, so they agree up to reassociation:

    |fl_order(q . x) - q . x|  <=  C_BAND * 2^-23 * sum_j |q_j x_j|        for either order (measured: a small fraction of it),

and ids can only trade places where two rows' exact scores are closer than the two bands together.  blas_order_search restates
the faiss shape with numpy (OpenBLAS sgemm, 1024-row blocks, (score desc, row asc) top-k); tie_band_report measures, for a
result (D, I) of this package on the same inputs,
  * how many of the nq x k positions hold a different id,
  * that every pair of rows the two lists order differently, and every row that is in one list but not the other, lies inside a
    tie band (exact scores closer than the two rows' bands together),
  * that both score sets are within the band of the float64 scores.
"""
import numpy as np

C_BAND = 8.0          # the constant of the band above (n = 768 terms: the worst-case bound would be ~768; 8 holds with a wide margin)
EPS = 2.0 ** -23
BLOCK_ROWS = 1024     # faiss: distance_compute_blas_database_bs


def blas_order_search(x, q, k, block_rows=BLOCK_ROWS):
    """(D float32 [nq, k], I int64 [nq, k]): top-k by the sgemm-order scores, order (score desc, row asc)."""
    x = np.ascontiguousarray(x, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    nq, n = q.shape[0], x.shape[0]
    kk = min(k, n)
    best_s = np.full((nq, kk), -np.inf, np.float32)
    best_i = np.full((nq, kk), np.iinfo(np.int64).max, np.int64)
    for b0 in range(0, n, block_rows):
        sblk = q @ x[b0:b0 + block_rows].T
        cs = np.concatenate([best_s, sblk], 1)
        ci = np.concatenate([best_i, np.broadcast_to(np.arange(b0, b0 + sblk.shape[1], dtype=np.int64), sblk.shape)], 1)
        # (score desc, row asc): lexsort per row would be slow; argpartition by score, then an exact sort of the survivors
        sel = np.argpartition(-cs, kk - 1, axis=1)[:, :kk]
        # rows tied with the kk-th score but cut by argpartition: keep the smaller ids -- resolve by a stable two-key sort of a
        # slightly larger candidate set whenever the boundary score repeats
        kth = np.take_along_axis(cs, sel, 1).min(1, keepdims=True)
        ties = (cs == kth).sum(1) > (np.take_along_axis(cs, sel, 1) == kth).sum(1)
        for r in np.nonzero(ties)[0]:
            order = np.lexsort((ci[r], -cs[r]))[:kk]
            sel[r] = order
        best_s, best_i = np.take_along_axis(cs, sel, 1), np.take_along_axis(ci, sel, 1)
    out_d = np.empty((nq, kk), np.float32)
    out_i = np.empty((nq, kk), np.int64)
    for r in range(nq):
        order = np.lexsort((best_i[r], -best_s[r]))
        out_d[r], out_i[r] = best_s[r][order], best_i[r][order]
    return out_d, out_i


def tie_band_report(x, q, k, D, I, blas=None):
    """Compare this package's (D, I) with the BLAS-order search of the same inputs.  Returns a dict of counts and the three
    checks' verdicts; raises nothing (tests assert on the fields)."""
    x = np.asarray(x, np.float32)
    q = np.asarray(q, np.float32)
    nq = q.shape[0]
    bD, bI = blas if blas is not None else blas_order_search(x, q, k)
    kk = bI.shape[1]
    D, I = np.asarray(D)[:, :kk], np.asarray(I)[:, :kk]
    moved = int((bI != I).sum())
    set_diff = 0
    worst_pos, worst_set, worst_err = 0.0, 0.0, 0.0
    q64 = q.astype(np.float64)
    for r in range(nq):
        ids = np.union1d(I[r], bI[r])
        xs = x[ids].astype(np.float64)
        exact = xs @ q64[r]
        band = C_BAND * EPS * (np.abs(xs) @ np.abs(q64[r]))
        at = {int(i): n for n, i in enumerate(ids)}
        gi = np.array([at[int(i)] for i in I[r]])
        bi = np.array([at[int(i)] for i in bI[r]])
        # scores against float64, in units of the band
        worst_err = max(worst_err, float(np.max(np.abs(D[r].astype(np.float64) - exact[gi]) / band[gi])),
                        float(np.max(np.abs(bD[r].astype(np.float64) - exact[bi]) / band[bi])))
        # rows that trade places: every pair the two lists order differently has exact scores within the two bands of each other
        # (a row that jumps ahead of a near-tied one shifts everything between them by a position: positions are counted above,
        # but only INVERSIONS are evidence of anything)
        common, cg, cb = np.intersect1d(I[r], bI[r], return_indices=True)
        if common.size > 1:
            inv = (cg[:, None] < cg[None, :]) & (cb[:, None] > cb[None, :])
            if inv.any():
                n_ = np.array([at[int(i)] for i in common])
                gap = np.abs(exact[n_][:, None] - exact[n_][None, :]) / (band[n_][:, None] + band[n_][None, :])
                worst_pos = max(worst_pos, float(gap[inv].max()))
        # ids in one list only: the list that left a row out did so within the bands of its own k-th row
        only_g = np.setdiff1d(I[r], bI[r])
        only_b = np.setdiff1d(bI[r], I[r])
        set_diff += int(only_g.size)
        for only, kth_idx in ((only_g, bi[-1]), (only_b, gi[-1])):
            for i in only:
                n = at[int(i)]
                worst_set = max(worst_set, float((exact[n] - exact[kth_idx]) / (band[n] + band[kth_idx])))
    per = 1000.0 * 100.0 / float(nq * kk)
    return {"nq": int(nq), "k": int(kk), "rows": int(x.shape[0]), "moved_positions": moved, "moved_positions_per_1000x100": round(moved * per, 2),
            "ids_in_one_list_only": set_diff, "ids_in_one_list_only_per_1000x100": round(set_diff * per, 2),
            "set_recall_at_k": round(1.0 - set_diff / float(nq * kk), 6),
            "max_score_error_over_band": round(worst_err, 4), "max_inversion_gap_over_bands": round(worst_pos, 4), "max_exclusion_gap_over_bands": round(worst_set, 4),
            "band": f"{C_BAND:g} x 2^-23 x sum_j |q_j x_j|",
            "every_difference_inside_a_tie_band": bool(worst_pos <= 1.0 and worst_set <= 1.0), "scores_within_band": bool(worst_err <= 1.0)}
