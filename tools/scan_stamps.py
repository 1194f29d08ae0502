#!/usr/bin/env python3
"""Development: in-kernel s_memtime stamps of the prefilter scan (scan_split.inc, -DSH_STAMP builds made by tools/mkvariant.sh).

  slices     -DSH_STAMP -DSH_ST_ALL -DSH_ST_R=300 -DSH_ST_T=0 : per wave, one slice of round SH_ST_R: start, third MFMA phase done,
             at the hand-over barrier, past it
  epilogue   -DSH_STAMP -DSH_ST_ALL -DSH_ST_EPI -DSH_ST_R=300 -DSH_ST_T=0 : per wave, the round boundary: k-loop done, candidates
             appended, at the barrier, past it, next round's first step started / its MFMAs done / second step started
  candidates -DSH_STAMP ... : candidates left in the lists after one search (pass rate of the thresholds)

  HAC_LIBRARY_PATH=$PWD/scratch/v/libhaconvdr_X.so python tools/scan_stamps.py slices|epilogue|candidates [rows]
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from haconvdr_amd.index import FlatIPIndex
    mode = sys.argv[1] if len(sys.argv) > 1 else "slices"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    dev = torch.device("cuda", 0)
    idx = FlatIPIndex(768)
    bench.fill_index(idx, 0, rows, dev, rows // 8)
    q = bench.gen_rows(0xBEEF, 1000, dev)
    lib = ctypes.CDLL(os.environ["HAC_LIBRARY_PATH"])
    for rep in range(3):
        idx.search_tensor(q, 100)
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 64)()
        assert lib.hac_debug_scan_stamps(out) == 0
        if mode == "candidates":
            c = out[60] // (rep + 1)   # the counter accumulates over launches
            print(f"candidates in the lists after a search: {c} = {c / (rows * 1024.0) * 100:.4f} % of the (row, query) pairs, {c / 1024:.0f} per query; {idx.last_plan()}")
            return
        if mode == "epilogue":
            st = np.array(out[:64], dtype=np.int64).reshape(8, 8)
            b = st[:, 0].min()
            print("rep", rep, " wave: k-loop done | appended | at barrier | past barrier | next round step 0 | its MFMAs done | step 1   (cycles)")
            for w in range(8):
                print(f"   w{w}: " + " ".join(f"{st[w][k] - b:6d}" for k in range(7)))
        else:
            st = np.array(out[:32], dtype=np.int64).reshape(8, 4)
            b = st[:, 0].min()
            print("rep", rep, " wave: slice start | third MFMA phase done | at hand-over barrier | past it   (cycles from the first wave's start)")
            for w in range(8):
                print(f"   w{w}: {st[w][0] - b:5d} {st[w][1] - b:5d} {st[w][2] - b:5d} {st[w][3] - b:5d}")


if __name__ == "__main__":
    main()
