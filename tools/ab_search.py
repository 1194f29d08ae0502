#!/usr/bin/env python3
"""Search timing for A/B runs of kernel variants (HAC_LIBRARY_PATH picks the library): the scan of 1000 queries over N rows.
  HAC_LIBRARY_PATH=scratch/v/libX.so python tools/ab_search.py [rows] [reps] [option=value ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from haconvdr_amd.index import FlatIPIndex
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    idx = FlatIPIndex(768)
    opts = [a.split("=", 1) for a in sys.argv[3:]]
    for name, value in opts:
        idx.set_option(name, value)
    bench.fill_index(idx, 0, rows, dev, rows // 8)
    q = bench.gen_rows(0xBEEF, 1000, dev)
    for _ in range(2):
        idx.search_tensor(q, 100)
    torch.cuda.synchronize()
    idx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        idx.search_tensor(q, 100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ms = idx.profile_drain()
    print(f"{os.environ.get('HAC_LIBRARY_PATH', 'in-tree')} {opts}: rows {rows} search {dt * 1e3:.3f} ms, scan kernels min {min(ms):.3f} med {np.median(ms):.3f} ms | {idx.last_plan()}", flush=True)


if __name__ == "__main__":
    main()
