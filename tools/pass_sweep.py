#!/usr/bin/env python3
"""Development: prefilter scan time against the number of passes (threshold refreshes between them) and the seeding size,
1000 queries, by corpus size -- what scan_passes() picks by size is checked against every pinned value.
  python tools/pass_sweep.py rows [rows ...]"""
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
    dev = torch.device("cuda", 0)
    q = bench.gen_rows(0xBEEF, 1000, dev)
    for rows in [int(a) for a in sys.argv[1:]] or [1_000_000, 6_750_000]:
        idx = FlatIPIndex(768)
        bench.fill_index(idx, 0, rows, dev, max(bench.CH, rows // 8))
        for _ in range(3):
            idx.search_tensor(q, 100)
        torch.cuda.synchronize()
        best = {}
        for rnd in range(3):
            for passes in ("auto", "1", "2", "3", "4", "5"):
                idx.set_option("scan_passes", passes)
                idx.search_tensor(q, 100)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    idx.search_tensor(q, 100)
                torch.cuda.synchronize()
                best[passes] = min(best.get(passes, 1e9), (time.perf_counter() - t0) / 5 * 1e3)
                if passes == "auto":
                    auto_plan = idx.last_plan()
        idx.set_option("scan_passes", "auto")
        print(f"{rows} rows, 1000 queries, ms per search: " + "  ".join(f"passes={p}: {best[p]:.3f}" for p in best) + f"   | auto = {auto_plan.split('passes=')[1].split()[0]}", flush=True)
        del idx
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
