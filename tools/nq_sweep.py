#!/usr/bin/env python3
"""Development: search time against the number of queries per call, exact fp32 kernels (split = 0) against the fp16 prefilter +
exact rescoring (split = 1), same index, same queries, results compared bit for bit -- where should "auto" switch?
  python tools/nq_sweep.py [rows] [k] [nq,nq,...]"""
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
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_750_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dev = torch.device("cuda", 0)
    idx = FlatIPIndex(768)
    bench.fill_index(idx, 0, rows, dev, max(bench.CH, rows // 8))
    qall = bench.gen_rows(0xBEEF, 1024, dev)
    if rows < 1_000_000:
        idx.set_option("rescore_rows", "0")    # (keeps the comparison to the scan paths; the row-major copy is a separate switch)
    print(f"{rows} rows, top-{k}; ms per search (best of 3 rounds of 5), wall clock around search_tensor + synchronize", flush=True)
    nqs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 4, 8, 16, 17, 24, 32, 40, 48, 64, 96, 128, 256]
    for nq in nqs:
        q = qall[:nq].contiguous()
        res, out, plan = {}, {}, {}
        for rnd in range(3):
            for split in ("0", "1", "auto"):
                idx.set_option("split", split)
                for _ in range(2):
                    D, I = idx.search_tensor(q, k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    D, I = idx.search_tensor(q, k)
                torch.cuda.synchronize()
                res[split] = min(res.get(split, 1e9), (time.perf_counter() - t0) / 5 * 1e3)
                out[split] = (D.clone(), I.clone())
                plan[split] = idx.last_plan().split(" ")[0] + " " + idx.last_plan().split(" ")[1]
        same = bool(torch.equal(out["0"][0], out["1"][0]) and torch.equal(out["0"][1], out["1"][1]))
        print(f"nq {nq:4d}: exact {res['0']:8.3f} ms ({plan['0']:28s})  prefilter {res['1']:8.3f} ms  auto {res['auto']:8.3f} ms ({plan['auto']:28s})  "
              f"prefilter / exact {res['1'] / res['0']:.2f}  same bits {same}", flush=True)
    idx.set_option("split", "auto")


if __name__ == "__main__":
    main()
