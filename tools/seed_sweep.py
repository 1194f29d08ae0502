"""Development: search time against the size of the seeding pass (index option seed_groups_max); in-tree library or HAC_LIBRARY_PATH.
  python tools/seed_sweep.py [rows] [cap,cap,...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from haconvdr_amd.index import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
caps = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1500, 2500, 3500, 8000]
dev = torch.device("cuda", 0)
idx = FlatIPIndex(768)
bench.fill_index(idx, 0, rows, dev, rows // 8)
q = bench.gen_rows(0xBEEF, 1000, dev)
for cap in caps + caps[:1]:
    idx.set_option("seed_groups_max", str(cap))
    for _ in range(2):
        idx.search_tensor(q, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        idx.search_tensor(q, 100)
    torch.cuda.synchronize()
    print(f"rows {rows} seed_groups_max {cap}: search {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms", flush=True)
