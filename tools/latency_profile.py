"""Development: the two latency-bound shapes under rocprofv3 --kernel-trace --stats (per-kernel durations):
  enc : ANCE forward of B x L fully padded queries, plain launches (graph = off), N repetitions
  cfg2: 1000 pre-encoded queries over a 1M x 768 corpus (BASELINE configs[1]), N repetitions
  single: one query over the same corpus
  python tools/latency_profile.py enc 4 512 50 | cfg2 30 | single 200"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from haconvdr_amd import synth


def main():
    what = sys.argv[1]
    if what == "enc":
        from haconvdr_amd.encoder import ANCEEncoder
        B, L, N = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        enc = ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False))
        if len(sys.argv) > 5:
            enc.set_option("graph", sys.argv[5])
        else:
            enc.set_option("graph", "off")
        tok, _ = synth.token_batch(0x5EE, B, L, fixed_len=L)
        ids = torch.from_numpy(tok.astype(np.int64)).cuda()
        mask = torch.ones_like(ids)
        for _ in range(3):
            enc(ids, mask)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            enc(ids, mask)
        torch.cuda.synchronize()
        print(f"enc {B}x{L}: {(time.perf_counter() - t0) / N * 1e3:.4f} ms per forward, {enc.last_plan()}")
    else:
        import bench
        from haconvdr_amd.index import FlatIPIndex
        N = int(sys.argv[2])
        dev = torch.device("cuda", 0)
        idx = FlatIPIndex(768)
        xs = torch.cat([bench.gen_rows(0xC0FFEE + c, bench.CH, dev) for c in range(8)])
        idx.add_tensor(xs)
        q = bench.gen_rows(0xBEEF, 1000, dev)
        if what == "single":        # one query over the same 1M resident rows (the image is there from the third search on)
            q = q[:1].contiguous()
        for _ in range(5):
            idx.search_tensor(q, 100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            idx.search_tensor(q, 100)
        torch.cuda.synchronize()
        print(f"{what}: {(time.perf_counter() - t0) / N * 1e3:.4f} ms per search, {idx.last_plan()}")


if __name__ == "__main__":
    main()
