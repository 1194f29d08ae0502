#!/usr/bin/env python3
"""Development: forward time of small batches with the split-K of the two residual GEMMs pinned ("ksplit_pin" = "out/down"),
every combination, interleaved rounds -- what the cost model in run_forward (pick_ksplit) is checked against.
  python tools/ks_sweep.py [BxL ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from haconvdr_amd import synth
    from haconvdr_amd import encoder as E
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(1, 256), (4, 256), (4, 512), (8, 512)]
    enc = E.ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pins = [(a, b) for a in (1, 2, 3, 4) for b in (1, 2, 3, 4, 6, 8, 16)] + [(0, 0)]
    for B, L in shapes:
        ids, _ = synth.token_batch(5, B, L, fixed_len=L)
        ids_t = torch.from_numpy(ids.astype(np.int64)).cuda()
        mask_t = torch.ones_like(ids_t)
        n_rep = 60
        best = {}
        plan = {}
        for rnd in range(3):
            for pin in pins:
                enc.set_option("ksplit_pin", f"{pin[0]}/{pin[1]}")
                for _ in range(3):
                    enc(ids_t, mask_t)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(n_rep):
                    enc(ids_t, mask_t)
                e1.record()
                torch.cuda.synchronize()
                best[pin] = min(best.get(pin, 1e9), e0.elapsed_time(e1) / n_rep)
                plan[pin] = enc.last_plan().split("ksplit=")[1]
        print(f"{B} x {L}: the model's choice {plan[(0, 0)]}: {best[(0, 0)]:.4f} ms")
        print("   out \\ down " + " ".join(f"{b:8d}" for b in (1, 2, 3, 4, 6, 8, 16)))
        for a in (1, 2, 3, 4):
            print(f"   {a:10d} " + " ".join(f"{best[(a, b)]:8.4f}" for b in (1, 2, 3, 4, 6, 8, 16)))
        bp = min((p for p in pins if p != (0, 0)), key=lambda p: best[p])
        print(f"   best pinned: {bp[0]}/{bp[1]} {best[bp]:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
