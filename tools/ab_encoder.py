#!/usr/bin/env python3
"""Encoder timing for A/B runs of an option: the bench's 1000 x 512 forward with option values interleaved in one process.
  python tools/ab_encoder.py g8_stagger auto off [B] [L]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from haconvdr_amd import synth
    from haconvdr_amd import encoder as E
    name, values = sys.argv[1], sys.argv[2:4]
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
    Lq = int(sys.argv[5]) if len(sys.argv) > 5 else 512
    ids, _ = synth.token_batch(5, B, Lq, fixed_len=Lq)
    ids_t = torch.from_numpy(ids.astype(np.int64)).cuda()
    mask_t = torch.ones_like(ids_t)
    enc = E.ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False))
    ref = None
    best = {v: 1e9 for v in values}
    for rnd in range(4):
        for v in values:
            enc.set_option(name, v)
            for _ in range(2):
                out = enc(ids_t, mask_t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out = enc(ids_t, mask_t)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            best[v] = min(best[v], dt)
            if ref is None:
                ref = out.clone()
            print(f"round {rnd} {name}={v}: {dt * 1e3:.2f} ms  same bits as the first run: {bool(torch.equal(out, ref))}", flush=True)
    print("best: " + ", ".join(f"{name}={v} {best[v] * 1e3:.2f} ms" for v in values), flush=True)


if __name__ == "__main__":
    main()
