#!/usr/bin/env python3
"""Development: forward time for every value of one encoder option, interleaved rounds, per batch shape.
  python tools/opt_sweep.py NAME v1,v2,... [BxL ...]      e.g.  python tools/opt_sweep.py attn_qs_pin 0,1,2,4,8,16 1x256 4x512"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from haconvdr_amd import synth
    from haconvdr_amd import encoder as E
    name, values = sys.argv[1], sys.argv[2].split(",")
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[3:]] or [(1, 256), (4, 256), (4, 512), (8, 512)]
    enc = E.ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for B, L in shapes:
        ids, _ = synth.token_batch(5, B, L, fixed_len=L)
        ids_t = torch.from_numpy(ids.astype(np.int64)).cuda()
        mask_t = torch.ones_like(ids_t)
        best, outs = {}, {}
        for rnd in range(4):
            for v in values:
                enc.set_option(name, v)
                for _ in range(3):
                    out = enc(ids_t, mask_t)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(60):
                    out = enc(ids_t, mask_t)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best.get(v, 1e9), e0.elapsed_time(e1) / 60)
                outs[v] = out.clone()
        same = all(torch.equal(outs[values[0]], outs[v]) for v in values)
        print(f"{B:3d} x {L:3d}  " + "  ".join(f"{name}={v}: {best[v]:.4f}" for v in values) + f"   same bits: {same}", flush=True)
    enc.set_option(name, values[0])


if __name__ == "__main__":
    main()
