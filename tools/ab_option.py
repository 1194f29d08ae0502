#!/usr/bin/env python3
"""A/B of ONE encoder option inside one library: forward time per call (hipEvents around N calls, graph replay as shipped)
with the option at value A and at value B, interleaved R rounds, per batch shape -- and whether the two give the same bits.
  python tools/ab_option.py NAME A B [BxL ...]        e.g.  python tools/ab_option.py xcd_deal auto off 1x256 4x512"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from haconvdr_amd import synth
    from haconvdr_amd import encoder as E
    name, va, vb = sys.argv[1:4]
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[4:]] or [(1, 256), (4, 256), (4, 512), (16, 512), (64, 512)]
    enc = E.ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for B, L in shapes:
        ids, _ = synth.token_batch(5, B, L, fixed_len=L)
        ids_t = torch.from_numpy(ids.astype(np.int64)).cuda()
        mask_t = torch.ones_like(ids_t)
        n_rep = max(5, min(200, int(400 / (B * L / 2048 + 1))))
        res, outs = {va: [], vb: []}, {}
        for rnd in range(5):
            for v in (va, vb):
                enc.set_option(name, v)
                for _ in range(3):
                    out = enc(ids_t, mask_t)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(n_rep):
                    out = enc(ids_t, mask_t)
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / n_rep)
                outs[v] = out.clone()
        same = bool(torch.equal(outs[va], outs[vb]))
        a, b = min(res[va]), min(res[vb])
        print(f"{B:4d} x {L:3d}  {name}={va}: {a:8.4f} ms   {name}={vb}: {b:8.4f} ms   {100 * (a / b - 1):+6.2f} %   same bits: {same}   "
              f"rounds {va}: {' '.join(f'{x:.4f}' for x in res[va])} | {vb}: {' '.join(f'{x:.4f}' for x in res[vb])}   plan: {enc.last_plan()}", flush=True)


if __name__ == "__main__":
    main()
