#!/usr/bin/env python3
"""A/B of encoder options in ONE process, interleaved rounds (cdna_hip_programming.md 5.4 rule 24): per kernel class ms of a
1000 x 512 forward (the bench's encode leg) for each value of an option.
  python tools/ab_encoder.py g8_split 0 15 [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from haconvdr_amd import synth
    from haconvdr_amd.encoder import ANCEEncoder
    name, values = sys.argv[1], sys.argv[2:]
    rounds = 3
    if values and values[-1].startswith("r="):
        rounds = int(values.pop()[2:])
    enc = ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False))
    tok, _ = synth.token_batch(0x70C, 1000, 512, fixed_len=512)
    ids = torch.from_numpy(tok.astype(np.int64)).cuda()
    mask = torch.ones_like(ids)
    res = {v: [] for v in values}
    for r in range(rounds + 1):
        for v in values:
            enc.set_option(name, v)
            enc.set_profiling(True, classes="all")
            enc(ids, mask)
            torch.cuda.synchronize()
            stack = float(np.sum(enc.profile_drain()))
            per = {c: float(np.sum(enc.profile_drain_class(c))) for c in enc.KERNEL_CLASSES}
            enc.set_profiling(False)
            if r:
                res[v].append((stack, per))
    for v in values:
        st = np.array([x[0] for x in res[v]])
        line = f"{name}={v}: stack min {st.min():.2f} med {np.median(st):.2f} ms |"
        for c in enc.KERNEL_CLASSES:
            a = np.array([x[1][c] for x in res[v]])
            line += f" {c} {a.min():.2f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
