#!/usr/bin/env python3
"""Per-class kernel times of one encoder forward (hipEvent pairs inside the library), best of N forwards.
  [HAC_LIBRARY_PATH=...] python tools/enc_classes.py [B] [L] [option=value ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from haconvdr_amd import synth
    from haconvdr_amd import encoder as E
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    ids, _ = synth.token_batch(5, B, L, fixed_len=L)
    ids_t = torch.from_numpy(ids.astype(np.int64)).cuda()
    mask_t = torch.ones_like(ids_t)
    enc = E.ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False))
    for a in sys.argv[3:]:
        n, v = a.split("=", 1)
        enc.set_option(n, v)
    for _ in range(3):
        enc(ids_t, mask_t)
    torch.cuda.synchronize()
    best = {}
    enc.set_profiling(True, classes="all")
    for _ in range(6):
        enc(ids_t, mask_t)
        torch.cuda.synchronize()
        stack = float(np.sum(enc.profile_drain()))
        best["stack"] = min(best.get("stack", 1e9), stack)
        for name in enc.KERNEL_CLASSES:
            ms = enc.profile_drain_class(name)
            if ms:
                best[name] = min(best.get(name, 1e9), float(np.sum(ms)))
    enc.set_profiling(False)
    print(os.path.basename(os.environ.get("HAC_LIBRARY_PATH", "in-tree")), f"{B}x{L}", " ".join(f"{k}={v:.3f}" for k, v in best.items()), enc.last_plan(), flush=True)


if __name__ == "__main__":
    main()
