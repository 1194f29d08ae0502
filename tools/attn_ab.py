#!/usr/bin/env python3
"""Development A/B of the two streaming attention forms (attn_pipe = auto | off): outputs must be bit-identical, times per class.
  python tools/attn_ab.py [quick]"""
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
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    enc = E.ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False))
    sens = E.ANCEEncoder.from_state_dict(synth.ance_state_dict(0xFACE, 2, rich=True, layer_matrix_std=0.08))
    sd = dict(synth.ance_state_dict(0xFACE, 2))
    for i in range(2):           # peaked: Q and K scaled x 6 (logits x 36): many items leave the window and take the fix-up pass
        for nm in ("query", "key"):
            for part in ("weight", "bias"):
                key = f"roberta.encoder.layer.{i}.attention.self.{nm}.{part}"
                sd[key] = (sd[key] * 6.0).astype(np.float32)
    peaked = E.ANCEEncoder.from_state_dict(sd)
    cases = [("fixed 64x512", 64, 512, 512), ("fixed 1000x512", 1000, 512, 512), ("ragged 300x512", 300, 512, None), ("ragged 700x96", 700, 96, None),
             ("fixed 500x384", 500, 384, 384), ("ragged 400x300", 400, 300, None), ("fixed 40x32", 40, 32, 32), ("ragged 600x256", 600, 256, None)]
    if quick:
        cases = cases[:3]
    ok = True
    for name, B, L, fixed in cases:
        if fixed:
            ids, _ = synth.token_batch(5, B, L, fixed_len=fixed)
            mask = np.ones_like(ids)
        else:
            ids, lens = synth.token_batch(7 + B, B, L, min_len=1)
            mask = (np.arange(L)[None, :] < lens[:, None]).astype(ids.dtype)
        ids_t = torch.from_numpy(ids.astype(np.int64)).cuda()
        mask_t = torch.from_numpy(mask.astype(np.int64)).cuda()
        for which, e in (("12-layer", enc), ("sens 2-layer", sens), ("peaked 2-layer", peaked)):
            outs = {}
            for mode in ("off", "auto"):
                e.set_option("attn_pipe", "all" if mode == "auto" else mode)
                outs[mode] = e(ids_t, mask_t).cpu().numpy()
                plan = e.last_plan()
            same = np.array_equal(outs["off"], outs["auto"])
            finite = np.isfinite(outs["auto"]).all()
            ok = ok and same and finite
            d = np.abs(outs["off"] - outs["auto"]).max()
            print(f"{name:16s} {which:13s} bit-identical={same} finite={finite} max|diff|={d:.3e} plan[{plan}]", flush=True)
    # timings
    B, L = 1000, 512
    ids, _ = synth.token_batch(5, B, L, fixed_len=L)
    ids_t = torch.from_numpy(ids.astype(np.int64)).cuda()
    mask_t = torch.ones_like(ids_t)
    for mode in ("off", "auto", "off", "auto"):
        enc.set_option("attn_pipe", mode)
        for _ in range(2):
            enc(ids_t, mask_t)
        torch.cuda.synchronize()
        enc.set_profiling(True, classes="all")
        best = {}
        for _ in range(4):
            enc(ids_t, mask_t)
            torch.cuda.synchronize()
            stack = float(np.sum(enc.profile_drain()))
            best["stack"] = min(best.get("stack", 1e9), stack)
            for nm in enc.KERNEL_CLASSES:
                ms = enc.profile_drain_class(nm)
                if ms:
                    best[nm] = min(best.get(nm, 1e9), float(np.sum(ms)))
        enc.set_profiling(False)
        print(f"attn_pipe={mode:5s} 1000x512:", " ".join(f"{k}={v:.3f}" for k, v in best.items()), flush=True)
    print("ALL BIT-IDENTICAL" if ok else "MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
