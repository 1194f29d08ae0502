#!/usr/bin/env python3
"""A/B of two BUILDS of the library on the encoder forward: child processes (HAC_LIBRARY_PATH picks the library at load),
interleaved A B A B ..., best-of per child.
  python tools/ab_lib.py A.so B.so [rounds] [B] [L]
  python tools/ab_lib.py --child B L        (what each child runs)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(B, L):
    import numpy as np
    import torch
    from haconvdr_amd import synth
    from haconvdr_amd import encoder as E
    ids, _ = synth.token_batch(5, B, L, fixed_len=L)
    ids_t = torch.from_numpy(ids.astype(np.int64)).cuda()
    mask_t = torch.ones_like(ids_t)
    enc = E.ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False))
    for _ in range(3):
        enc(ids_t, mask_t)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        for _ in range(5):
            out = enc(ids_t, mask_t)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    print("ABRES " + json.dumps({"ms": best * 1e3, "sum": float(out.double().sum().item())}), flush=True)


def main():
    if sys.argv[1] == "--child":
        return child(int(sys.argv[2]), int(sys.argv[3]))
    libs = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    B, L = (sys.argv[4], sys.argv[5]) if len(sys.argv) > 5 else ("1000", "512")
    best = {l: 1e9 for l in libs}
    for r in range(rounds):
        for l in libs:
            env = dict(os.environ, HAC_LIBRARY_PATH=os.path.abspath(l))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", B, L], env=env, capture_output=True, text=True)
            line = [x for x in out.stdout.splitlines() if x.startswith("ABRES ")]
            if not line:
                print(l, "FAILED", out.stderr[-400:], flush=True)
                continue
            res = json.loads(line[0][6:])
            best[l] = min(best[l], res["ms"])
            print(f"round {r} {os.path.basename(l)}: {res['ms']:.3f} ms  sum {res['sum']:.6f}", flush=True)
    a, b = libs
    print(f"best: {os.path.basename(a)} {best[a]:.3f} ms, {os.path.basename(b)} {best[b]:.3f} ms, B/A = {best[b] / best[a]:.4f}", flush=True)


if __name__ == "__main__":
    main()
