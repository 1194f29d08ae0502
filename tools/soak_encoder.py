"""Development: soak of the encoder -- one handle encodes a stream of random batches (sizes, lengths, both GEMM families); every
batch is encoded again by a FRESH handle with the same weights: the two must agree bit for bit (nothing a call leaves behind in
the workspaces may reach the next one).
  python tools/soak_encoder.py [seed] [seconds] [layers] [option=value ...]   (options go to BOTH handles)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from haconvdr_amd import synth
    from haconvdr_amd.encoder import ANCEEncoder
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    layers = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rng = np.random.default_rng(seed)
    sd = synth.ance_state_dict(0xA11CE, layers)
    opts = [a.split("=", 1) for a in sys.argv[4:]]
    enc = ANCEEncoder.from_state_dict(sd)
    for name, value in opts:
        enc.set_option(name, value)
    t0 = time.time()
    n = 0
    while time.time() - t0 < budget:
        b = int(rng.choice([1, 3, 8, 40, 130, 300, 700]))
        lmax = int(rng.choice([16, 64, 200, 384, 512]))
        fixed = bool(rng.integers(0, 2))
        ids, lens = synth.token_batch(int(rng.integers(1 << 30)), b, lmax, fixed_len=lmax if fixed else None)
        mask = (np.arange(lmax)[None, :] < lens[:, None]).astype(ids.dtype)
        out = enc(ids, mask)
        plan = enc.last_plan()
        fresh = ANCEEncoder.from_state_dict(sd)
        for name, value in opts:
            fresh.set_option(name, value)
        ref = fresh(ids, mask)
        del fresh
        if not np.array_equal(out, ref):
            d = np.abs(out - ref).max()
            print(f"MISMATCH batch {b} x {lmax} fixed={fixed}: max |diff| {d}; plan {plan}", flush=True)
            sys.exit(1)
        n += 1
        if n % 10 == 0:
            print(f"{n} batches ok, {time.time() - t0:.0f} s (last: {b} x {lmax}, {plan})", flush=True)
    print(f"soak ok: {n} batches in {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
