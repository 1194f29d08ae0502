#!/usr/bin/env python3
"""Development: differential soak of the two streaming attention forms -- random batches (sizes, padded lengths, ragged or full
lengths) through one handle with attn_pipe = all (the woven kernel for every whole item, both length classes, fix-up pass behind it)
and attn_pipe = off (the one-block kernel everywhere): the embeddings must be equal bit for bit.  Three weight sets: soft attention,
content-sensitive (logit sigma ~5), peaked (logits x 64: most items take the fix-up pass).
  python tools/soak_attention.py [seed] [seconds]"""
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
    rng = np.random.default_rng(seed)
    sd_peaked = dict(synth.ance_state_dict(0xFACE, 2))
    for i in range(2):
        for nm in ("query", "key"):
            for part in ("weight", "bias"):
                key = f"roberta.encoder.layer.{i}.attention.self.{nm}.{part}"
                sd_peaked[key] = (sd_peaked[key] * 8.0).astype(np.float32)
    encs = {"soft": ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 2)),
            "sens": ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 3, layer_matrix_std=0.08)),
            "peaked": ANCEEncoder.from_state_dict(sd_peaked)}
    t0, n, redo_seen = time.time(), 0, 0
    while time.time() - t0 < budget:
        name = ("soft", "sens", "peaked")[int(rng.integers(0, 3))]
        enc = encs[name]
        b = int(rng.choice([11, 24, 40, 130, 300, 700]))
        lmax = int(rng.choice([32, 64, 96, 160, 256, 288, 320, 384, 448, 512]))
        fixed = bool(rng.integers(0, 3) == 0)
        ids, lens = synth.token_batch(int(rng.integers(1 << 30)), b, lmax, min_len=1, fixed_len=lmax if fixed else None)
        mask = (np.arange(lmax)[None, :] < lens[:, None]).astype(ids.dtype)
        outs = {}
        for mode in ("off", "all"):
            enc.set_option("attn_pipe", mode)
            outs[mode] = enc(ids, mask)
            if mode == "all":
                woven = "attn_form=woven" in enc.last_plan()
                redo_seen += woven and enc.attention_redo() > 0
        if not np.array_equal(outs["off"], outs["all"]):
            bad = np.flatnonzero((outs["off"] != outs["all"]).any(1))
            print(f"MISMATCH: {name} {b} x {lmax} fixed={fixed}: rows {bad[:10].tolist()} lens {lens[bad[:10]].tolist()} ({enc.last_plan()})", flush=True)
            return 1
        if not np.isfinite(outs["all"]).all():
            print(f"NON-FINITE: {name} {b} x {lmax}", flush=True)
            return 1
        n += 1
        if n % 50 == 0:
            print(f"{n} batches ok, {time.time() - t0:.0f} s (last: {name} {b} x {lmax}, {enc.last_plan()})", flush=True)
    print(f"soak ok: {n} batches bit-equal between the woven and the one-block attention kernels ({redo_seen} of them with items through the fix-up pass) in {time.time() - t0:.0f} s", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
