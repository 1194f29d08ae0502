"""Development: is a forward bit-reproducible?  One handle runs every random batch TWICE; any difference is a defect (round 4 found
one this way: LABNOTES rounds 1-4, 2.4, "A wrong bit the soak found").
  python tools/repeat_encoder.py SECONDS LAYERS B[,B...] L[,L...] [option=value ...]     e.g.  20 2 130,100 64,16 gemm=auto"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haconvdr_amd import synth
from haconvdr_amd.encoder import ANCEEncoder
budget = float(sys.argv[1]); layers = int(sys.argv[2]); Bs = [int(x) for x in sys.argv[3].split(",")]; Ls = [int(x) for x in sys.argv[4].split(",")]
opts = [a.split("=", 1) for a in sys.argv[5:]]
sd = synth.ance_state_dict(0xA11CE, layers)
enc = ANCEEncoder.from_state_dict(sd)
enc.set_option("graph", "off")
for n, v in opts: enc.set_option(n, v)
rng = np.random.default_rng(7)
t0 = time.time(); n = 0; bad = 0; nrows = 0
while time.time() - t0 < budget:
    b = int(rng.choice(Bs)); lmax = int(rng.choice(Ls))
    fixed = bool(rng.integers(0, 2))
    ids, lens = synth.token_batch(int(rng.integers(1 << 30)), b, lmax, fixed_len=lmax if fixed else None)
    mask = (np.arange(lmax)[None, :] < lens[:, None]).astype(ids.dtype)
    a = enc(ids, mask); c = enc(ids, mask)
    n += 1
    if not np.array_equal(a, c):
        bad += 1; nrows += int((a != c).any(1).sum())
print(f"layers={layers} B={Bs} L={Ls} {opts}: {n} pairs, {bad} differing ({nrows} rows), {enc.last_plan()}", flush=True)
