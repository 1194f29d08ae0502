"""Soak of the multi-pass scan at sizes where the automatic policy cuts it (>= 1.6M rows): prefilter vs exact kernels, bit for bit."""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from haconvdr_amd.index import FlatIPIndex
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
t0 = time.time(); n_cases = 0; multi = 0
while time.time() - t0 < budget:
    n = int(rng.choice([1_600_000, 2_000_003, 3_300_000, 4_000_000, 8_500_000]))
    nq = int(rng.choice([130, 257, 1000, 1300]))
    k = int(rng.choice([10, 100, 200]))
    kind = rng.choice(["gauss", "dups", "scaled", "sorted"])
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    idxs = {s: FlatIPIndex(768) for s in ("1", "0")}
    for s in idxs: idxs[s].set_option("split", s)
    done = 0
    while done < n:
        m = min(500_000, n - done)
        x = torch.randn((m, 768), generator=g, device="cuda")
        if kind == "dups": x[m // 2:] = x[: m - m // 2].clone()
        elif kind == "scaled": x *= torch.rand((m, 1), generator=g, device="cuda") * 4
        elif kind == "sorted": x *= (1.0 + (done + torch.arange(m, device="cuda", dtype=torch.float32)[:, None]) / n)   # norms grow along the corpus: every pass raises the bar
        for s in idxs: idxs[s].add_tensor(x)
        done += m
    q = torch.randn((nq, 768), generator=g, device="cuda")
    res = {}
    for s in idxs:
        D, I = idxs[s].search_tensor(q, k); torch.cuda.synchronize(); idxs[s].check_status()
        res[s] = (D.clone(), I.clone(), idxs[s].last_plan())
    same = torch.equal(res["1"][1], res["0"][1]) and torch.equal(res["1"][0], res["0"][0])
    if not same:
        print("MISMATCH", n, nq, k, kind, res["1"][2], flush=True); sys.exit(1)
    multi += "passes=1 " not in res["1"][2] and res["1"][2].startswith("split:")
    n_cases += 1
    print(f"{n_cases} ok ({n} rows, {nq} q, k {k}, {kind}): {res['1'][2][:110]}", flush=True)
    del idxs
print(f"soak ok: {n_cases} cases, {multi} in several passes, {time.time() - t0:.0f} s", flush=True)
