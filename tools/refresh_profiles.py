#!/usr/bin/env python3
"""Copy the summaries of one profiling campaign (gpurun_out/<tag>_{trace,fetch,write}, bench JSON) into
profiles/r<round>_*: kernel trace/stats, the two PMC passes and the per-launch HBM traffic that bench.py reads.

  python tools/refresh_profiles.py <tag> <bench.json> [round]
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pick(js, name, field, tag=None, grid=None):
    for c in js.get("counters", []):
        if name in c["kernel"] and (tag is None or tag in c["kernel"]) and (grid is None or c["grid_threads_total"] == grid):
            return c.get(field)
    return None


def main():
    tag, bench = sys.argv[1], sys.argv[2]
    rnd = sys.argv[3] if len(sys.argv) > 3 else "01"
    pre = os.path.join(ROOT, "profiles", f"r{rnd}_")
    summ = os.path.join(ROOT, "tools", "summarize_profile.py")
    for kind, out in (("trace", "bench_kernel_trace"), ("fetch", "bench_pmc_fetch"), ("write", "bench_pmc_write")):
        subprocess.run([sys.executable, summ, os.path.join(ROOT, "gpurun_out", f"{tag}_{kind}"), pre + out, "--rows", "1000000", "--nq", "1000"],
                       check=True, stdout=subprocess.DEVNULL)
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_trace", "*", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], pre + "bench_kernel_stats.csv")
    shutil.copy(bench, pre + "bench.json")
    f = json.load(open(pre + "bench_pmc_fetch.json"))
    w = json.load(open(pre + "bench_pmc_write.json"))
    plan = json.load(open(bench))["roofline"]["kernel"]                 # "scanh_kernel<T>"
    full, seed = plan[:-1] + ", false>", plan[:-1] + ", true>"         # the two instantiations a search launches
    rl, rs = pick(f, full, "hbm_read_bytes_per_launch"), pick(f, seed, "hbm_read_bytes_per_launch")
    wl, ws = pick(w, full, "hbm_write_bytes_per_launch"), pick(w, seed, "hbm_write_bytes_per_launch")
    out = {"rows": 1000000, "nq": 1000,
           "kernel": plan + " (both launches of one search: maxima-only seeding pass over the first sixteenth of the corpus, then the full pass)",
           "hbm_read_bytes_per_launch": int(rl + rs), "hbm_write_bytes_per_launch": int(wl + ws), "hbm_bytes_per_launch": int(rl + rs + wl + ws),
           "per_phase": {"seeding_pass": {"read": int(rs), "write": int(ws)}, "full_pass": {"read": int(rl), "write": int(wl)}},
           "rescore_kernel": {"hbm_read_bytes_per_launch": int(pick(f, "rescore_kernel", "hbm_read_bytes_per_launch")),
                              "note": "16-byte gathers from the T64 tiles: one useful piece per 64-B sector; the x2 streaming correction does not apply to gathers, halve this figure"},
           "exact_kernels": {"kernel": "scanq_kernel<NT=2,W=8>", "hbm_read_bytes_per_launch": int(pick(f, "scanq_kernel<2, 8>", "hbm_read_bytes_per_launch"))},
           "hbm_regime": {"kernel": "scan16_kernel", "nq": 16, "hbm_read_bytes_per_launch": int(pick(f, "scan16_kernel", "hbm_read_bytes_per_launch", grid=65536)),
                          "hbm_write_bytes_per_launch": int(pick(w, "scan16_kernel", "hbm_write_bytes_per_launch", grid=65536)), "algorithmic_bytes": 3072068352},
           "source": f"profiles/r{rnd}_bench_pmc_fetch.json (FETCH_SIZE KiB x 1024 x 2, the gfx950 correction for 16-B/lane streaming reads) + "
                     f"profiles/r{rnd}_bench_pmc_write.json (WRITE_SIZE KiB x 1024); separate rocprofv3 --pmc passes of "
                     "`bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encode`"}
    json.dump(out, open(pre + "pmc_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
