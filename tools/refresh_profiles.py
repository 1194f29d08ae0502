#!/usr/bin/env python3
"""Turn one profiling campaign (tools/profile_round.sh <tag>, outputs under gpurun_out/<tag>_{trace,fetch,write,sq}) into
the tracked summaries under profiles/:

  python tools/refresh_profiles.py <tag> [--raw DIR] [--out DIR] [--bench bench.json]

(on the GPU box profile_round.sh runs it with --raw /tmp/hac_prof --out gpurun_out/<tag>_profiles; the files that come
back are then copied into profiles/)

  profiles/<tag>_bench.json                 the bench line of an un-profiled run (if given)
  profiles/<tag>_kernel_trace.{json,md}     per (kernel, grid): calls, avg / median / min / max duration, VGPR / LDS / scratch
  profiles/<tag>_kernel_stats.csv           rocprofv3's own --stats table
  profiles/<tag>_pmc_fetch|write.{json,md}  FETCH_SIZE / WRITE_SIZE per launch (KiB; read bytes = KiB x 1024 x 2 on gfx950 for
                                            16-B/lane streaming reads, write bytes = KiB x 1024: MI355X_MICROARCH.md, HBM)
  profiles/<tag>_sq.{json,md}               SQ_* / GRBM per launch + derived: matrix-pipe busy share, wave-time split, clock
  profiles/<tag>_pmc_traffic.json           what bench.py reads back: HBM bytes per launch of the dominant kernels, MFMA busy;
                                            stamped with the git HEAD (HAC_GIT_HEAD on the GPU box, which has no .git) and the
                                            sha256 of the four kernel sources -- bench.py withholds the numbers when they differ
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 256 * 4


def rows_of(js, needle):
    return [c for c in js.get("counters", []) if needle in c["kernel"]]


def per_search_counter(raw_dir, counter, kernel_needle="scanh_kernel<1, false, 16>", end_needle="rescore_kernel"):
    """Sum of `counter` over the launches of the prefilter's main scan, search by search (a search's scan is up to three launches over
    consecutive row ranges; the rescore_kernel launch behind them closes the search).  Read from the raw counter CSVs in dispatch order."""
    rows = []
    for c in glob.glob(os.path.join(raw_dir, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(c)):
            if r["Counter_Name"] == counter or end_needle in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])))
    rows.sort()
    per, cur, launches, seen_end = [], 0.0, 0, set()
    for disp, kname, cname, val in rows:
        if end_needle in kname:
            if disp not in seen_end and launches:
                per.append({"value": cur, "launches": launches})
                cur, launches = 0.0, 0
            seen_end.add(disp)
        elif kernel_needle in kname and cname == counter:
            cur += val
            launches += 1
    return per


def main():
    tag = sys.argv[1]
    opts = dict(zip(sys.argv[2::2], sys.argv[3::2]))
    raw = opts.get("--raw", os.path.join(ROOT, "gpurun_out"))
    outdir = opts.get("--out", os.path.join(ROOT, "profiles"))
    bench = opts.get("--bench")
    os.makedirs(outdir, exist_ok=True)
    pre = os.path.join(outdir, f"{tag}_")
    summ = os.path.join(ROOT, "tools", "summarize_profile.py")
    for kind, out in (("trace", "kernel_trace"), ("fetch", "pmc_fetch"), ("write", "pmc_write"), ("sq", "sq")):
        src = os.path.join(raw, f"{tag}_{kind}")
        if os.path.isdir(src):
            subprocess.run([sys.executable, summ, src, pre + out], check=True, stdout=subprocess.DEVNULL)
    stats = glob.glob(os.path.join(raw, f"{tag}_trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], pre + "kernel_stats.csv")
    if bench:
        shutil.copy(bench, pre + "bench.json")
    sys.path.insert(0, ROOT)
    import bench
    head = os.environ.get("HAC_GIT_HEAD")
    if not head:
        try:
            head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        except Exception:
            head = "unknown (no .git on the GPU box: pass HAC_GIT_HEAD)"
    out = {"git_head": head, "kernel_sources_sha256": bench.kernel_sources_sha256(), "kernel_sources": list(bench.KERNEL_SOURCES),
           "source": f"profiles/{tag}_pmc_fetch.json, {tag}_pmc_write.json, {tag}_sq.json: separate rocprofv3 --pmc passes of "
                     "`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` (tools/profile_round.sh); bench.py quotes these "
                     "numbers only while the digest of the kernel sources of the running tree equals kernel_sources_sha256"}

    def load(name):
        p = pre + name + ".json"
        return json.load(open(p)) if os.path.exists(p) else {}
    f, w, sq = load("pmc_fetch"), load("pmc_write"), load("sq")

    def traffic(needle, long_runs=False):
        fr = [c for c in rows_of(f, needle) if (not long_runs or "short" not in c["kernel"])]
        wr = [c for c in rows_of(w, needle) if (not long_runs or "short" not in c["kernel"])]
        if not fr or not wr:
            return None
        # the instantiation with the largest traffic (the main pass of a kernel that also has a small seeding pass)
        r = max(c["hbm_read_bytes_per_launch"] for c in fr)
        ww = max(c["hbm_write_bytes_per_launch"] for c in wr)
        return {"read": int(r), "write": int(ww), "total": int(r + ww)}
    t = traffic("gemm8_kernel<3,")
    if t:
        out["ffn_up_hbm_bytes_per_launch"] = t["total"]
        out["ffn_up"] = t
    # the search's main scan: ALL its launches of a search together (round 4: up to three passes over consecutive row ranges), search
    # by search -- and the check VERDICT r3 item 2a asks for: no search of the pass may fetch more than 1.15 x the fp16 image
    fs = per_search_counter(os.path.join(raw, f"{tag}_fetch"), "FETCH_SIZE")
    ws = per_search_counter(os.path.join(raw, f"{tag}_write"), "WRITE_SIZE")
    image = float(bench.CFG3_ROWS) * bench.D_EMB * 2
    if fs and ws:
        reads = [x["value"] * 1024 * 2 for x in fs]
        writes = [x["value"] * 1024 for x in ws]
        t = {"read": int(sum(reads) / len(reads)), "write": int(sum(writes) / len(writes))}
        t["total"] = t["read"] + t["write"]
        out["search_hbm_bytes_per_launch"] = t["total"]
        out["search"] = dict(t, per="search (all main-pass launches of scanh_kernel<1, false, 16>)", launches_per_search=fs[0]["launches"],
                             read_per_search=[int(x) for x in reads], fp16_image_bytes=int(image),
                             read_over_image=[round(x / image, 4) for x in reads])
        if max(reads) > 1.15 * image:
            json.dump(out, open(pre + "pmc_traffic.json", "w"), indent=1)
            sys.exit(f"scanh_kernel<1, false, 16> fetched {max(reads) / image:.3f} x the fp16 image in one of {len(reads)} searches (limit 1.15): "
                     "the query tiles of a row range no longer share their rows in the L2")
    for name, needle in (("qkv", "gemm8_kernel<0,"), ("out_proj", "gemm8_kernel<2, false>"), ("ffn_down", "gemm8_kernel<2, true>"),
                         ("attention", "attention_pipe_kernel<8>"), ("attention_one_block", "attention_stream_kernel<16>"),
                         ("rescore", "rescore_kernel")):
        t = traffic(needle)
        if t:
            out[name] = t
    util = {}
    for c in sq.get("counters", []):
        if "mfma_busy_share" in c:
            util[c["kernel"]] = {k: c[k] for k in ("mfma_busy_share", "clock_GHz", "wave_parked", "wave_issue_stalled", "wave_issuing") if k in c}
    out["mfma_util"] = util
    json.dump(out, open(pre + "pmc_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
