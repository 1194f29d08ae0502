#!/usr/bin/env python3
"""Summarise a rocprofv3 output directory into the small files kept under profiles/.

  python tools/summarize_profile.py <rocprof_dir> <out_prefix> [--rows N --nq Q]

Reads *_kernel_trace.csv (per-dispatch durations; hac kernels are grouped by name AND grid so
that the seed scan, the main scan and the 16-query HBM-regime scan are told apart) and, when
present, *_counter_collection.csv (PMC values per dispatch).  FETCH_SIZE is in KiB and, on gfx950,
counts a wide coalesced streaming read at exactly half its bytes (MI355X_MICROARCH.md §HBM), so
hbm_read_bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 is exact for 16-B/lane stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("hac::(anonymous namespace)::", "")
    return name.split("(")[0]


def split_bimodal(groups, meta):
    """The seed scan and the main scan share kernel and grid; their durations differ by >10x.
    Split such groups at the geometric mean so that each population gets its own row."""
    out, om = {}, {}
    for key, us in groups.items():
        lo, hi = min(us), max(us)
        if lo > 0 and hi > 4 * lo:
            cut = (lo * hi) ** 0.5
            a = [u for u in us if u < cut]
            b = [u for u in us if u >= cut]
            out[key + ("short",)] = a
            out[key + ("long",)] = b
            om[key + ("short",)] = om[key + ("long",)] = meta[key]
        else:
            out[key + ("",)] = us
            om[key + ("",)] = meta[key]
    return out, om


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    extra = {}
    args = sys.argv[3:]
    for i in range(0, len(args), 2):
        extra[args[i].lstrip("-")] = int(args[i + 1])
    out = {"source": os.path.basename(os.path.normpath(d)), **extra, "kernels": []}
    traces = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    groups = defaultdict(list)
    meta = {}
    disp = {}
    for t in traces:
        for r in csv.DictReader(open(t)):
            if "hac::" not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
            groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[key] = dict(lds=int(r["LDS_Block_Size"]), vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]),
                             sgpr=int(r["SGPR_Count"]), scratch=int(r["Scratch_Size"]))
            disp[r["Dispatch_Id"]] = key
    groups, meta = split_bimodal(groups, meta)
    for key, us in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        us_sorted = sorted(us)
        out["kernels"].append({"kernel": key[0] + (f" [{key[4]} runs]" if key[4] else ""), "grid_threads": [int(key[1]), int(key[2])], "block": int(key[3]),
                               "calls": len(us), "avg_us": round(sum(us) / len(us), 2), "min_us": round(us_sorted[0], 2),
                               "median_us": round(us_sorted[len(us) // 2], 2), "max_us": round(us_sorted[-1], 2),
                               "total_ms": round(sum(us) / 1e3, 3), **meta[key]})
    counters = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if counters:
        agg = defaultdict(lambda: defaultdict(list))
        rows = []
        durs = defaultdict(list)
        for c in counters:
            for r in csv.DictReader(open(c)):
                if "hac::" not in r["Kernel_Name"]:
                    continue
                key = (short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"])
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                rows.append((key, r["Counter_Name"], float(r["Counter_Value"]), dur))
                durs[key].append(dur)
        agg_dur = defaultdict(list)
        for key, name, val, dur in rows:
            lo, hi = min(durs[key]), max(durs[key])
            tag = ""
            if lo > 0 and hi > 4 * lo:
                tag = " [short runs]" if dur < (lo * hi) ** 0.5 else " [long runs]"
            agg[(key[0] + tag, key[1], key[2])][name].append(val)
            agg_dur[(key[0] + tag, key[1], key[2])].append(dur)
        out["counters"] = []
        for key, cs in agg.items():
            row = {"kernel": key[0], "grid_threads_total": int(key[1]), "block": int(key[2])}
            for name, vals in cs.items():
                row[name + "_avg"] = sum(vals) / len(vals)
                row[name + "_n"] = len(vals)
            if "FETCH_SIZE" in cs:
                row["hbm_read_bytes_per_launch"] = row["FETCH_SIZE_avg"] * 1024 * 2
            if "WRITE_SIZE" in cs:
                row["hbm_write_bytes_per_launch"] = row["WRITE_SIZE_avg"] * 1024
            if "GRBM_GUI_ACTIVE" in cs:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = kernel cycles; the SQ counters are summed over the chip's
                # 1024 SIMDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* quad-cycles
                # (MI355X_MICROARCH.md, cycle constants)
                cyc = row["GRBM_GUI_ACTIVE_avg"] / 8.0
                row["kernel_cycles"] = cyc
                d_us = sum(agg_dur[key]) / len(agg_dur[key])
                row["avg_us_under_pmc"] = round(d_us, 2)
                if d_us > 0:
                    row["clock_GHz"] = round(cyc / d_us / 1e3, 3)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cyc > 0:
                    row["mfma_busy_share"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (cyc * 1024.0), 4)
                if "SQ_WAVE_CYCLES" in cs and row["SQ_WAVE_CYCLES_avg"] > 0:
                    wv = row["SQ_WAVE_CYCLES_avg"]
                    for src, dst in (("SQ_WAIT_ANY", "wave_parked"), ("SQ_WAIT_INST_ANY", "wave_issue_stalled"), ("SQ_ACTIVE_INST_ANY", "wave_issuing")):
                        if src in cs:
                            row[dst] = round(row[src + "_avg"] / wv, 4)
            out["counters"].append(row)
    json.dump(out, open(prefix + ".json", "w"), indent=1)
    with open(prefix + ".md", "w") as f:
        f.write(f"# rocprofv3 summary: {out['source']}\n\n| kernel | grid (threads) | block | calls | avg µs | median µs | min µs | max µs | LDS B | VGPR | AGPR | scratch |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k in out["kernels"]:
            f.write(f"| {k['kernel']} | {k['grid_threads'][0]}x{k['grid_threads'][1]} | {k['block']} | {k['calls']} | {k['avg_us']} | {k['median_us']} | {k['min_us']} | {k['max_us']} | {k['lds']} | {k['vgpr']} | {k['agpr']} | {k['scratch']} |\n")
        if "counters" in out:
            f.write("\n## PMC counters (per launch averages)\n\n")
            for c in out["counters"]:
                f.write("- " + json.dumps(c) + "\n")
    print(open(prefix + ".md").read())


if __name__ == "__main__":
    main()
