#!/usr/bin/env python3
"""The figures of a bench.py JSON line that README / DESIGN quote, one per line.   python tools/show_bench.py FILE.json"""
import json
import sys


def main():
    lines = [ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")]
    j = json.loads(lines[-1])
    r = j["roofline"]
    print(f"value {j['value']} {j['unit']}  ms/step {j['ms_per_step']}  n_gpus {j['n_gpus']}  steps {j['steps']}")
    print(f"roofline: {r.get('kernel', '')[:40]}  frac {r.get('frac')}  kernel_ms {r.get('kernel_ms')}  clock {r.get('sustained_shader_clock_MHz')} MHz"
          f"  frac@clock {r.get('frac_of_clock_held_peak')}  traffic {r.get('traffic')}")
    es = r.get("encoder_stack")
    if es:
        print(f"encoder stack: {es['ms_per_step']} ms  frac {es['mfma_bf16_frac']}  frac@clock {es.get('mfma_bf16_frac_of_clock_held_peak')}")
    s = r.get("search", r)
    print(f"search: {s.get('kernel')}  kernel_ms {s.get('kernel_ms')}  frac {s.get('frac')}  frac@clock {s.get('frac_of_clock_held_peak')}  traffic {s.get('traffic')}")
    for k in r.get("kernels", []):
        print(f"   {k['class']:12s} {k['ms_per_step']:8.3f} ms  share {k['share_of_step']:.3f}  frac {k.get('frac_of_2.5PF', k.get('frac'))}")
    cb = j.get("cpu_baseline")
    if cb:
        print(f"cpu_baseline: {cb['value']} {cb['unit']} on {cb['cores']} cores ({cb['kind']})")
        tb = (cb.get("encode") or {}).get("timed_batch_check")
        if tb:
            print(f"   timed batch vs oracle: 1-cos {max(tb['one_minus_cos']):.2e}  inter-sequence min {tb.get('inter_sequence_min_1_minus_cos'):.2e}  "
                  f"error/spread {tb.get('error_over_inter_sequence_distance'):.3f}  centred {tb.get('centred_1_minus_cos'):.2e}  rel L2 {tb.get('relative_l2'):.3f}  "
                  f"passes {tb.get('passes_fixture_scaled_bounds')}  neighbour would pass {tb.get('neighbours_embedding_would_pass')}")
    for name in ("step_parts", "real_query_lengths", "north_star_10M", "cfg4_shard_step", "cfg4_one_rank_parts", "cfg2_search_only", "collective"):
        if name in j:
            v = {k: x for k, x in j[name].items() if not isinstance(x, (dict, list)) and k not in ("what", "note")}
            print(f"{name}: {v}")
    for name in ("hbm_regime", "passages_L384", "three_call_protocol", "sustained", "attention_peaked"):
        if name in j:
            print(f"{name}: {json.dumps(j[name])[:600]}")
    sw = j.get("encoder_batch_sweep")
    if sw:
        for row in sw["rows"]:
            print(f"   B={row['B']:4d} L={row['L']:4d}  {row['ms']:8.4f} ms  frac {row['mfma_bf16_frac']}  graph_off {row.get('graph_off_ms')}")
    print("extras present:", [k for k in j if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                                       "dtype", "data", "config", "roofline", "cpu_baseline")])


if __name__ == "__main__":
    main()
