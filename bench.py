#!/usr/bin/env python3
"""Headline benchmark: queries/sec of exact inner-product top-100 search over a 768-d
passage-embedding corpus (BASELINE.json configs[1]: 1M x 768 corpus, 1000 queries,
pre-encoded embeddings resident in HBM, one MI355X).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A step = one search of the whole query batch over the resident corpus.  With N>1 the 1M-row
corpus is split into N contiguous shards (strong scaling: total work fixed), every rank
scans its shard for all queries, one RCCL all-gather of the packed per-shard top-k keys and
an on-device merge finish the step on every rank.

Rank 0 prints ONE JSON line; see DESIGN.md §measurement for the roofline arithmetic.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

D_EMB = 768
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PEAK_F32_MFMA_TF = 157.3    # fp32-input MFMA dense peak
PEAK_F16_MFMA_TF = 2500.0   # fp16 / bf16 MFMA dense peak


def gen_rows(seed, n, device):
    """Row-standardised Gaussian rows (||x|| = sqrt(768)), the shape of the ANCE head's output."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn((n, D_EMB), generator=g, device=device, dtype=torch.float32)
    return (x - x.mean(1, keepdim=True)) / x.std(1, unbiased=False, keepdim=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000, help="total corpus rows (configs[1]: 1M)")
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encode", action="store_true", help="skip the encode / end-to-end extras")
    ap.add_argument("--query-len", type=int, default=512, help="padded query length (TopiOCQA: 512, QReCC: 256)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from haconvdr_amd.index import FlatIPIndex, keys_to_results
    from haconvdr_amd.sharded import ShardedSearcher, shard_range

    # ---- synthetic data, generated in HBM (seeds per 125k-row chunk: same corpus for every N)
    lo, hi = shard_range(args.rows, rank, world)
    index = FlatIPIndex(D_EMB, devices=(local_rank,))
    CH = 125_000
    keep_for_cpu = []
    for c0 in range(0, args.rows, CH):
        a, b = max(lo, c0), min(hi, c0 + CH)
        if a >= b:
            continue
        xb = gen_rows(0xC0FFEE + c0 // CH, min(CH, args.rows - c0), dev)[a - c0:b - c0].contiguous()
        index.add_tensor(xb)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            keep_for_cpu.append(xb.cpu().numpy())
        del xb
    q = gen_rows(0xBEEF, args.nq, dev)
    torch.cuda.synchronize()
    searcher = ShardedSearcher(index, shard_base=lo)

    def step():
        return searcher.search(q, args.k)

    for _ in range(args.warmup):
        step()
    index.set_profiling(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        D, I = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    scan_ms = index.profile_drain()
    plan = index.last_plan()
    kname = plan.split(" ")[0]
    index.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = args.nq * args.steps / dt

    # ---- roofline of the dominant kernel (scan16_kernel), algorithmic figures per launch
    n_local = hi - lo
    scan_avg_ms = float(np.sum(scan_ms)) / args.steps if scan_ms else float("nan")   # a search may time several chunk scans
    alg_bytes = n_local * D_EMB * 4 + args.nq * D_EMB * 4 + args.nq * args.k * 12
    alg_flops = 2.0 * args.nq * n_local * D_EMB
    hbm_gbs = alg_bytes / (scan_avg_ms * 1e-3) / 1e9
    mfma_tf = alg_flops / (scan_avg_ms * 1e-3) / 1e12
    # fp32 arithmetic intensity nq/2 flop/B against the ridge 157.3 TF / 8 TB/s = 19.7 flop/B
    mfma_bound = (alg_flops / alg_bytes) > (PEAK_F32_MFMA_TF * 1e12) / (PEAK_HBM_GBS * 1e9)
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc_path) and world == 1:
        try:
            pmc = json.load(open(pmc_path))
            if pmc.get("rows") == args.rows and pmc.get("nq") == args.nq:
                traffic = pmc.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    if plan.startswith("split:"):
        # The dominant kernel is the fp16-MFMA prefilter (scan_split.inc): scanh_kernel<TERMS> runs TERMS fp16
        # products per (query, row) pair on the 2.5 PF/s matrix pipe and is followed by exact fp32 rescoring of the
        # ~k candidates it certifies.  `achieved` stays the ALGORITHMIC rate 2*nq*n*d / t (what the exact fp32
        # kernels would have to sustain: their ceiling is 157.3 TF); the kernel executes TERMS times that.
        terms = int(plan.split("scanh_kernel<")[1].split(">")[0])
        kname = f"scanh_kernel<{terms}>"
        roofline = {"kernel": kname, "plan": plan, "bound": "mfma", "achieved": round(mfma_tf, 2), "peak": PEAK_F16_MFMA_TF,
                    "unit": "TFLOP/s", "frac": round(mfma_tf / PEAK_F16_MFMA_TF, 4), "traffic": traffic,
                    "kernel_ms": round(scan_avg_ms, 4), "executed_TFLOPs": round(terms * mfma_tf, 2),
                    "executed_frac": round(terms * mfma_tf / PEAK_F16_MFMA_TF, 4),
                    "vs_fp32_mfma_peak": round(mfma_tf / PEAK_F32_MFMA_TF, 3),
                    "hbm_GBps_same_kernel": round(hbm_gbs, 1), "hbm_frac_same_kernel": round(hbm_gbs / PEAK_HBM_GBS, 4),
                    "note": "kernel_ms spans both scanh launches of a search (maxima-only seeding pass over the first sixteenth of "
                            "the corpus, then the full pass) and the threshold selection between them, summed over the "
                            "search's query chunks of 1024; peak = dense fp16 MFMA"}
    elif mfma_bound:
        roofline = {"kernel": kname, "plan": plan, "bound": "mfma", "achieved": round(mfma_tf, 2), "peak": PEAK_F32_MFMA_TF,
                    "unit": "TFLOP/s", "frac": round(mfma_tf / PEAK_F32_MFMA_TF, 4), "traffic": traffic,
                    "kernel_ms": round(scan_avg_ms, 4), "hbm_GBps_same_kernel": round(hbm_gbs, 1),
                    "hbm_frac_same_kernel": round(hbm_gbs / PEAK_HBM_GBS, 4)}
    else:
        roofline = {"kernel": kname, "plan": plan, "bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(hbm_gbs / PEAK_HBM_GBS, 4), "traffic": traffic,
                    "kernel_ms": round(scan_avg_ms, 4), "mfma_TFLOPs_same_kernel": round(mfma_tf, 2)}

    # ---- the same search by the exact fp32 kernels alone (prefilter off): the round's earlier headline path
    exact_kernels = None
    if world == 1 and plan.startswith("split:"):
        os.environ["HAC_SPLIT"] = "0"
        try:
            for _ in range(2):
                index.search_tensor(q, args.k)
            index.set_profiling(True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                De, Ie = index.search_tensor(q, args.k)
            torch.cuda.synchronize()
            dte = (time.perf_counter() - t1) / 3
            mse = float(np.sum(index.profile_drain())) / 3
            index.set_profiling(False)
            tfe = alg_flops / (mse * 1e-3) / 1e12
            exact_kernels = {"plan": index.last_plan(), "kernel_ms": round(mse, 4), "search_ms": round(dte * 1e3, 4),
                             "queries_per_sec": round(args.nq / dte, 1), "achieved_TFLOPs": round(tfe, 2),
                             "frac_of_fp32_mfma_peak": round(tfe / PEAK_F32_MFMA_TF, 4),
                             "ids_and_scores_equal_to_prefilter_path": bool(torch.equal(Ie, I) and torch.equal(De, D))}
        finally:
            del os.environ["HAC_SPLIT"]

    # ---- the HBM-bound regime of the same kernel (<= 16 queries per corpus pass), N=1 only
    hbm_regime = None
    if world == 1:
        q16 = q[:16].contiguous()
        for _ in range(3):
            index.search_tensor(q16, args.k)
        index.set_profiling(True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            index.search_tensor(q16, args.k)
        torch.cuda.synchronize()
        dt16 = (time.perf_counter() - t1) / reps
        ms16 = float(np.mean(index.profile_drain()))
        plan16 = index.last_plan()
        index.set_profiling(False)
        b16 = n_local * D_EMB * 4 + 16 * D_EMB * 4 + 16 * args.k * 12
        hbm_regime = {"nq": 16, "plan": plan16, "kernel_ms": round(ms16, 4), "search_ms": round(dt16 * 1e3, 4),
                      "achieved_GBps": round(b16 / (ms16 * 1e-3) / 1e9, 1),
                      "frac_of_8TBps": round(b16 / (ms16 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                      "queries_per_sec": round(16 / dt16, 1)}

    # ---- the same kernel family at other query-batch sizes per corpus pass (SURVEY §8d asks for 1, 8, 32)
    if world == 1 and hbm_regime is not None:
        sweep = []
        for nqs in (1, 8, 32):
            qs = q[:nqs].contiguous()
            for _ in range(2):
                index.search_tensor(qs, args.k)
            index.set_profiling(True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                index.search_tensor(qs, args.k)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t1) / 10
            mss = float(np.mean(index.profile_drain()))
            index.set_profiling(False)
            bs = n_local * D_EMB * 4 + nqs * D_EMB * 4 + nqs * args.k * 12
            sweep.append({"nq": nqs, "kernel": index.last_plan().split(" ")[0], "kernel_ms": round(mss, 4), "search_ms": round(dts * 1e3, 4),
                          "achieved_GBps": round(bs / (mss * 1e-3) / 1e9, 1), "mfma_TFLOPs": round(2.0 * nqs * n_local * D_EMB / (mss * 1e-3) / 1e12, 2)})
        hbm_regime["sweep"] = sweep

    # ---- CPU baseline: the oracle (C port, OpenMP) on the host cores, bounded sample of the same workload
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        xh = np.concatenate(keep_for_cpu)
        qh = q.cpu().numpy()
        cores = len(os.sched_getaffinity(0))
        os.environ["OMP_NUM_THREADS"] = str(cores)
        nq_probe = min(args.nq, 8 * cores)
        tp = time.perf_counter()
        oracle.flat_ip_search(xh, qh[:nq_probe], args.k)
        probe = time.perf_counter() - tp
        nq_s = int(min(args.nq, max(nq_probe, (args.cpu_seconds / max(probe, 1e-3)) * nq_probe // (8 * cores) * 8 * cores)))
        tp = time.perf_counter()
        oD, oI = oracle.flat_ip_search(xh, qh[:nq_s], args.k)
        tcpu = time.perf_counter() - tp
        same = bool(np.array_equal(oI, I[:nq_s].cpu().numpy()) and np.array_equal(oD, D[:nq_s].cpu().numpy()))
        cpu_baseline = {"value": round(nq_s / tcpu, 2), "unit": "queries/s", "cores": oracle.num_threads(), "kind": "port",
                        "sample": f"first {nq_s} of the {args.nq} queries over the full {args.rows}x768 corpus, "
                                  f"oracle/flat_ip_oracle.c (OpenMP, AVX2 fmaf chain), {tcpu:.1f} s",
                        "ids_and_scores_equal_to_gpu": same}

    # ---- extras: ANCE query encode (bf16 MFMA GEMMs) and end-to-end encode + top-k ------------------
    encode = end_to_end = None
    if not args.no_encode:
        from haconvdr_amd import synth
        from haconvdr_amd.encoder import ANCEEncoder
        enc = ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 12, rich=False), device=local_rank)
        Lq = args.query_len
        nq_loc = (args.nq + world - 1) // world                       # queries are encoded data-parallel
        tok, _ = synth.token_batch(0x70C + rank, nq_loc, Lq, fixed_len=Lq)   # fully padded = the reference's behaviour
        ids_t = torch.from_numpy(tok.astype(np.int64)).to(dev)
        mask_t = torch.ones_like(ids_t)

        def e2e_step(mask=None):
            emb = enc(ids_t, mask_t if mask is None else mask)
            if world > 1:
                allq = torch.empty((world * nq_loc, D_EMB), dtype=torch.float32, device=dev)
                dist.all_gather_into_tensor(allq, emb)
                emb = allq[:args.nq]
            return searcher.search(emb.contiguous(), args.k)

        for _ in range(2):
            enc(ids_t, mask_t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_enc = 3
        enc.set_profiling(True)
        for _ in range(n_enc):
            enc(ids_t, mask_t)
        torch.cuda.synchronize()
        dt_enc = (time.perf_counter() - t1) / n_enc
        stack_ms = float(np.sum(enc.profile_drain())) / n_enc     # a forward may run as several sub-batches
        enc.set_profiling(False)
        fl = nq_loc * 12.0 * (14155776.0 * Lq + 4.0 * Lq * Lq * 768.0) + nq_loc * 2.0 * 768 * 768
        encode = {"queries_per_sec_per_gpu": round(nq_loc / dt_enc, 1), "seq_len": Lq, "batch": nq_loc, "ms": round(dt_enc * 1e3, 3),
                  "layer_stack_ms": round(stack_ms, 3), "achieved_TFLOPs": round(fl / (stack_ms * 1e-3) / 1e12, 1),
                  "mfma_bf16_frac_of_2.5PF": round(fl / (stack_ms * 1e-3) / 2.5e15, 4), "dtype": "bf16 MFMA operands, fp32 accumulate/LN/softmax",
                  "weights": "synthetic N(0,0.02) RoBERTa-base"}
        # passages (BASELINE configs[4] shape: L=384): fully padded = what the reference computes, and
        # varlen = only the real tokens (lens ~ clipped N(180, 80) in [8, 384])
        Bp, Lp = 1000, 384
        ptok, _ = synth.token_batch(0xD0C + rank, Bp, Lp, fixed_len=Lp)
        plens = np.clip(np.rint(180.0 + 80.0 * synth.normal(0x1E45 + rank, (Bp,))), 8, Lp).astype(np.int64)
        pid_t = torch.from_numpy(ptok.astype(np.int64)).to(dev)
        full_mask = torch.ones_like(pid_t)
        var_mask = (torch.arange(Lp, device=dev)[None, :] < torch.from_numpy(plens).to(dev)[:, None]).to(torch.int64)
        pres = {}
        for name, m in (("padded", full_mask), ("varlen", var_mask)):
            for _ in range(2):
                enc(pid_t, m)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                enc(pid_t, m)
            torch.cuda.synchronize()
            pres[name] = Bp / ((time.perf_counter() - t1) / 3)
        # queries with their real lengths (uniform in [64, 512], padded to 512 with a prefix mask):
        # the reference computes the padding too; this encoder only the real tokens (varlen packing)
        qlens = 64 + (synth.uniform_u32(0x91E45 + rank, nq_loc) % np.uint32(max(1, Lq - 64 + 1))).astype(np.int64)
        qvar_mask = (torch.arange(Lq, device=dev)[None, :] < torch.from_numpy(qlens).to(dev)[:, None]).to(torch.int64)
        for _ in range(2):
            enc(ids_t, qvar_mask)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            enc(ids_t, qvar_mask)
        torch.cuda.synchronize()
        encode["queries_varlen"] = {"queries_per_sec_per_gpu": round(nq_loc / ((time.perf_counter() - t1) / 3), 1),
                                    "mean_len": round(float(qlens.mean()), 1), "lens": "uniform in [64, 512], prefix mask"}
        encode["passages_L384"] = {"docs_per_sec_per_gpu_padded": round(pres["padded"], 1),
                                   "docs_per_sec_per_gpu_varlen": round(pres["varlen"], 1),
                                   "mean_len_varlen": round(float(plens.mean()), 1), "batch": Bp}
        # the whole passage-encoding loop of gen_doc_embeddings.py (record reader -> H2D -> encode -> block D2H ->
        # pickle-4 block files), on a 20k-passage synthetic collection with the varlen lengths above
        try:
            import shutil
            import tempfile
            from haconvdr_amd import passages as psg_mod
            n_p = 20000
            tmpd = tempfile.mkdtemp(prefix="hac_bench_")
            ptok_all, _ = synth.token_batch(0xD0C5 + rank, n_p, Lp, fixed_len=Lp)
            plens_all = np.clip(np.rint(180.0 + 80.0 * synth.normal(0x1E46 + rank, (n_p,))), 8, Lp).astype(np.int64)
            psg_mod.write_tokenized_passages(os.path.join(tmpd, "passages"), ptok_all.astype(np.int32), plens_all)
            coll = psg_mod.TokenizedPassages(os.path.join(tmpd, "passages"))
            psg_mod.encode_passages(enc, coll, os.path.join(tmpd, "warm"), per_gpu_eval_batch_size=1000)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            psg_mod.encode_passages(enc, coll, os.path.join(tmpd, "out"), per_gpu_eval_batch_size=1000)
            torch.cuda.synchronize()
            encode["passages_L384"]["pipeline_docs_per_sec_per_gpu"] = round(n_p / (time.perf_counter() - t1), 1)
            encode["passages_L384"]["pipeline"] = f"encode_passages over {n_p} tokenized records (reader, H2D, encode, block files), mean len {plens_all.mean():.0f}"
            shutil.rmtree(tmpd, ignore_errors=True)
        except Exception as ex:   # an extra, never the headline
            encode["passages_L384"]["pipeline_error"] = repr(ex)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # CPU baseline of the encoder: the fp32 oracle (torch CPU ops, all host cores) on a bounded sample
            from oracle import ance_oracle
            sd_cpu = synth.ance_state_dict(0xA11CE, 12, rich=False)
            n_s = 8
            torch.set_num_threads(min(64, len(os.sched_getaffinity(0))))   # torch CPU GEMMs stop scaling well before 128+ threads
            tp = time.perf_counter()
            ref = ance_oracle.ance_forward(sd_cpu, tok[:n_s].astype(np.int64), np.ones((n_s, Lq), np.int64))
            tcpu = time.perf_counter() - tp
            got = enc(ids_t[:n_s], mask_t[:n_s]).cpu().numpy()
            cosd = 1.0 - (got * ref).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(ref, axis=1))
            encode["cpu_baseline"] = {"value": round(n_s / tcpu, 2), "unit": "queries/s", "cores": torch.get_num_threads(), "kind": "port",
                                      "sample": f"{n_s} of the {nq_loc} queries (L={Lq}), oracle/ance_oracle.py fp32 torch CPU ops, {tcpu:.1f} s",
                                      "max_1_minus_cos_vs_gpu": float(cosd.max())}
        e2e_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        n_e2e = 3
        for _ in range(n_e2e):
            e2e_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_e2e = (time.perf_counter() - t1) / n_e2e
        if world > 1:
            t = torch.tensor([dt_e2e], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_e2e = float(t.item())
        end_to_end = {"queries_per_sec": round(args.nq / dt_e2e, 1), "ms_per_step": round(dt_e2e * 1e3, 3),
                      "what": f"ANCE encode of {args.nq} queries (L={Lq}, data-parallel over {world} GPU) + exact top-{args.k} over {args.rows} passages"}
        # the same with the queries' real lengths (prefix masks, mean ~290 of 512 tokens) instead of full padding
        e2e_step(qvar_mask)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(n_e2e):
            e2e_step(qvar_mask)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_v = (time.perf_counter() - t1) / n_e2e
        if world > 1:
            t = torch.tensor([dt_v], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_v = float(t.item())
        end_to_end["queries_per_sec_real_lengths"] = round(args.nq / dt_v, 1)

    if rank == 0:
        out = {
            "metric": "queries/sec (top-100 exact IP search) over N-passage 768-d corpus",
            "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if not plan.startswith("split:") else "f32 (scores and order: fp32 fmaf chain; fp16-MFMA prefilter with a certified bound, exact fp32 rescoring)",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: " if (args.rows, args.nq, args.k) == (1_000_000, 1000, 100) else "custom: ")
                                   + f"{args.rows}x768 fp32 corpus resident in HBM, {args.nq} queries/step, "
                                   f"top-{args.k}, IP search only (pre-encoded embeddings)",
                       "corpus_rows": args.rows, "queries_per_step": args.nq, "k": args.k,
                       "parallelism": f"corpus sharded {world}-way, all-gather of packed top-k keys" if world > 1 else "single GPU"},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "exact_kernels": exact_kernels,
            "hbm_regime": hbm_regime,
            "encode": encode,
            "end_to_end": end_to_end,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
