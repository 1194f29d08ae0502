"""bench.py as the driver runs it, on the GPU box: the N > 1 control flow (ranks started by bench.py itself, shards,
data-parallel encode, all-gather of embeddings and of packed top-k keys, on-device merge) rehearsed with two ranks sharing
the one GPU of this pool's boxes (gloo; RCCL wants one device per rank), and the merged result of a step checked against a
single-process search over the same rows.  Reference split: src/test_HAConvDR_topiocqa.py:55-66 (faiss shard=True)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_rank_rehearsal_matches_single_process(tmp_path):
    import torch
    dump = str(tmp_path / "step.npz")
    env = dict(os.environ, HAC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "2000000", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--extras", "default", "--north-star-rows", "1500000", "--dump-results", dump]      # (default = what the driver's N > 1 runs execute)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["steps"] == 2 and js["warmup"] == 1 and js["scaling"] == "weak" and "rehearsal" in js
    # the N > 1 line verifies itself (round 5: the first real 8-GPU run cannot be rehearsed): the backend saw both ranks, the
    # all-gather delivers the slabs in rank order, both collectives are timed, the sharded north-star step equals a single-GPU
    # exact search on eight queries, and the like-for-like anchor's VALUE is in the line
    c = js["collective"]
    assert c["backend"] == "gloo" and c["world"] == 2 and c["ranks_seen"] == 2 and c["allgather_slabs_in_rank_order"] is True and "error" not in c
    assert c["allgather_emb_ms"] > 0 and c["allgather_keys_ms"] > 0 and c["allgather_keys_bytes_per_rank"] == 1000 * 100 * 8
    v = js["north_star_10M"]["verify"]
    assert v["queries"] == 8 and v["rows"] == 1_500_000 and v["ids_equal"] is True and v["scores_equal"] is True and "scan" in v["referee"]
    a = js["cfg4_shard_step"]
    assert a["queries_per_sec"] > 0 and a["min_over_ranks"] <= a["queries_per_sec"] <= a["max_over_ranks"] * 1.0001 and a["rows_per_gpu"] == 1_000_000
    # configs[4] on N ranks: no collective, every rank writes the blocks it owns -- checked through the real encode loop (round 6)
    own = js["passages_L384"]["block_ownership"]
    assert own["every_block_once_by_its_owner"] is True and own["blocks"] == 5 and own["passages_in_blocks"] == own["passages"], own
    assert js["passages_L384"]["docs_per_sec_padded"] > 0 and js["passages_L384"]["gpus"] == 2
    w = js["wall_clock_s"]
    assert 0 < w["corpus_resident"] <= w["timed_steps_done"] <= w["extras_done"] < 600, w
    assert js["config"]["corpus_rows"] == 2_000_000 and js["config"]["rows_per_gpu"] == 1_000_000
    assert js["value"] > 0 and abs(js["value"] - 1000 / (js["ms_per_step"] * 1e-3)) < 1e-3 * js["value"]
    g = np.load(dump)
    assert g["emb"].shape == (1000, 768) and g["D"].shape == (1000, 100) and g["I"].shape == (1000, 100)
    # the same 2M rows in ONE index, the same embeddings: the sharded step's merged result must be this, bit for bit
    sys.path.insert(0, ROOT)
    import bench
    from haconvdr_amd.index import FlatIPIndex
    dev = torch.device("cuda", 0)
    idx = FlatIPIndex(768)
    bench.fill_index(idx, 0, int(g["rows"]), dev, int(g["block_rows"]))
    D, I = idx.search_tensor(torch.from_numpy(g["emb"]).to(dev), int(g["k"]))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(I.cpu().numpy(), g["I"])
    np.testing.assert_array_equal(D.cpu().numpy(), g["D"])
    assert int(g["I"].max()) >= 1_000_000 and int(g["I"].min()) < 1_000_000      # both shards contribute


def test_bench_launched_the_drivers_way_through_torch_distributed_run():
    """The driver starts the N > 1 bench as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE from the launcher), not through bench.py's own rank
    spawning: the same self-verifying line must come out of that launch too (two ranks sharing this box's GPU over gloo; default
    extras = what the driver's run executes)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HAC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "1000000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
           "--north-star-rows", "800000"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0's line only
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and "rehearsal" in js and js["value"] > 0
    c = js["collective"]
    assert c["world"] == 2 and c["ranks_seen"] == 2 and c["allgather_slabs_in_rank_order"] is True and "error" not in c
    v = js["north_star_10M"]["verify"]
    assert v["ids_equal"] is True and v["scores_equal"] is True and v["rows"] == 800_000
    assert js["passages_L384"]["block_ownership"]["every_block_once_by_its_owner"] is True
    assert js["cfg4_shard_step"]["rows_per_gpu"] == 500_000


def test_bench_four_rank_rehearsal_with_a_query_count_the_ranks_do_not_divide(tmp_path):
    """The N = 8 control flow of BASELINE configs[3] as far as one GPU box allows: this pool admits at most six processes on a
    card and the test process itself holds it, so FOUR ranks share it over gloo (VERDICT r3 asked for eight: the run with six
    ranks was killed by the box's process guard; the eight-rank plumbing itself runs on CPU in
    tests/test_host_logic.py::test_sharded_search_gloo_world[8]).  nq = 1001 is not a multiple of 4: every rank encodes
    ceil(1001 / 4) = 251 queries, the last slab ends in padding, all_gather + [:nq] must still hand every rank the same 1001
    embeddings, and the merged (D, I) must equal a single-process search."""
    import torch
    dump = str(tmp_path / "step4.npz")
    env = dict(os.environ, HAC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rows", "1200000", "--nq", "1001", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--dump-results", dump]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    js = json.loads(lines[0])
    assert js["n_gpus"] == 4 and "rehearsal" in js and js["config"]["per_gpu_rows"] == 300_000 and "like_for_like_n1" in js
    c = js["collective"]
    assert c["world"] == 4 and c["ranks_seen"] == 4 and c["allgather_slabs_in_rank_order"] is True and "error" not in c
    assert c["allgather_emb_bytes_per_rank"] == 251 * 768 * 4 and c["allgather_keys_bytes_per_rank"] == 1001 * 100 * 8
    g = np.load(dump)
    assert g["emb"].shape == (1001, 768) and g["I"].shape == (1001, 100)
    sys.path.insert(0, ROOT)
    import bench
    from haconvdr_amd.index import FlatIPIndex
    dev = torch.device("cuda", 0)
    idx = FlatIPIndex(768)
    bench.fill_index(idx, 0, int(g["rows"]), dev, int(g["block_rows"]))
    D, I = idx.search_tensor(torch.from_numpy(g["emb"]).to(dev), int(g["k"]))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(I.cpu().numpy(), g["I"])
    np.testing.assert_array_equal(D.cpu().numpy(), g["D"])
    assert len(np.unique(g["I"] // 300_000)) == 4      # every shard contributes


def test_last_clock_reports_a_plausible_shader_clock():
    """hac_encoder_last_clock (bench.py's sustained_shader_clock_MHz): workgroup 0 of the FFN-up kernel reads (s_memtime,
    s_memrealtime) at its first and last instruction when class profiling is on: a shader clock inside the part's range over
    an interval about as long as the launch's hipEvent bracket; nothing without profiling."""
    from haconvdr_amd import synth
    from haconvdr_amd.encoder import ANCEEncoder
    enc = ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 2, rich=False))
    ids, _ = synth.token_batch(3, 320, 512, fixed_len=512)
    mask = np.ones_like(ids)
    enc(ids, mask)
    assert enc.last_clock_mhz() is None
    enc.set_profiling(True, classes="all")
    enc(ids, mask)
    ms = enc.profile_drain_class("ffn_up")
    mhz, seconds = enc.last_clock_mhz()
    enc.set_profiling(False)
    assert enc.last_plan().startswith("gemm=gemm8") and len(ms) == 1
    assert 400.0 < mhz < 2600.0, mhz
    assert 0.5 * ms[-1] * 1e-3 < seconds <= 1.05 * ms[-1] * 1e-3, (seconds, ms)
