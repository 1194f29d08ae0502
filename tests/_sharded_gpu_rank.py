"""One rank of tests/test_search_gpu.py::test_sharded_search_across_ranks_sharing_the_gpu (started by that test).

argv: out_path n_rows nq k.  Every rank builds ITS shard of the corpus on the HIP index (device 0: the box has one GPU,
so the ranks share it and the collective runs over gloo -- RCCL refuses two ranks on one device; with HAC_TEST_BACKEND=nccl,
set by test_sharded_search_rccl_one_gpu_per_rank on a box with >= 2 GPUs, rank r uses GPU r and RCCL), encodes nothing, and
searches through ShardedSearcher with its HIP defaults; queries are split data-parallel and all-gathered first, as
bench.py does with the embeddings.  Rank 0 saves (D, I)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_path, n_rows, nq, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rccl = os.environ.get("HAC_TEST_BACKEND") == "nccl"
    device = rank if rccl else 0
    torch.cuda.set_device(device)
    if rccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from haconvdr_amd.index import FlatIPIndex
    from haconvdr_amd.sharded import ShardedSearcher, shard_range
    from tests.golden import cases
    x, q, _ = cases.search_case_inputs("dup", 0x5AAD, n_rows, nq)    # the same corpus on every rank; each keeps its rows.
    # "dup": the second half repeats the first, so every score ties across shards (order = global position)
    lo, hi = shard_range(n_rows, rank, world)
    idx = FlatIPIndex(768, devices=(device,))
    half = (hi - lo) // 2
    idx.add(x[lo:lo + half])                                     # two segments per shard
    idx.add(x[lo + half:hi])
    nq_loc = -(-nq // world)
    mine = np.zeros((nq_loc, 768), dtype=np.float32)
    part = q[rank * nq_loc:(rank + 1) * nq_loc]
    mine[:len(part)] = part
    allq = torch.empty((world * nq_loc, 768), dtype=torch.float32, device="cuda")
    dist.all_gather_into_tensor(allq, torch.from_numpy(mine).cuda())
    D, I = ShardedSearcher(idx, shard_base=lo).search(allq[:nq], k)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, D=D.cpu().numpy(), I=I.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
