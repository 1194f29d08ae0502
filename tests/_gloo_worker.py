"""One rank of the gloo tests (world 2 and world 8) of the sharded search plumbing (CPU, test doubles)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from haconvdr_amd.sharded import ShardedSearcher, shard_range  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import doubles  # noqa: E402
from tests.golden import cases  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, nq, k = 1501, 7, 100
    x, q, ids = cases.search_case_inputs("dup", 9090, n, nq)   # duplicated rows straddle the shard boundary
    lo, hi = shard_range(n, rank, world)
    s = ShardedSearcher(None, shard_base=lo, id_map=ids, local_keys=doubles.make_local_keys(oracle, x[lo:hi]),
                        merge=doubles.merge, to_results=doubles.to_results)
    D, I = s.search(torch.from_numpy(q), k)
    # every rank holds the same, complete answer == the reference's block merge with shards as blocks
    bounds = [shard_range(n, r, world) for r in range(world)]
    mD, mI = oracle.search_one_by_one([(x[a:b], ids[a:b]) for a, b in bounds], q, k)
    ok = np.array_equal(I, mI) and np.array_equal(D.astype(np.float64), mD)
    oD, oI = oracle.flat_ip_search(x, q, k)
    ok = ok and np.array_equal(I, ids[oI]) and np.array_equal(D, oD)
    # the step of `bench.py --gpus N` with a query count the ranks do not divide: every rank "encodes" ceil(nq / N) queries
    # (its slab of the query matrix, the last slab partly padding), ONE all-gather brings all of them to every rank, the
    # first nq rows are searched over the local shard, ONE all-gather of packed keys + merge finishes the step
    nq_loc = (nq + world - 1) // world
    slab = torch.zeros((nq_loc, q.shape[1]), dtype=torch.float32)
    mine = q[rank * nq_loc:min(nq, (rank + 1) * nq_loc)]
    slab[:len(mine)] = torch.from_numpy(mine)
    allq = torch.empty((world * nq_loc, q.shape[1]), dtype=torch.float32)
    dist.all_gather_into_tensor(allq, slab)
    ok = ok and np.array_equal(allq[:nq].numpy(), q)
    D2, I2 = s.search(allq[:nq], k)
    ok = ok and np.array_equal(I2, I) and np.array_equal(D2, D)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    print(f"rank {rank}: {'OK' if ok else 'MISMATCH'} shard=[{lo},{hi})", flush=True)
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
