"""Corpus sharding across GPUs — one process per GPU, torch.distributed over RCCL/xGMI.

SURVEY.md §8(e): rows are independent, so the corpus is partitioned by contiguous
global-offset range (what faiss ``shard=True`` does inside one process,
src/test_HAConvDR_topiocqa.py:55-66).  Every rank computes an exact local top-k whose
keys already carry GLOBAL positions, then ONE all-gather of the packed keys
([nq, k] uint64 = 8 B per entry; 0.8 MB per rank at nq=1000, k=100) is followed by a
deterministic R-way merge on every rank.  Key order (score desc, global position asc)
equals the reference's sequential ``>=`` block merge (:138) when shards are in block
order.  No other collective touches the data path.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world_size):
    """Rows [lo, hi) of rank: contiguous ceil(N/R)-sized ranges."""
    per = (n_total + world_size - 1) // world_size
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


class ShardedSearcher:
    """search(q, k) over a corpus sharded across the ranks of ``group``.

    local_keys(q, k, pos_base) -> int64(uint64 bits) tensor [nq, k]   (default: the HIP index)
    merge(lists [R, nq, k])    -> [nq, k]                              (default: HIP merge kernel)
    to_results(keys, id_map)   -> (D, I)                               (default: HIP kernel)
    The three hooks exist so that the distributed plumbing can be exercised on CPU with
    gloo and a test double; the product path always uses the HIP defaults.
    """

    def __init__(self, index, shard_base, group=None, id_map=None, local_keys=None, merge=None, to_results=None):
        from . import index as _index
        self.index = index
        self.shard_base = int(shard_base)
        self.group = group
        self.id_map = id_map
        self._local_keys = local_keys or (lambda q, k, base: index.search_keys_tensor(q, k, pos_base=base))
        self._merge = merge or _index.merge_keys
        self._to_results = to_results or _index.keys_to_results

    def search_keys(self, q, k):
        keys = self._local_keys(q, k, self.shard_base)
        if not dist.is_initialized():
            return keys                                   # a single process holding the whole corpus
        world = dist.get_world_size(self.group)
        nq, k = keys.shape
        gathered = torch.empty((world * nq, k), dtype=keys.dtype, device=keys.device)   # rank-major slabs
        dist.all_gather_into_tensor(gathered, keys.contiguous(), group=self.group)
        return self._merge(gathered.view(world, nq, k))

    def search(self, q, k):
        return self._to_results(self.search_keys(q, k), self.id_map)
