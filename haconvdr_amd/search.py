"""Block-wise retrieval: host-side mirror of ``search_one_by_one_with_faiss``
(src/test_HAConvDR_topiocqa.py:74-162 = src/test_HAConvDR_qrecc.py:74-162).

Same arguments, same file formats, same result contract (float64 scores, int64
passage ids, global top-``topN`` in the first ``topN`` columns, earlier block wins
score ties), but the per-block python merge (:111-149) is replaced by keeping every
block resident in HBM and running ONE exact top-k over all of them, with the
``passage_embedding2id[I]`` remap (:110) fused into the result kernel.
"""
import logging
import os

import numpy as np

logger = logging.getLogger(__name__)


def iter_embedding_blocks(passage_embeddings_dir, passage_block_num, mmap=True):
    """Yield (emb float32 [n,768], ids int64 [n]) for block 0,1,… ; stops at the first
    block that cannot be loaded, like the reference's bare ``except: break`` (:94-95).
    mmap=True maps the pickled ndarray payload in place (haconvdr_amd.passages) instead of copying
    the 7.7 GB block through ``pickle.load`` — the reference's dominant wall time at search."""
    from .passages import read_embedding_block
    for block_id in range(passage_block_num):
        try:
            emb, ids = read_embedding_block(passage_embeddings_dir, block_id, mmap=mmap)
        except Exception:
            break
        yield emb, np.asarray(ids)


def search_blocks(index, blocks, query_embeddings, topN):
    """Resident multi-block search.  blocks: iterable of (emb, ids) in block order.
    Returns (D float64 [nq, topN], I int64 [nq, topN])."""
    import torch
    dev = torch.device("cuda", index.devices[0])
    index.reset()
    id_parts, sizes, last_ids = [], [], []
    for emb, ids in blocks:
        index.add(emb)                                    # :98 (blocks stay resident; no reset between them)
        ids = np.asarray(ids, dtype=np.int64)
        id_parts.append(ids)
        sizes.append(len(ids))
        last_ids.append(int(ids[-1]) if len(ids) else -1)
    q = np.ascontiguousarray(query_embeddings, dtype=np.float32)
    nq = q.shape[0]
    if not id_parts:
        raise ValueError("no passage block could be loaded")  # the reference fails too (None has no len, :152)
    id_map = torch.from_numpy(np.concatenate(id_parts)).to(dev)
    if len(index.devices) == 1:
        D, I = index.search_tensor(torch.from_numpy(q).to(dev), topN, id_map=id_map)
        D, I = D.cpu().numpy(), I.cpu().numpy()
    else:
        D, I = index.search(q, topN)
        idm = id_map.cpu().numpy()
        I = np.where(I >= 0, idm[np.clip(I, 0, None)], -1)
    index.reset()                                         # :122
    D = D.astype(np.float64)                              # .tolist() widens to python float (:111)
    total = int(sum(sizes))
    if total < topN:
        # Degenerate corpus (< topN rows in all blocks together): the reference's padded
        # slots carry score -FLT_MAX and, through numpy's negative indexing at :110
        # (I == -1), the LAST id of their block; earlier blocks' pads sort first (:138).
        pads = []
        for n_b, lid in zip(sizes, last_ids):
            pads += [lid] * max(0, topN - n_b)
        I[:, total:] = np.asarray(pads[:topN - total], dtype=np.int64)[None, :]
    return D, I


def search_one_by_one(args, passage_embeddings_dir, index, query_embeddings, topN):
    """Drop-in for search_one_by_one_with_faiss(args, dir, index, Q, topN)."""
    blocks = iter_embedding_blocks(passage_embeddings_dir, args.passage_block_num)
    merged_D, merged_I = search_blocks(index, blocks, query_embeddings, topN)
    logger.info(merged_I.shape)
    return merged_D, merged_I


class ResidentCorpus:
    """All passage blocks of a directory resident in HBM, searched many times (the reference reloads
    every block for every evaluation run, :77-123).  Same result contract as search_one_by_one."""

    def __init__(self, passage_embeddings_dir, passage_block_num, index=None, device=0):
        import torch
        from .index import FlatIPIndex
        self.index = index if index is not None else FlatIPIndex(768, devices=(device,))
        self.index.reset()
        ids, self.sizes = [], []
        for emb, bid in iter_embedding_blocks(passage_embeddings_dir, passage_block_num):
            self.index.add(emb)
            ids.append(np.asarray(bid, dtype=np.int64))
            self.sizes.append(len(bid))
        if not ids:
            raise ValueError("no passage block could be loaded")
        self.dev = torch.device("cuda", self.index.devices[0])
        self.id_map = torch.from_numpy(np.concatenate(ids)).to(self.dev)

    @property
    def ntotal(self):
        return self.index.ntotal

    def search(self, query_embeddings, topN):
        """query_embeddings: float32 [nq,768] (numpy or CUDA tensor) -> (D float64 [nq,topN], I int64 [nq,topN])."""
        import torch
        q = query_embeddings if isinstance(query_embeddings, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(query_embeddings, dtype=np.float32))
        D, I = self.index.search_tensor(q.to(self.dev), topN, id_map=self.id_map)
        return D.cpu().numpy().astype(np.float64), I.cpu().numpy()
