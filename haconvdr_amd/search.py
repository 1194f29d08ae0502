"""Block-wise retrieval: host-side mirror of ``search_one_by_one_with_faiss``
(src/test_HAConvDR_topiocqa.py:74-162 = src/test_HAConvDR_qrecc.py:74-162).

Same arguments, same file formats, same result contract (float64 scores, int64
passage ids, global top-``topN`` in the first ``topN`` columns, earlier block wins
score ties), but the per-block python merge (:111-149) is replaced by keeping every
block resident in HBM and running ONE exact top-k over all of them, with the
``passage_embedding2id[I]`` remap (:110) fused into the result kernel.

Shape of the result.  The reference's merge loop appends BOTH lists' leftovers (:144-149), so with two or more
blocks it returns ``(nq, 2*topN)``: the global top-``topN`` followed by what is left of {top-``topN`` of blocks
0..B-2, top-``topN`` of block B-1} in merge order; its consumers slice ``[:topN]`` (:238-239).  By default this
module returns the ``(nq, topN)`` that is read; ``reference_shape=True`` returns the reference's literal matrix
(every column equal to the reference's, tested against the goldens' full ``ref_D`` / ``ref_I``) at the price of
a second search (the last block on its own).
"""
import logging
import os

import numpy as np

logger = logging.getLogger(__name__)


def _load_block(passage_embeddings_dir, block_id, mmap, readahead):
    from .passages import read_embedding_block
    emb, ids = read_embedding_block(passage_embeddings_dir, block_id, mmap=mmap)
    if readahead and isinstance(emb, np.memmap):
        # ask the kernel to start reading the payload now (asynchronous read-ahead into the page cache): by the time add()
        # walks these pages -- pageable -> pinned staging -> H2D, double-buffered inside hac_index_add -- they are resident
        try:
            import mmap as _mmap
            emb._mmap.madvise(_mmap.MADV_WILLNEED)
        except Exception:
            pass
    return emb, np.asarray(ids)


def iter_embedding_blocks(passage_embeddings_dir, passage_block_num, mmap=True, prefetch=True):
    """Yield (emb float32 [n,768], ids int64 [n]) for block 0,1,… ; stops at the first
    block that cannot be loaded, like the reference's bare ``except: break`` (:94-95).
    mmap=True maps the pickled ndarray payload in place (haconvdr_amd.passages) instead of copying
    the 7.7 GB block through ``pickle.load`` — the reference's dominant wall time at search.
    prefetch=True opens block b+1 on a background thread and starts its read-ahead while the caller still works on
    block b (its add(): the H2D of 7.7 GB; in the reference's own loop also its search, :98-122): disk and PCIe overlap."""
    if not prefetch or passage_block_num <= 1:
        for block_id in range(passage_block_num):
            try:
                yield _load_block(passage_embeddings_dir, block_id, mmap, False)
            except Exception:
                break
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = pool.submit(_load_block, passage_embeddings_dir, 0, mmap, True)
        for block_id in range(passage_block_num):
            try:
                cur = nxt.result()
            except Exception:
                break
            if block_id + 1 < passage_block_num:
                nxt = pool.submit(_load_block, passage_embeddings_dir, block_id + 1, mmap, True)
            yield cur


def _with_last_flag(it):
    """(item, is_last) for every item of an iterator (one item of look-ahead)."""
    it = iter(it)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, False
        cur = nxt
    yield cur, True


def _search_resident(index, q, dev, id_parts, sizes, last_ids, topN):
    """One exact top-topN over the blocks now resident in `index` (rows in block order), ids remapped on the device,
    the reference's padding of a degenerate corpus reproduced.  Resets the index (:122)."""
    import torch
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


def search_blocks(index, blocks, query_embeddings, topN, reference_shape=False):
    """Resident multi-block search.  blocks: iterable of (emb, ids) in block order.
    Returns (D float64 [nq, topN], I int64 [nq, topN]); with reference_shape=True and two or more blocks the
    reference's literal (nq, 2*topN) matrices (module docstring)."""
    import torch
    dev = torch.device("cuda", index.devices[0])
    q = np.ascontiguousarray(query_embeddings, dtype=np.float32)
    index.reset()
    id_parts, sizes, last_ids = [], [], []
    head = None                                           # reference_shape: result over blocks 0..B-2
    for (emb, ids), is_last in _with_last_flag(blocks):
        if reference_shape and is_last and id_parts:
            head = _search_resident(index, q, dev, id_parts, sizes, last_ids, topN)
            id_parts, sizes, last_ids = [], [], []
        index.add(emb)                                    # :98 (blocks stay resident; no reset between them)
        ids = np.asarray(ids, dtype=np.int64)
        id_parts.append(ids)
        sizes.append(len(ids))
        last_ids.append(int(ids[-1]) if len(ids) else -1)
    if not id_parts:
        raise ValueError("no passage block could be loaded")  # the reference fails too (None has no len, :152)
    D, I = _search_resident(index, q, dev, id_parts, sizes, last_ids, topN)
    if head is None:
        return D, I
    # :131-149 on (merged[:topN], cur[:topN]): a stable two-way merge, the running list first on equal scores (`>=`,
    # :138), BOTH leftovers appended.  Both lists are sorted descending (pads, -FLT_MAX, last), so the merge is a stable
    # sort of their concatenation by descending score.
    cD, cI = np.concatenate([head[0], D], 1), np.concatenate([head[1], I], 1)
    order = np.argsort(-cD, axis=1, kind="stable")
    return np.take_along_axis(cD, order, 1), np.take_along_axis(cI, order, 1)


def search_one_by_one(args, passage_embeddings_dir, index, query_embeddings, topN, reference_shape=False):
    """Drop-in for search_one_by_one_with_faiss(args, dir, index, Q, topN).  reference_shape=True: the reference's
    literal (nq, 2*topN) result when two or more blocks load (:144-162), see the module docstring."""
    blocks = iter_embedding_blocks(passage_embeddings_dir, args.passage_block_num)
    merged_D, merged_I = search_blocks(index, blocks, query_embeddings, topN, reference_shape=reference_shape)
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
