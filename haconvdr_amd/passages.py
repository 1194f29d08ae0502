"""Passage-embedding generation: host-side mirror of gen_doc_embeddings.py (SURVEY.md §8 a7–a9, f-1, f-2).

* ``TokenizedPassages``   — bulk reader of the tokenized ``passages`` file the reference's
  ``EmbeddingCache`` reads one record at a time (src/utils.py:300-350; format written by
  gen_tokenized_doc.py:164-179,244): ``total_number`` records of ``4 + 4·L`` bytes,
  ``[len: uint32 big-endian][ids: int32 × L, little-endian]``, plus ``passages_meta`` JSON
  ``{"type": "int32", "total_number": N, "embedding_size": L}``.  Memory-mapped, sliced in bulk.
* ``encode_passages``     — the loop of ``InferenceEmbeddingFromStreamDataLoader``
  (gen_doc_embeddings.py:65-158): same batch size rule (:73), same block rule
  (``2_500_000 // batch`` batches per block, :87-88,:127), same output files
  ``passage_emb_block_{b}.pb`` (float32 [n,768]) / ``passage_embid_block_{b}.pb`` (int64 [n], the
  enumerate index = file offset, src/utils.py:140-143), pickle protocol 4 (:131-135).
  Differences are in how, not what: records are parsed in bulk, only real tokens are encoded
  (varlen), batches stay on the GPU until a block is flushed, and with several ranks whole blocks
  are dealt round-robin so the union of the files equals the single-process output.
* ``write_embedding_block`` / ``read_embedding_block`` — the block format; the reader maps the
  pickled ndarray payload instead of copying it through ``pickle.load`` (the reference's dominant
  wall time at search, test_HAConvDR_topiocqa.py:82-93).
"""
import json
import os
import pickle
import pickletools

import numpy as np


class TokenizedPassages:
    def __init__(self, base_path):
        self.base_path = base_path
        with open(base_path + "_meta", "r") as f:
            meta = json.load(f)
        self.dtype = np.dtype(meta["type"])
        self.total_number = int(meta["total_number"])
        self.L = int(meta["embedding_size"])
        self.record_size = self.L * self.dtype.itemsize + 4           # src/utils.py:307-308
        if self.dtype != np.dtype("int32"):
            raise ValueError(f"unsupported token dtype {self.dtype} (the reference writes int32)")
        size = os.path.getsize(base_path)
        if size < self.total_number * self.record_size:
            raise ValueError(f"{base_path}: {size} bytes < {self.total_number} records of {self.record_size}")
        self._mm = np.memmap(base_path, dtype=np.uint8, mode="r", shape=(self.total_number, self.record_size))

    def __len__(self):
        return self.total_number

    def batch(self, lo, hi):
        """Records [lo, hi) -> (ids int32 [m, L], lens int32 [m]) — one vectorised pass."""
        rec = self._mm[lo:hi]
        lens = rec[:, :4].astype(np.uint32)
        lens = (lens[:, 0] << 24) | (lens[:, 1] << 16) | (lens[:, 2] << 8) | lens[:, 3]   # big-endian (:325)
        ids = np.ascontiguousarray(rec[:, 4:]).view(np.int32)
        return ids, lens.astype(np.int32)

    def __getitem__(self, key):
        """EmbeddingCache.__getitem__ (src/utils.py:336-342): (passage_len, ids)."""
        if key < 0 or key > self.total_number:
            raise IndexError(f"Index {key} is out of bound for cached embeddings of size {self.total_number}")
        ids, lens = self.batch(key, key + 1)
        return int(lens[0]), ids[0]


def write_tokenized_passages(base_path, ids, lens):
    """Writer of the same format (gen_tokenized_doc.py:164-179,244) — used by tests and tools."""
    ids = np.ascontiguousarray(ids, dtype="<i4")
    n, L = ids.shape
    with open(base_path, "wb") as f:
        for i in range(n):
            f.write(int(lens[i]).to_bytes(4, "big") + ids[i].tobytes())
    with open(base_path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": int(n), "embedding_size": int(L)}, f)


# ---------------------------------------------------------------------------- block files
def write_embedding_block(out_dir, block_id, emb, ids):
    """gen_doc_embeddings.py:129-135 — pickle protocol 4 of two ndarrays."""
    with open(os.path.join(out_dir, f"passage_emb_block_{block_id}.pb"), "wb") as h:
        pickle.dump(np.ascontiguousarray(emb, np.float32), h, protocol=4)
    with open(os.path.join(out_dir, f"passage_embid_block_{block_id}.pb"), "wb") as h:
        pickle.dump(np.ascontiguousarray(ids, np.int64), h, protocol=4)


def _map_pickled_ndarray(path):
    """Zero-copy view of a pickle-4 C-contiguous ndarray.  numpy pickles an array as
    ``_reconstruct`` + ``(1, shape, dtype, fortran_order, <raw bytes>)``; the raw bytes are one
    BINBYTES8/BINBYTES opcode.  The opcode stream is walked WITHOUT materialising the payload
    (pickletools.genops would read it), shape/dtype are taken from the small opcodes around it,
    and the payload is memory-mapped in place.  Anything unexpected falls back to pickle.load."""
    import struct
    try:
        with open(path, "rb") as f:
            head = f.read(1 << 16)
        # locate the first large bytes opcode in the header region
        pos = blob_pos = blob_len = None
        for opc, hdr, fmt in ((b"\x8e", 9, "<Q"), (b"B", 5, "<I")):
            i = head.find(opc)
            while i != -1:
                n = struct.unpack_from(fmt, head, i + 1)[0]
                if n > 64 and i + hdr + n <= os.path.getsize(path):
                    cand = (i, i + hdr, n)
                    if pos is None or cand[0] < pos:
                        pos, blob_pos, blob_len = cand
                    break
                i = head.find(opc, i + 1)
        if blob_pos is None:
            raise ValueError("no bytes payload found")
        # rebuild the (tiny) pickle with an EMPTY payload to get shape/dtype through numpy itself
        tail_start = blob_pos + blob_len
        with open(path, "rb") as f:
            f.seek(tail_start)
            tail = f.read()
        stub = head[:pos] + b"C\x00" + tail          # SHORT_BINBYTES of length 0
        shape = dtype = None
        for op, arg, _ in pickletools.genops(stub):
            pass                                      # validates the opcode stream
        # shape and dtype: parse from numpy's reduce args via a restricted unpickle of the stub
        class _Probe:
            def __init__(self):
                self.state = None
            def __setstate__(self, st):
                self.state = st
        class _U(pickle.Unpickler):
            def find_class(self, module, name):
                if name == "_reconstruct":
                    return lambda *a: _Probe()
                return super().find_class(module, name)
        import io
        probe = _U(io.BytesIO(stub)).load()
        _, shape, dtype, fortran, _ = probe.state
        if fortran or int(np.prod(shape)) * dtype.itemsize != blob_len:
            raise ValueError("unexpected array state")
        return np.memmap(path, dtype=dtype, mode="r", offset=blob_pos, shape=tuple(shape))
    except Exception:
        with open(path, "rb") as f:
            return pickle.load(f)


def read_embedding_block(emb_dir, block_id, mmap=False):
    """(emb float32 [n,768], ids int64 [n]) of one block (reader side: test_HAConvDR_topiocqa.py:82-93)."""
    pe = os.path.join(emb_dir, f"passage_emb_block_{block_id}.pb")
    pi = os.path.join(emb_dir, f"passage_embid_block_{block_id}.pb")
    if mmap:
        emb = _map_pickled_ndarray(pe)
    else:
        with open(pe, "rb") as h:
            emb = pickle.load(h)
    with open(pi, "rb") as h:
        ids = pickle.load(h)
    return emb, ids


# ---------------------------------------------------------------------------- the encode loop
def encode_passages(model, passages, out_dir, per_gpu_eval_batch_size=250, n_gpu=1, rank=0, world_size=1,
                    expect_per_block_passage_num=2_500_000, log_every=0):
    """Mirror of InferenceEmbeddingFromStreamDataLoader (gen_doc_embeddings.py:65-158).

    model: ``ANCEEncoder`` (or anything callable as model(ids, mask) -> [B,768] CUDA tensor).
    Writes the block files this rank owns (block b belongs to rank b % world_size) and returns the
    number of passages written by this rank."""
    import torch
    batch = max(1, n_gpu) * per_gpu_eval_batch_size                    # :73
    block_batches = max(1, expect_per_block_passage_num // batch)      # :88
    block_rows = block_batches * batch
    n = len(passages)
    n_blocks = (n + block_rows - 1) // block_rows
    written = 0
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device("cuda", getattr(model, "device", 0))
    for b in range(rank, n_blocks, world_size):
        lo, hi = b * block_rows, min(n, (b + 1) * block_rows)
        outs = []
        for s in range(lo, hi, batch):
            e = min(hi, s + batch)
            ids, lens = passages.batch(s, e)
            # (the reader hands out slices of a READ-ONLY mmap: torch.from_numpy of those warns, and a write through such a tensor
            # is undefined behaviour -- the batch is copied once, into writable memory, before torch sees it)
            ids = np.require(ids, requirements=["C", "W"])
            lens = np.require(lens, requirements=["C", "W"])
            ids_t = torch.from_numpy(ids).to(dev, non_blocking=True)
            # attention_mask = [1]*passage_len + [0]*pad_len  (gen_doc_embeddings.py:38-40)
            mask_t = (torch.arange(ids.shape[1], device=dev)[None, :] < torch.from_numpy(lens).to(dev)[:, None]).to(torch.int32)
            outs.append(model(ids_t, mask_t))                          # :106-110, stays on the GPU
        emb = torch.cat(outs).cpu().numpy()                            # one D2H per block (the reference syncs per batch, :112)
        bad = np.flatnonzero(np.isnan(emb).any(axis=1))
        if bad.size:   # the device path flags a sequence it cannot encode with a NaN row (include/haconvdr.h)
            raise ValueError(f"passage {lo + int(bad[0])}: empty record or token id outside the vocabulary ({bad.size} bad in block {b})")
        write_embedding_block(out_dir, b, emb, np.arange(lo, hi, dtype=np.int64))   # ids = enumerate index (utils.py:140-143)
        written += hi - lo
        if log_every:
            print(f"rank {rank}: wrote block {b} ({hi - lo} passages)")
    return written


def load_model(model_type, model_path, device=0):
    """Mirror of load_model(model_type, model_path) (src/models.py:112-140) for the ANCE types: returns
    ``(tokenizer, model)``.  The tokenizer is the checkpoint's RobertaTokenizer when its vocabulary files are in the
    directory, else None (the passage pipeline never uses it: it reads pre-tokenized records)."""
    from .encoder import ANCEEncoder
    if model_type not in ("ANCE_Query", "ANCE_Passage"):
        raise ValueError("{} is not supported by the MI355X path (ANCE_Query / ANCE_Passage only)".format(model_type))
    tokenizer = None
    if os.path.exists(os.path.join(model_path, "vocab.json")) and os.path.exists(os.path.join(model_path, "merges.txt")):
        from transformers import RobertaTokenizer
        tokenizer = RobertaTokenizer.from_pretrained(model_path, do_lower_case=True)
    return tokenizer, ANCEEncoder.from_pretrained(model_path, device=device)


def generate_new_ann(args):
    """Mirror of generate_new_ann(args) (gen_doc_embeddings.py:190-212): ``args`` carries the keys of
    Config/gen_doc_embeddings.toml (model_type, pretrained_passage_encoder, per_gpu_eval_batch_size, n_gpu,
    tokenized_passage_collection_dir_path, data_output_path).  The reference wraps the model in nn.DataParallel over
    ``n_gpu`` devices of one process; here every rank of a torch.distributed launch (RANK / WORLD_SIZE / LOCAL_RANK)
    encodes the blocks it owns on its own GPU — ``n_gpu`` only sizes the batches and blocks, as there (:73, :88)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    _, model = load_model(getattr(args, "model_type", "ANCE") + "_Passage", args.pretrained_passage_encoder, device=local)
    passages = TokenizedPassages(os.path.join(args.tokenized_passage_collection_dir_path, "passages"))
    return encode_passages(model, passages, args.data_output_path, args.per_gpu_eval_batch_size, getattr(args, "n_gpu", 1), rank, world)
