"""Index seam of the reference, host-side mirror.

``build_index(args)`` stands where ``build_faiss_index(args)`` does in
src/test_HAConvDR_topiocqa.py:39-71 and returns an object with the three calls
``search_one_by_one_with_faiss`` makes on it (:98 add, :102 search, :122 reset),
with the same argument meaning, result dtypes and error behaviour (exceptions
propagate).  All arithmetic runs in the gfx950 HIP kernels of libhaconvdr.so.
"""
import ctypes

import numpy as np

from . import _lib


def _stream_ptr(stream):
    if stream is None:
        import torch
        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(getattr(stream, "cuda_stream", stream))


def _keep_alive(t, stream):
    """The kernels read ``t`` asynchronously on ``stream`` (a torch stream, a raw hipStream_t or None = the
    current stream): tell the caching allocator, or it may hand the memory out again too early."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream()
    elif not hasattr(stream, "cuda_stream"):
        stream = torch.cuda.ExternalStream(int(stream))
    t.record_stream(stream)


class FlatIPIndex:
    """faiss.IndexFlatIP(d) drop-in: exact fp32 inner product, results ordered by
    (score desc, row asc), padded with -FLT_MAX / -1.

    devices: HIP ordinals.  One device is the normal case (one process per GPU);
    several reproduce faiss's in-process ``shard=True`` clone (:55-66)."""

    def __init__(self, d=768, devices=(0,)):
        self.d = int(d)
        self.devices = tuple(int(x) for x in devices)
        self._h = ctypes.c_void_p()
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        _lib.check(_lib.lib().hac_index_create(self.d, arr, len(self.devices), ctypes.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().hac_index_destroy(h)
            except Exception:
                pass

    # ---- the three calls of the reference ---------------------------------
    def add(self, x):
        """index.add(passage_embedding): float32 [n, d]; copied before return."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add expects [n, {self.d}] float32, got {x.shape}")
        _lib.check(_lib.lib().hac_index_add(self._h, x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), x.shape[0]))

    def search(self, q, k):
        """D, I = index.search(query_embeddings, topN) -> float32 [nq,k], int64 [nq,k]."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"search expects [nq, {self.d}] float32, got {q.shape}")
        k = int(k)
        D = np.empty((q.shape[0], k), np.float32)
        I = np.empty((q.shape[0], k), np.int64)
        _lib.check(_lib.lib().hac_index_search(self._h, q.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), q.shape[0], k,
                                               D.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                               I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return D, I

    def reset(self):
        _lib.check(_lib.lib().hac_index_reset(self._h))

    @property
    def ntotal(self):
        return int(_lib.lib().hac_index_ntotal(self._h))

    # ---- device-resident variants (torch tensors on the index's GPU) -------
    def add_tensor(self, x, stream=None):
        """x: torch float32 CUDA tensor [n, d], contiguous.  Enqueued on ``stream`` (default: torch's current
        stream).  The host API (``add`` / ``search``) runs on the index's own stream: synchronize the stream used
        here before mixing in host-API calls."""
        assert x.is_cuda and x.dtype.is_floating_point and x.dim() == 2 and x.shape[1] == self.d
        x = x.contiguous().float()
        _lib.check(_lib.lib().hac_index_add_device(self._h, ctypes.c_void_p(x.data_ptr()), x.shape[0], _stream_ptr(stream)))
        _keep_alive(x, stream)

    def search_tensor(self, q, k, id_map=None, stream=None):
        """q: torch float32 CUDA [nq, d] -> (D float32 [nq,k], I int64 [nq,k]) CUDA tensors,
        enqueued on the current stream (no host sync).  id_map: optional int64 CUDA
        tensor [ntotal], fuses ``passage_embedding2id[I]`` (:110)."""
        import torch
        assert q.is_cuda and q.dim() == 2 and q.shape[1] == self.d
        q = q.contiguous().float()
        D = torch.empty((q.shape[0], k), dtype=torch.float32, device=q.device)
        I = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
        mp = ctypes.c_void_p(id_map.data_ptr()) if id_map is not None else ctypes.c_void_p()
        _lib.check(_lib.lib().hac_index_search_device(self._h, ctypes.c_void_p(q.data_ptr()), q.shape[0], int(k),
                                                      ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()), mp,
                                                      _stream_ptr(stream)))
        for t in (q, D, I):
            _keep_alive(t, stream)
        return D, I

    def search_keys_tensor(self, q, k, pos_base=0, stream=None):
        """Packed top-k keys (int64 view of uint64) [nq, k]; the unit exchanged between shards."""
        import torch
        q = q.contiguous().float()
        keys = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
        _lib.check(_lib.lib().hac_index_search_keys_device(self._h, ctypes.c_void_p(q.data_ptr()), q.shape[0], int(k),
                                                           ctypes.c_void_p(keys.data_ptr()), int(pos_base),
                                                           _stream_ptr(stream)))
        for t in (q, keys):
            _keep_alive(t, stream)
        return keys

    def set_option(self, name, value):
        """Tuning / test switch of this handle (include/haconvdr.h: hac_index_set_option), e.g.
        ``set_option("split", "0")`` = exact fp32 kernels only."""
        _lib.check(_lib.lib().hac_index_set_option(self._h, str(name).encode(), str(value).encode()))

    def set_profiling(self, on=True):
        _lib.check(_lib.lib().hac_index_set_profiling(self._h, int(bool(on))))

    def last_plan(self):
        return _lib.lib().hac_index_last_plan(self._h).decode()

    def check_status(self):
        """Raise HacError (code HAC_ERR_INTERNAL) if a scan of a search enqueued so far gave up at its pass bound
        (hac_index_last_status).  The host API reports this itself; after ``search_tensor`` / ``search_keys_tensor``
        synchronize the stream you searched on, then call this.  Reading clears the error word."""
        _lib.check(_lib.lib().hac_index_last_status(self._h))

    def profile_drain(self, cap=4096):
        """Durations (ms) of the main scan kernel of every search since the last drain."""
        buf = (ctypes.c_float * cap)()
        n = ctypes.c_int()
        _lib.check(_lib.lib().hac_index_profile_drain(self._h, buf, cap, ctypes.byref(n)))
        return [float(buf[i]) for i in range(n.value)]


def merge_keys(lists, stream=None):
    """lists: int64(uint64) CUDA tensor [L, nq, k] of per-shard sorted keys -> [nq, k]."""
    import torch
    L, nq, k = lists.shape
    lists = lists.contiguous()
    out = torch.empty((nq, k), dtype=torch.int64, device=lists.device)
    _lib.check(_lib.lib().hac_merge_keys_device(lists.device.index or 0, ctypes.c_void_p(lists.data_ptr()), L, nq, k,
                                                ctypes.c_void_p(out.data_ptr()), _stream_ptr(stream)))
    return out


def keys_to_results(keys, id_map=None, stream=None):
    """keys [nq,k] -> (D float32, I int64); id_map optional int64 CUDA tensor (position -> id)."""
    import torch
    keys = keys.contiguous()
    D = torch.empty(keys.shape, dtype=torch.float32, device=keys.device)
    I = torch.empty(keys.shape, dtype=torch.int64, device=keys.device)
    mp = ctypes.c_void_p(id_map.data_ptr()) if id_map is not None else ctypes.c_void_p()
    _lib.check(_lib.lib().hac_keys_to_results_device(keys.device.index or 0, ctypes.c_void_p(keys.data_ptr()), keys.numel(), mp,
                                                     ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                                     _stream_ptr(stream)))
    return D, I


def build_index(args):
    """Mirror of build_faiss_index(args) (src/test_HAConvDR_topiocqa.py:39-71).

    ``args.n_gpu`` devices hold contiguous shards of every added block; the reference's
    CPU branch (``use_gpu`` false, :68-69) has no counterpart here — this package is
    the GPU path and refuses to run without one."""
    n_gpu = max(1, int(getattr(args, "n_gpu", 1)))
    return FlatIPIndex(768, devices=tuple(range(n_gpu)))
