"""ctypes binding of haconvdr_amd/csrc/libhaconvdr.so (C ABI: include/haconvdr.h).

There is no CPU fallback.  If the HIP library has not been built, or no MI355X is
visible, the first call raises — loudly — instead of computing anything on the host.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HAC_LIBRARY_PATH: development only (A/B runs of kernel variants built beside the tree); the product is the in-tree library
LIB_PATH = os.environ.get("HAC_LIBRARY_PATH") or os.path.join(_HERE, "csrc", "libhaconvdr.so")
_LIB = None

HAC_MAX_K = 2048
HAC_ERR_INTERNAL = 5   # include/haconvdr.h: the device detected a broken invariant of the library (never a hang, never wrong bits)


class HacError(RuntimeError):
    """A C-ABI call returned a non-zero hac_status."""

    def __init__(self, code, msg):
        super().__init__(f"haconvdr error {code}: {msg}")
        self.code = code


def _declare(L):
    c_f32p = ctypes.POINTER(ctypes.c_float)
    vp = ctypes.c_void_p
    i64 = ctypes.c_int64
    L.hac_last_error.restype = ctypes.c_char_p
    L.hac_version.restype = ctypes.c_char_p
    L.hac_index_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(vp)]
    L.hac_index_destroy.argtypes = [vp]
    L.hac_index_destroy.restype = None
    L.hac_index_add.argtypes = [vp, c_f32p, i64]
    L.hac_index_add_device.argtypes = [vp, vp, i64, vp]
    L.hac_index_search.argtypes = [vp, c_f32p, i64, ctypes.c_int, c_f32p, ctypes.POINTER(i64)]
    L.hac_index_search_device.argtypes = [vp, vp, i64, ctypes.c_int, vp, vp, vp, vp]
    L.hac_index_search_keys_device.argtypes = [vp, vp, i64, ctypes.c_int, vp, ctypes.c_uint32, vp]
    L.hac_index_reset.argtypes = [vp]
    L.hac_index_ntotal.argtypes = [vp]
    L.hac_index_ntotal.restype = i64
    L.hac_index_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    L.hac_index_set_profiling.argtypes = [vp, ctypes.c_int]
    L.hac_index_profile_drain.argtypes = [vp, c_f32p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.hac_index_last_plan.argtypes = [vp]
    L.hac_index_last_status.argtypes = [vp]
    L.hac_index_last_status.restype = ctypes.c_int
    L.hac_index_last_plan.restype = ctypes.c_char_p
    L.hac_merge_keys_device.argtypes = [ctypes.c_int, vp, ctypes.c_int, i64, ctypes.c_int, vp, vp]
    L.hac_keys_to_results_device.argtypes = [ctypes.c_int, vp, i64, vp, vp, vp, vp]
    L.hac_encoder_create.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp)]
    L.hac_encoder_destroy.argtypes = [vp]
    L.hac_encoder_destroy.restype = None
    L.hac_encoder_set_weight.argtypes = [vp, ctypes.c_char_p, c_f32p, ctypes.c_size_t]
    L.hac_encoder_finalize.argtypes = [vp]
    L.hac_encoder_forward.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int, c_f32p]
    L.hac_encoder_forward_device.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    L.hac_encoder_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    L.hac_encoder_last_plan.argtypes = [vp]
    L.hac_encoder_last_plan.restype = ctypes.c_char_p
    L.hac_encoder_set_profiling.argtypes = [vp, ctypes.c_int]
    L.hac_encoder_profile_drain.argtypes = [vp, c_f32p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.hac_encoder_profile_drain_class.argtypes = [vp, ctypes.c_int, c_f32p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.hac_encoder_last_clock.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.hac_encoder_attention_redo.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    for name in ("hac_encoder_create", "hac_encoder_set_weight", "hac_encoder_finalize", "hac_encoder_forward",
                 "hac_encoder_forward_device", "hac_encoder_set_option", "hac_encoder_set_profiling", "hac_encoder_profile_drain",
                 "hac_encoder_profile_drain_class", "hac_encoder_last_clock", "hac_encoder_attention_redo"):
        getattr(L, name).restype = ctypes.c_int
    for name in ("hac_index_create", "hac_index_add", "hac_index_add_device", "hac_index_search",
                 "hac_index_search_device", "hac_index_search_keys_device", "hac_index_reset",
                 "hac_index_set_option", "hac_index_set_profiling", "hac_index_profile_drain", "hac_merge_keys_device",
                 "hac_keys_to_results_device"):
        getattr(L, name).restype = ctypes.c_int


# every symbol include/haconvdr.h declares (checked by tests/test_cabi_symbols.py)
EXPORTED_SYMBOLS = (
    "hac_last_error", "hac_version", "hac_index_create", "hac_index_destroy", "hac_index_add",
    "hac_index_add_device", "hac_index_search", "hac_index_search_device", "hac_index_search_keys_device",
    "hac_index_reset", "hac_index_ntotal", "hac_index_set_option", "hac_index_set_profiling", "hac_index_profile_drain", "hac_index_last_plan",
    "hac_index_last_status", "hac_merge_keys_device", "hac_keys_to_results_device",
    "hac_encoder_create", "hac_encoder_destroy", "hac_encoder_set_weight", "hac_encoder_finalize", "hac_encoder_forward",
    "hac_encoder_forward_device", "hac_encoder_set_option", "hac_encoder_last_plan", "hac_encoder_set_profiling", "hac_encoder_profile_drain",
    "hac_encoder_profile_drain_class", "hac_encoder_last_clock", "hac_encoder_attention_redo",
)


class EncoderConfig(ctypes.Structure):
    _fields_ = [("n_layers", ctypes.c_int), ("hidden", ctypes.c_int), ("n_heads", ctypes.c_int), ("ffn", ctypes.c_int),
                ("vocab", ctypes.c_int), ("max_pos", ctypes.c_int), ("type_vocab", ctypes.c_int), ("pad_token_id", ctypes.c_int),
                ("ln_eps", ctypes.c_float)]


def lib():
    """The loaded library; raises if it was never built (run __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C haconvdr_amd/csrc).  haconvdr_amd has no CPU fallback.")
        # torch bundles its own libamdhip64.so.7; it must be the one HIP runtime of the process,
        # so it is loaded first and libhaconvdr.so binds to it by soname (two runtimes in one
        # process leave the second without devices).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        _declare(L)
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise HacError(rc, lib().hac_last_error().decode("utf-8", "replace"))
