"""Encoder seam of the reference, host-side mirror.

``ANCEEncoder`` is called exactly like the reference's module: ``model(input_ids,
attention_mask)`` with integer tensors [B, L] (src/test_HAConvDR_topiocqa.py:211,
gen_doc_embeddings.py:110) and returns a float32 tensor [B, 768] on the same device
(= src/models.py:44).  Weights use the checkpoint's own key names (``roberta.*``,
``embeddingHead.*``, ``norm.*``; ``classifier.*`` is ignored as in the reference's forward).
All arithmetic runs in the gfx950 HIP kernels of libhaconvdr.so; there is no CPU fallback.
"""
import ctypes

import numpy as np

from . import _lib


class ANCEEncoder:
    def __init__(self, n_layers=12, vocab=50265, max_pos=514, type_vocab=1, pad_token_id=1, ln_eps=1e-5, device=0):
        self.device = int(device)
        self.n_layers = int(n_layers)
        cfg = _lib.EncoderConfig(self.n_layers, 768, 12, 3072, int(vocab), int(max_pos), int(type_vocab), int(pad_token_id), float(ln_eps))
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().hac_encoder_create(ctypes.byref(cfg), self.device, ctypes.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().hac_encoder_destroy(h)
            except Exception:
                pass

    # ---- weights -----------------------------------------------------------
    def load_state_dict(self, sd):
        """sd: name -> float32 array / torch tensor with the reference checkpoint's names."""
        L = _lib.lib()
        for name, v in sd.items():
            if name.startswith("classifier.") or name.endswith("position_ids"):
                continue                       # present in the checkpoint, unused by ANCE.forward (models.py:26)
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            _lib.check(L.hac_encoder_set_weight(self._h, name.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.size))
        _lib.check(L.hac_encoder_finalize(self._h))
        return self

    @classmethod
    def from_state_dict(cls, sd, device=0, **kw):
        n_layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("roberta.encoder.layer."))
        vocab, _ = np.shape(sd["roberta.embeddings.word_embeddings.weight"])
        max_pos, _ = np.shape(sd["roberta.embeddings.position_embeddings.weight"])
        tv, _ = np.shape(sd["roberta.embeddings.token_type_embeddings.weight"])
        return cls(n_layers=n_layers, vocab=vocab, max_pos=max_pos, type_vocab=tv, device=device, **kw).load_state_dict(sd)

    @classmethod
    def from_pretrained(cls, path, device=0):
        """Checkpoint directory as ``ANCE.from_pretrained(model_path, config=RobertaConfig.from_pretrained(model_path))``
        reads it (src/models.py:113-122): ``config.json`` + ``pytorch_model.bin`` or ``model.safetensors``.

        config.json supplies what the tensors cannot: ``layer_norm_eps`` and ``pad_token_id`` (position ids), and is
        checked against them — layer count, hidden / FFN / head geometry, vocabulary, positions, activation.  A
        checkpoint this encoder was not built for is refused here, not mis-encoded."""
        import json
        import os
        import torch
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        sd = {k: v.float() for k, v in sd.items()}
        kw = {}
        cfg_path = os.path.join(path, "config.json")
        if os.path.exists(cfg_path):
            with open(cfg_path) as f:
                cfg = json.load(f)
            n_layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("roberta.encoder.layer."))
            vocab, hidden = sd["roberta.embeddings.word_embeddings.weight"].shape
            found = {"num_hidden_layers": n_layers, "hidden_size": hidden, "vocab_size": vocab,
                     "max_position_embeddings": sd["roberta.embeddings.position_embeddings.weight"].shape[0],
                     "type_vocab_size": sd["roberta.embeddings.token_type_embeddings.weight"].shape[0],
                     "intermediate_size": sd["roberta.encoder.layer.0.intermediate.dense.weight"].shape[0]}
            for key, have in found.items():
                if key in cfg and int(cfg[key]) != int(have):
                    raise ValueError(f"{cfg_path}: {key} = {cfg[key]} but the checkpoint's tensors say {have}")
            if int(cfg.get("num_attention_heads", 12)) != 12 or int(cfg.get("hidden_size", 768)) != 768 or int(cfg.get("intermediate_size", 3072)) != 3072:
                raise ValueError(f"{cfg_path}: only the RoBERTa-base geometry of ANCE is built (hidden 768, 12 heads, FFN 3072)")
            if cfg.get("hidden_act", "gelu") != "gelu":
                raise ValueError(f"{cfg_path}: hidden_act = {cfg['hidden_act']!r}; the kernels implement erf GELU")
            if cfg.get("position_embedding_type", "absolute") != "absolute":
                raise ValueError(f"{cfg_path}: position_embedding_type = {cfg['position_embedding_type']!r} is not supported")
            kw = {"ln_eps": float(cfg.get("layer_norm_eps", 1e-5)), "pad_token_id": int(cfg.get("pad_token_id", 1))}
        return cls.from_state_dict(sd, device=device, **kw)

    # ---- forward -----------------------------------------------------------
    def __call__(self, input_ids, attention_mask, wrap_pooler=False):
        """model(input_ids, attention_mask) -> float32 [B, 768].  torch CUDA tensors stay on the GPU
        (enqueued on the current stream); numpy / CPU inputs take the synchronous host path."""
        import torch
        if isinstance(input_ids, torch.Tensor) and input_ids.is_cuda:
            ids = input_ids.contiguous()
            mask = attention_mask.to(ids.dtype).contiguous()
            if ids.dtype not in (torch.int32, torch.int64):
                ids, mask = ids.long(), mask.long()
            B, L = ids.shape
            out = torch.empty((B, 768), dtype=torch.float32, device=ids.device)
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().hac_encoder_forward_device(self._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(mask.data_ptr()),
                                                            ids.element_size(), B, L, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st)))
            return out
        ids = np.ascontiguousarray(np.asarray(input_ids), dtype=np.int32)
        mask = np.ascontiguousarray(np.asarray(attention_mask), dtype=np.int32)
        B, L = ids.shape
        out = np.empty((B, 768), np.float32)
        i32p = ctypes.POINTER(ctypes.c_int32)
        _lib.check(_lib.lib().hac_encoder_forward(self._h, ids.ctypes.data_as(i32p), mask.ctypes.data_as(i32p), B, L,
                                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return torch.from_numpy(out) if isinstance(input_ids, torch.Tensor) else out

    forward = query_emb = doc_emb = __call__       # models.py:39-49,63-64

    def eval(self):
        return self

    def to(self, device):
        return self

    def set_option(self, name, value):
        """Tuning / test switch of this handle (include/haconvdr.h: hac_encoder_set_option), e.g. ("gemm", "classic")."""
        _lib.check(_lib.lib().hac_encoder_set_option(self._h, str(name).encode(), str(value).encode()))

    def last_plan(self):
        """Kernel families of the most recent forward: "gemm=gemm8|classic256|classic128 attn=... sub_batches=N rows=R"."""
        return _lib.lib().hac_encoder_last_plan(self._h).decode()

    KERNEL_CLASSES = ("qkv", "attention", "out_proj", "ffn_up", "ffn_down", "layernorm")   # HAC_ENC_CLASS_* of include/haconvdr.h

    def set_profiling(self, on=True, classes=()):
        """hipEvent pairs around the layer stack of every forward (``on``) and around every launch of the named
        kernel classes (``classes``: names from KERNEL_CLASSES, or "all")."""
        if classes == "all":
            classes = self.KERNEL_CLASSES
        mask = int(bool(on))
        for c in classes:
            mask |= 2 << self.KERNEL_CLASSES.index(c)
        _lib.check(_lib.lib().hac_encoder_set_profiling(self._h, mask))

    def profile_drain(self, cap=4096):
        buf = (ctypes.c_float * cap)()
        n = ctypes.c_int()
        _lib.check(_lib.lib().hac_encoder_profile_drain(self._h, buf, cap, ctypes.byref(n)))
        return [float(buf[i]) for i in range(n.value)]

    def last_clock_mhz(self):
        """(shader clock in MHz, seconds) inside the most recent large-batch FFN-up launch made with class profiling on, or
        None (hac_encoder_last_clock): what fractions of a peak are normalised by across boxes that hold different clocks."""
        out = (ctypes.c_uint64 * 2)()
        _lib.check(_lib.lib().hac_encoder_last_clock(self._h, out))
        if not out[1]:
            return None
        return 100.0 * int(out[0]) / int(out[1]), int(out[1]) / 100e6

    def attention_redo(self):
        """How often the woven attention kernel handed an item to its fix-up pass in the most recent forward (upper bound of
        the items, 0 iff none; hac_encoder_attention_redo).  Test aid; waits for the device."""
        n = ctypes.c_longlong()
        _lib.check(_lib.lib().hac_encoder_attention_redo(self._h, ctypes.byref(n)))
        return int(n.value)

    def profile_drain_class(self, name, cap=16384):
        """Durations (ms, launch order) of the launches of one kernel class since the last drain."""
        buf = (ctypes.c_float * cap)()
        n = ctypes.c_int()
        _lib.check(_lib.lib().hac_encoder_profile_drain_class(self._h, self.KERNEL_CLASSES.index(name), buf, cap, ctypes.byref(n)))
        return [float(buf[i]) for i in range(n.value)]
