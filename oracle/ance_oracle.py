"""ORACLE — TEST INFRASTRUCTURE ONLY.

fp32 CPU restatement (torch ops) of the reference's encoder path:
  ANCE.forward / query_emb / masked_mean_or_first   /root/reference/src/models.py:39-64
  RobertaModel forward (third-party: transformers, pinned 4.2.0 at README.md:11; 5.15.0 is
  what is installed in the authoring container and what the goldens were produced with).
The arithmetic restated is the published RoBERTa-base forward (post-LN BERT block):
  emb = LN(word[id] + pos[cumsum(id != 1)*(id != 1) + 1] + type[0])
  x   = LN(x + Wo . softmax(Q K^T / sqrt(64) + mask) V)        (12 heads x 64)
  x   = LN(x + W2 . gelu_erf(W1 . x))
  out = LN_768(embeddingHead(x[:, 0]))                           (use_mean = False, models.py:30,:56)
Pinned against the reference itself: tests/golden/encoder_*.npz hold outputs of the reference's
``models.ANCE`` (run by tests/golden/make_golden_encoder.py) on seeded synthetic weights/inputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, name):
    v = sd[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def ance_forward(sd, input_ids, attention_mask, n_layers=None, n_heads=12, eps=1e-5, pad_id=1):
    """sd: name -> float32 array (reference checkpoint names).  input_ids/attention_mask: int [B, L].
    Returns float32 ndarray [B, 768] (= the reference's model(input_ids, attention_mask))."""
    ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long)
    mask = torch.as_tensor(np.asarray(attention_mask), dtype=torch.long)
    B, L = ids.shape
    if n_layers is None:
        n_layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("roberta.encoder.layer."))
    p = "roberta.embeddings."
    nonpad = (ids != pad_id).long()
    pos = torch.cumsum(nonpad, 1) * nonpad + pad_id                     # HF create_position_ids_from_input_ids
    x = _t(sd, p + "word_embeddings.weight")[ids] + _t(sd, p + "position_embeddings.weight")[pos] \
        + _t(sd, p + "token_type_embeddings.weight")[0]
    H = x.shape[-1]
    x = F.layer_norm(x, (H,), _t(sd, p + "LayerNorm.weight"), _t(sd, p + "LayerNorm.bias"), eps)
    dh = H // n_heads
    add_mask = (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    for i in range(n_layers):
        q = f"roberta.encoder.layer.{i}."

        def lin(name, t):
            return F.linear(t, _t(sd, q + name + ".weight"), _t(sd, q + name + ".bias"))

        def heads(t):
            return t.view(B, L, n_heads, dh).transpose(1, 2)
        Q, K, V = heads(lin("attention.self.query", x)), heads(lin("attention.self.key", x)), heads(lin("attention.self.value", x))
        s = Q @ K.transpose(-1, -2) / math.sqrt(dh) + add_mask
        ctx = (torch.softmax(s, -1) @ V).transpose(1, 2).reshape(B, L, H)
        x = F.layer_norm(x + lin("attention.output.dense", ctx), (H,), _t(sd, q + "attention.output.LayerNorm.weight"),
                         _t(sd, q + "attention.output.LayerNorm.bias"), eps)
        h = F.gelu(lin("intermediate.dense", x))                         # exact erf GELU (hidden_act = "gelu")
        x = F.layer_norm(x + lin("output.dense", h), (H,), _t(sd, q + "output.LayerNorm.weight"),
                         _t(sd, q + "output.LayerNorm.bias"), eps)
    cls = x[:, 0]                                                        # masked_mean_or_first, use_mean=False
    e = F.linear(cls, _t(sd, "embeddingHead.weight"), _t(sd, "embeddingHead.bias"))
    out = F.layer_norm(e, (e.shape[-1],), _t(sd, "norm.weight"), _t(sd, "norm.bias"), 1e-5)
    return out.numpy().astype(np.float32)
