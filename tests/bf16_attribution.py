#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (a script, not a test; CPU only): WHICH bf16 rounding of the encoder path costs the accuracy that
tools/parity_survey.py sweep measures?  The fp32 restatement of ANCE (oracle/ance_oracle.py's arithmetic, src/models.py:39-64) is run
with bf16 rounding injected at one place at a time -- the operands of the score product (Q, K), the operands of P.V, the operands of
the four projection GEMMs and the FFN (activations and weights), the residual stream between layers -- and everything together, on
content-sensitive synthetic weights of growing layer-matrix standard deviation; the figure is max 1-cos against the unrounded fp32 run.
  python tests/bf16_attribution.py [n_seq] [L]"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from haconvdr_amd import synth  # noqa: E402


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def forward(sd, ids, mask, where, n_layers=12, n_heads=12, eps=1e-5, pad_id=1):
    """where: set of {"qk", "pv", "gemm", "resid"}."""
    t = lambda k: torch.from_numpy(np.asarray(sd[k], np.float32))   # noqa: E731
    ids = torch.as_tensor(np.asarray(ids), dtype=torch.long)
    mask = torch.as_tensor(np.asarray(mask), dtype=torch.long)
    B, L = ids.shape
    p = "roberta.embeddings."
    nonpad = (ids != pad_id).long()
    pos = torch.cumsum(nonpad, 1) * nonpad + pad_id
    x = t(p + "word_embeddings.weight")[ids] + t(p + "position_embeddings.weight")[pos] + t(p + "token_type_embeddings.weight")[0]
    H = x.shape[-1]
    x = F.layer_norm(x, (H,), t(p + "LayerNorm.weight"), t(p + "LayerNorm.bias"), eps)
    dh = H // n_heads
    add_mask = (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    g = (lambda v: bf(v)) if "gemm" in where else (lambda v: v)
    for i in range(n_layers):
        q = f"roberta.encoder.layer.{i}."

        def lin(name, v):
            return F.linear(g(v), g(t(q + name + ".weight")), t(q + name + ".bias"))

        def heads(v):
            return v.view(B, L, n_heads, dh).transpose(1, 2)
        Q, K, V = heads(lin("attention.self.query", x)), heads(lin("attention.self.key", x)), heads(lin("attention.self.value", x))
        if "qk" in where:
            Q, K = bf(Q / math.sqrt(dh)) * math.sqrt(dh), bf(K)      # (the kernels round Q after its scale)
        s = Q @ K.transpose(-1, -2) / math.sqrt(dh) + add_mask
        P = torch.softmax(s, -1)
        if "pv" in where:
            P, V = bf(P), bf(V)
        ctx = (P @ V).transpose(1, 2).reshape(B, L, H)
        x = F.layer_norm(x + lin("attention.output.dense", ctx), (H,), t(q + "attention.output.LayerNorm.weight"), t(q + "attention.output.LayerNorm.bias"), eps)
        if "resid" in where:
            x = bf(x)
        h = F.gelu(lin("intermediate.dense", x))
        x = F.layer_norm(x + lin("output.dense", h), (H,), t(q + "output.LayerNorm.weight"), t(q + "output.LayerNorm.bias"), eps)
        if "resid" in where:
            x = bf(x)
    e = F.linear(x[:, 0], t("embeddingHead.weight"), t("embeddingHead.bias"))
    return F.layer_norm(e, (e.shape[-1],), t("norm.weight"), t("norm.bias"), 1e-5).numpy()


def omc(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float((1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))).max())


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    ids, lens = synth.token_batch(0x5EED, n_seq, L, min_len=L // 4)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    cases = [("scores: Q, K", {"qk"}), ("P.V: P, V", {"pv"}), ("projection / FFN operands", {"gemm"}), ("residual stream", {"resid"}),
             ("all four", {"qk", "pv", "gemm", "resid"})]
    print(f"{n_seq} sequences x {L} tokens (lens {lens.min()}..{lens.max()}), 12 layers; max 1-cos against the fp32 run", flush=True)
    print("layer-matrix std | " + " | ".join(n for n, _ in cases) + "", flush=True)
    with torch.no_grad():
        for std in (0.08, 0.10, 0.12, 0.16):
            sd = synth.ance_state_dict(0x0D17, 12, layer_matrix_std=std)
            ref = forward(sd, ids, mask, set())
            row = [f"{omc(forward(sd, ids, mask, w), ref):.2e}" for _, w in cases]
            print(f"{std:.2f} | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
