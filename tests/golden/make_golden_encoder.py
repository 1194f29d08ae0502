#!/usr/bin/env python3
"""Generate tests/golden/encoder_*.npz by RUNNING THE REFERENCE's own ``models.ANCE``
(/root/reference/src/models.py:20-64) in the authoring container on seeded synthetic weights
(haconvdr_amd.synth.ance_state_dict — regenerated from the seed on both sides, never committed) and
seeded token batches.  Fixtures are data only: seeds, token ids, masks and the reference's outputs.
Library versions here: torch 2.10 / transformers 5.15 (the reference pins 1.8.1 / 4.2.0).

Run:  python tests/golden/make_golden_encoder.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from haconvdr_amd import synth  # noqa: E402

REF = "/root/reference"


def encoder_case_inputs(seed, lens, L):
    """ids int64 [B, L] padded with token id 0 (both reference pipelines pad with 0:
    gen_tokenized_doc.py:18,242; src/data.py:8), mask = first len ones."""
    B = len(lens)
    ids, _ = synth.token_batch(seed, B, L, fixed_len=L)
    ids = ids.astype(np.int64)
    mask = np.zeros((B, L), np.int64)
    for b, n in enumerate(lens):
        ids[b, n - 1] = 2
        ids[b, n:] = 0
        mask[b, :n] = 1
    return ids, mask


CASES = [
    # name, n_layers, L, lens, special
    ("l2_mixed", 2, 512, [5, 33, 64, 100, 257, 384, 511, 512], "tok1"),
    ("l2_full384", 2, 384, [384, 384, 384, 384], None),
    ("l12_mixed", 12, 512, [8, 31, 64, 129, 256, 384, 512, 40], "tok1"),
    ("l12_qrecc256", 12, 256, [256, 17, 200, 64], None),
]


def main():
    sys.path[:0] = [REF, os.path.join(REF, "src")]
    import torch
    import models  # the reference's src/models.py
    from transformers import RobertaConfig
    for name, n_layers, L, lens, special in CASES:
        seed = int.from_bytes(name.encode()[:4], "little")
        cfg = RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=n_layers, num_attention_heads=12,
                            intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                            pad_token_id=1, bos_token_id=0, eos_token_id=2)
        model = models.ANCE(cfg).eval()
        sd = synth.ance_state_dict(0xA11CE, n_layers)
        missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected and all(m.startswith("classifier.") or "position_ids" in m for m in missing), (missing, unexpected)
        ids, mask = encoder_case_inputs(seed, lens, L)
        if special == "tok1":
            ids[1, 3] = 1        # RoBERTa's pad id INSIDE a sequence: exercises the cumsum position rule
            ids[6, 100] = 1
        with torch.no_grad():
            out = model(torch.from_numpy(ids), torch.from_numpy(mask)).numpy()
        # the reference's result must not depend on what sits in masked positions (SURVEY §3.3 [probed])
        ids2 = ids.copy()
        ids2[mask == 0] = 1
        with torch.no_grad():
            out2 = model(torch.from_numpy(ids2), torch.from_numpy(mask)).numpy()
        np.savez_compressed(os.path.join(HERE, f"encoder_{name}.npz"), seed=seed, n_layers=n_layers, L=L, lens=np.array(lens),
                            ids=ids.astype(np.int32), mask=mask.astype(np.int8), ref_out=out,
                            pad_invariance_maxdiff=np.abs(out - out2).max())
        print(name, out.shape, out.dtype, "norm", np.linalg.norm(out[0]), "pad-invariance", np.abs(out - out2).max())


if __name__ == "__main__":
    main()
