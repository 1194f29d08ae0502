#!/usr/bin/env python3
"""Generate tests/golden/passages61/: a 61-record tokenized file and the FOUR block files the reference's passage pipeline
writes for it -- three full blocks and the tail flush (gen_doc_embeddings.py:87-88 block rule, :127-142 block writer,
:144-155 tail) -- with content-sensitive 2-layer synthetic weights (synth.ance_state_dict(layer_matrix_std=0.08)).

The reference hard-codes 2 500 000 passages per block (:87); reaching a block boundary with its own constant would need
more than 1.25 million encoded passages on this CPU.  So the generator runs the reference's module WITH THAT ONE LITERAL
REPLACED by 20, in memory, at generation time: the module's source is read from /root/reference, the assignment
`expect_per_block_passage_num = 2500000` becomes `= 20`, and the result is executed under the module's own name -- every
other line (reader, dataset, dataloader, loop, pickle calls) is the reference's.  Nothing of it is stored: the fixture is
the tokenized input and the four pickled blocks.

Run:  python tests/golden/make_golden_passages_multiblock.py
"""
import argparse
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from haconvdr_amd import synth  # noqa: E402
from haconvdr_amd.passages import write_tokenized_passages  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "passages61")
PER_BLOCK = 20          # what the one patched literal becomes
N, L, BATCH, STD, LAYERS = 61, 32, 4, 0.08, 2


def inputs():
    ids, lens = synth.token_batch(0x61B10C, N, L, min_len=3)
    lens[7] = L                      # one full-length record
    ids[7, L - 1] = 2
    return ids, lens


def main():
    os.makedirs(OUT, exist_ok=True)
    ids, lens = inputs()
    write_tokenized_passages(os.path.join(OUT, "passages"), ids, lens)
    for name in ("toml", "IPython"):
        m = types.ModuleType(name)
        if name == "IPython":
            m.embed = lambda *a, **k: None
        sys.modules.setdefault(name, m)
    sys.path[:0] = [REF, os.path.join(REF, "src")]
    import torch
    import models
    sys.modules["transformers"].AdamW = torch.optim.AdamW
    src = open(os.path.join(REF, "gen_doc_embeddings.py")).read()
    needle = "expect_per_block_passage_num = 2500000"
    assert src.count(needle) == 1
    G = types.ModuleType("gen_doc_embeddings")
    G.__file__ = os.path.join(REF, "gen_doc_embeddings.py")
    exec(compile(src.replace(needle, f"expect_per_block_passage_num = {PER_BLOCK}"), G.__file__, "exec"), G.__dict__)
    from transformers import RobertaConfig
    cfg = RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=LAYERS, num_attention_heads=12, intermediate_size=3072,
                        max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    model = models.ANCE(cfg).eval()
    sd = synth.ance_state_dict(0xA11CE, LAYERS, layer_matrix_std=STD)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    args = argparse.Namespace(max_seq_length=L, per_gpu_eval_batch_size=BATCH, n_gpu=1, local_rank=-1, device=torch.device("cpu"),
                              data_output_path=OUT, disable_tqdm=True)
    cache = G.EmbeddingCache(os.path.join(OUT, "passages"))
    with cache as emb:
        G.StreamInferenceDoc(args, model, G.GetProcessingFn(args, query=False), "passage_", emb, is_query_inference=False)
    rows = []
    for b in range(8):
        p = os.path.join(OUT, f"passage_emb_block_{b}.pb")
        if not os.path.exists(p):
            break
        e = pickle.load(open(p, "rb"))
        i = pickle.load(open(os.path.join(OUT, f"passage_embid_block_{b}.pb"), "rb"))
        rows.append((b, e.shape, e.dtype, i[0], i[-1]))
    print("reference wrote", rows)


if __name__ == "__main__":
    main()
