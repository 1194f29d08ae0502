#!/usr/bin/env python3
"""Generate tests/golden/passages23/ by RUNNING THE REFERENCE's passage pipeline
(gen_doc_embeddings.py: EmbeddingCache -> StreamingDataset/GetProcessingFn -> DataLoader ->
InferenceEmbeddingFromStreamDataLoader -> pickle blocks) on a 23-record tokenized file with a 2-layer
synthetic-weight ANCE on CPU.  Commits: the tokenized ``passages`` file + ``passages_meta`` (data, in the
format of gen_tokenized_doc.py:164-179,244) and the reference's output block files.

Run:  python tests/golden/make_golden_passages.py
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
OUT = os.path.join(HERE, "passages23")


def main():
    os.makedirs(OUT, exist_ok=True)
    L, N = 64, 23
    ids, lens = synth.token_batch(0x9A55, N, L, min_len=3)
    lens[5] = L                      # one full-length record
    ids[5, L - 1] = 2
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
    import gen_doc_embeddings as G
    from transformers import RobertaConfig

    # the reference's reader must agree with the file written above
    cache = G.EmbeddingCache(os.path.join(OUT, "passages"))
    with cache as emb:
        for i in (0, 5, 22):
            plen, arr = emb[i]
            assert plen == lens[i] and np.array_equal(arr, ids[i]), i

    cfg = RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                        max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    model = models.ANCE(cfg).eval()
    sd = synth.ance_state_dict(0xA11CE, 2)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    args = argparse.Namespace(max_seq_length=L, per_gpu_eval_batch_size=4, n_gpu=1, local_rank=-1, device=torch.device("cpu"),
                              data_output_path=OUT, disable_tqdm=True)
    with cache as emb:
        G.StreamInferenceDoc(args, model, G.GetProcessingFn(args, query=False), "passage_", emb, is_query_inference=False)
    e = pickle.load(open(os.path.join(OUT, "passage_emb_block_0.pb"), "rb"))
    i = pickle.load(open(os.path.join(OUT, "passage_embid_block_0.pb"), "rb"))
    print("reference wrote", e.shape, e.dtype, i.shape, i.dtype, i[:5])


if __name__ == "__main__":
    main()
