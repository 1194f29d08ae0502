#!/usr/bin/env python3
"""Generate tests/golden/chain_qrecc.npz by RUNNING THE REFERENCE's ``main()`` chain
(/root/reference/src/test_HAConvDR_qrecc.py:375-383):

    build_faiss_index(args) -> get_test_query_embedding(args) (:165-219) -> gen_metric_score_and_save (:355-372)
        = search_one_by_one_with_faiss (:74-162) -> pickle.load(offset2pid) -> output_test_res (:222-286)

on the synthetic QReCC test file of tests/golden/queries/, with a 2-layer synthetic-weight ANCE checkpoint directory
(config.json + model.safetensors, regenerated from a seed on both sides), CPU only.  Three things are replaced because the
packages / files do not exist offline, none of them on the path under test: ``RobertaTokenizer.from_pretrained`` returns
the stub tokenizer (tests/golden/stub_tokenizer.py, used on both sides), ``build_faiss_index`` returns ``NumpyFmafIndex``
(make_golden_search.py: faiss is not installable here) and ``print_trec_res`` is a no-op (pytrec_eval is absent).

The fixture is data: the reference's query embeddings, ``embedding2id``, merged (D, I), the TREC run file's text, and the
seeds that regenerate checkpoint and corpus through haconvdr_amd.synth.  Run:  python tests/golden/make_golden_chain.py
"""
import argparse
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from haconvdr_amd import synth  # noqa: E402
from tests.golden.make_golden_search import NumpyFmafIndex, import_reference_search  # noqa: E402
from tests.golden.stub_tokenizer import StubTokenizer  # noqa: E402

CHAIN = dict(weights_seed=0xC4A1, n_layers=2, corpus_seed=0xC0895, corpus_rows=900, blocks=3, passage_block_num=5, top_k=20,
             test_type="convqa", max_query_length=32, max_doc_length=256, max_response_length=64, max_concat_length=128,
             per_gpu_test_batch_size=4, n_gpu=1, seed=42, use_PRL=False,
             # round 5: content-sensitive weights (synth.ance_state_dict's docstring): the reference's embeddings of
             # different queries are >= 0.05 apart in 1-cos, so the encode leg's bound can tell them apart
             layer_matrix_std=0.08)


def write_checkpoint(path, seed, n_layers, layer_matrix_std=0.02):
    """config.json + model.safetensors of a synthetic ANCE checkpoint (the key names of the reference's state dict)."""
    import json
    import torch
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    sd = synth.ance_state_dict(seed, n_layers, layer_matrix_std=layer_matrix_std)
    save_file({k: torch.from_numpy(v).contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    cfg = {"architectures": ["RobertaForSequenceClassification"], "model_type": "roberta", "vocab_size": 50265, "hidden_size": 768,
           "num_hidden_layers": n_layers, "num_attention_heads": 12, "intermediate_size": 3072, "hidden_act": "gelu",
           "hidden_dropout_prob": 0.1, "attention_probs_dropout_prob": 0.1, "max_position_embeddings": 514, "type_vocab_size": 1,
           "layer_norm_eps": 1e-5, "pad_token_id": 1, "bos_token_id": 0, "eos_token_id": 2, "position_embedding_type": "absolute"}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    return sd


def write_corpus(path, seed, rows, blocks):
    """``blocks`` passage blocks (pickle protocol 4, gen_doc_embeddings.py:127-142) + offset2pid.pickle (a python list,
    gen_tokenized_doc.py) with few distinct passage ids, so that output_test_res's de-duplication is exercised."""
    os.makedirs(path, exist_ok=True)
    x = synth.embeddings(seed, rows)
    bounds = np.linspace(0, rows, blocks + 1).astype(int)
    for b in range(blocks):
        lo, hi = int(bounds[b]), int(bounds[b + 1])
        with open(os.path.join(path, f"passage_emb_block_{b}.pb"), "wb") as f:
            pickle.dump(x[lo:hi], f, protocol=4)
        with open(os.path.join(path, f"passage_embid_block_{b}.pb"), "wb") as f:
            pickle.dump(np.arange(lo, hi, dtype=np.int64), f, protocol=4)
    offset2pid = [int(v) for v in (synth.uniform_u32(seed + 9, rows) % np.uint32(rows // 3)) + 1000]
    with open(os.path.join(path, "offset2pid.pickle"), "wb") as f:
        pickle.dump(offset2pid, f)
    return x, offset2pid


def chain_args(tmp, **over):
    c = dict(CHAIN, **over)
    return argparse.Namespace(
        test_file_path=os.path.join(HERE, "queries", "qrecc_test.jsonl"), passage_embeddings_dir_path=os.path.join(tmp, "emb"),
        passage_offset2pid_path=os.path.join(tmp, "emb", "offset2pid.pickle"), pretrained_encoder_path=os.path.join(tmp, "ckpt"),
        qrel_output_path=tmp, output_trec_file="run.trec", trec_gold_qrel_file_path=os.path.join(tmp, "qrel.tsv"),
        test_type=c["test_type"], use_PRL=c["use_PRL"], is_train=False, top_k=c["top_k"], n_gpu=c["n_gpu"], rel_threshold=1, seed=c["seed"],
        per_gpu_test_batch_size=c["per_gpu_test_batch_size"], passage_block_num=c["passage_block_num"], disable_tqdm=True, use_gpu=True,
        max_query_length=c["max_query_length"], max_doc_length=c["max_doc_length"], max_response_length=c["max_response_length"],
        max_concat_length=c["max_concat_length"],
        # fields the reference's dataset class reads beside the script's own flags
        is_PRF=False, PRF_top=3, is_pseudo_prepos=False, hard_neg_type="bm25")


def main():
    import torch
    ref = import_reference_search()

    class _Tok:
        @staticmethod
        def from_pretrained(path, do_lower_case=True):
            return StubTokenizer()

    ref.RobertaTokenizer = _Tok
    ref.print_trec_res = lambda *a, **k: {"stub": True}
    ref.build_faiss_index = lambda args: NumpyFmafIndex(768)
    with tempfile.TemporaryDirectory() as tmp:
        write_checkpoint(os.path.join(tmp, "ckpt"), CHAIN["weights_seed"], CHAIN["n_layers"], CHAIN["layer_matrix_std"])
        write_corpus(os.path.join(tmp, "emb"), CHAIN["corpus_seed"], CHAIN["corpus_rows"], CHAIN["blocks"])
        args = chain_args(tmp)
        args.device = torch.device("cpu")
        captured = {}
        real_q, real_s = ref.get_test_query_embedding, ref.search_one_by_one_with_faiss

        def q_spy(a):
            captured["emb"], captured["ids"] = real_q(a)
            return captured["emb"], captured["ids"]

        def s_spy(*a):
            captured["D"], captured["I"] = real_s(*a)
            return captured["D"], captured["I"]

        ref.get_test_query_embedding, ref.search_one_by_one_with_faiss = q_spy, s_spy
        ref.get_args = lambda: args
        ref.main()                                           # :375-383, verbatim
        text = open(os.path.join(tmp, "run.trec")).read()
    emb = np.asarray(captured["emb"])
    D, I = np.asarray(captured["D"]), np.asarray(captured["I"])
    assert emb.dtype == np.float32 and emb.shape[1] == 768
    np.savez_compressed(os.path.join(HERE, "chain_qrecc.npz"), embeddings=emb, embedding2id=np.array(captured["ids"]),
                        ref_D=D, ref_I=I, trec_text=np.array(text), batch_size=args.batch_size,
                        **{k: np.array(v) for k, v in CHAIN.items()})
    print("queries", emb.shape, "merged", D.shape, D.dtype, I.dtype, "trec lines", text.count("\n"), "batch_size", args.batch_size)
    print(text[:240])


if __name__ == "__main__":
    main()
