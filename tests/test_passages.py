"""Passage pipeline (SURVEY §8 a7–a9, f-1, f-2): formats on CPU, the encode loop on the GPU.
Golden: tests/golden/passages23/ = a 23-record tokenized file and the block files the REFERENCE's
gen_doc_embeddings pipeline wrote for it (tests/golden/make_golden_passages.py)."""
import os
import pickle

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden", "passages23")


def test_tokenized_reader_matches_reference_format():
    from haconvdr_amd import synth
    from haconvdr_amd.passages import TokenizedPassages
    tp = TokenizedPassages(os.path.join(G, "passages"))
    assert len(tp) == 23 and tp.L == 64 and tp.record_size == 4 + 64 * 4
    ids, lens = synth.token_batch(0x9A55, 23, 64, min_len=3)
    lens[5] = 64
    ids[5, 63] = 2
    got_ids, got_lens = tp.batch(0, 23)
    np.testing.assert_array_equal(got_ids, ids)
    np.testing.assert_array_equal(got_lens, lens)
    plen, arr = tp[7]                                     # EmbeddingCache.__getitem__ contract
    assert plen == lens[7] and np.array_equal(arr, ids[7])
    raw = open(os.path.join(G, "passages"), "rb").read()
    assert int.from_bytes(raw[:4], "big") == lens[0]      # big-endian length prefix (src/utils.py:325)


def test_block_reader_zero_copy_and_writer_format(tmp_path):
    from haconvdr_amd.passages import read_embedding_block, write_embedding_block
    ref_e = pickle.load(open(os.path.join(G, "passage_emb_block_0.pb"), "rb"))
    ref_i = pickle.load(open(os.path.join(G, "passage_embid_block_0.pb"), "rb"))
    e, i = read_embedding_block(G, 0, mmap=True)
    assert isinstance(e, np.memmap) and e.dtype == np.float32 and e.shape == (23, 768)
    np.testing.assert_array_equal(np.asarray(e), ref_e)
    np.testing.assert_array_equal(i, ref_i)
    write_embedding_block(str(tmp_path), 0, ref_e, ref_i)
    e2 = pickle.load(open(tmp_path / "passage_emb_block_0.pb", "rb"))      # readable exactly as the reference reads it (:82-93)
    i2 = pickle.load(open(tmp_path / "passage_embid_block_0.pb", "rb"))
    assert e2.dtype == np.float32 and i2.dtype == np.int64
    np.testing.assert_array_equal(e2, ref_e)
    np.testing.assert_array_equal(i2, ref_i)


@pytest.mark.gpu
def test_encode_passages_vs_reference_blocks(tmp_path):
    from haconvdr_amd import synth
    from haconvdr_amd.encoder import ANCEEncoder
    from haconvdr_amd.passages import TokenizedPassages, encode_passages, read_embedding_block
    enc = ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 2))
    tp = TokenizedPassages(os.path.join(G, "passages"))
    n = encode_passages(enc, tp, str(tmp_path / "one"), per_gpu_eval_batch_size=4, n_gpu=1)
    assert n == 23 and sorted(os.listdir(tmp_path / "one")) == ["passage_emb_block_0.pb", "passage_embid_block_0.pb"]
    e, i = read_embedding_block(str(tmp_path / "one"), 0)
    ref_e = pickle.load(open(os.path.join(G, "passage_emb_block_0.pb"), "rb"))
    ref_i = pickle.load(open(os.path.join(G, "passage_embid_block_0.pb"), "rb"))
    assert e.dtype == np.float32 and e.shape == ref_e.shape and i.dtype == np.int64
    np.testing.assert_array_equal(i, ref_i)
    cos = (e * ref_e).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(ref_e, axis=1))
    assert np.all(1 - cos < 1e-3) and np.all(1 - cos < 1e-4), 1 - cos
    # several blocks, two ranks: the union of the files equals the single-process output
    for r in range(2):
        encode_passages(enc, tp, str(tmp_path / "two"), per_gpu_eval_batch_size=4, n_gpu=1, rank=r, world_size=2,
                        expect_per_block_passage_num=8)
    blocks = [read_embedding_block(str(tmp_path / "two"), b) for b in range(3)]
    np.testing.assert_array_equal(np.concatenate([b[1] for b in blocks]), ref_i)
    np.testing.assert_array_equal(np.concatenate([b[0] for b in blocks]), e)


G61 = os.path.join(os.path.dirname(__file__), "golden", "passages61")


def test_multi_block_fixture_is_what_the_reference_writes():
    """tests/golden/passages61/ (make_golden_passages_multiblock.py: the reference's writer loop with its 2 500 000-per-block
    literal replaced by 20): 61 records at batch 4 -> three blocks of 20 and the tail flush of 1, ids = the enumerate index."""
    from haconvdr_amd.passages import TokenizedPassages, read_embedding_block
    from tests.golden.make_golden_passages_multiblock import inputs
    tp = TokenizedPassages(os.path.join(G61, "passages"))
    ids, lens = inputs()
    got_ids, got_lens = tp.batch(0, 61)
    np.testing.assert_array_equal(got_ids, ids)
    np.testing.assert_array_equal(got_lens, lens)
    sizes, all_ids = [], []
    for b in range(4):
        e, i = read_embedding_block(G61, b, mmap=True)
        assert e.dtype == np.float32 and e.shape[1] == 768 and i.dtype == np.int64 and len(i) == len(e)
        sizes.append(len(e))
        all_ids.append(i)
    assert sizes == [20, 20, 20, 1] and not os.path.exists(os.path.join(G61, "passage_emb_block_4.pb"))
    np.testing.assert_array_equal(np.concatenate(all_ids), np.arange(61))


@pytest.mark.gpu
def test_encode_passages_writes_the_references_blocks_across_block_boundaries(tmp_path):
    """The block rule (gen_doc_embeddings.py:87-88), the writer at a boundary (:127-142) and the tail flush (:144-155) against
    what the REFERENCE's loop wrote: same files, same shapes, same ids, embeddings inside the fixture-scaled bounds (content-
    sensitive weights: different passages are far apart, and the same embeddings shifted by one passage must fail); one rank
    and two ranks (block b belongs to rank b % 2) give the same files."""
    from haconvdr_amd import synth
    from haconvdr_amd.encoder import ANCEEncoder
    from haconvdr_amd.passages import TokenizedPassages, encode_passages, read_embedding_block
    from tests import parity
    from tests.golden import make_golden_passages_multiblock as mk
    enc = ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, mk.LAYERS, layer_matrix_std=mk.STD))
    tp = TokenizedPassages(os.path.join(G61, "passages"))
    n = encode_passages(enc, tp, str(tmp_path / "one"), per_gpu_eval_batch_size=mk.BATCH, n_gpu=1, expect_per_block_passage_num=mk.PER_BLOCK)
    assert n == 61
    assert sorted(os.listdir(tmp_path / "one")) == sorted(f for f in os.listdir(G61) if f.startswith("passage_emb"))
    got, ref = [], []
    for b in range(4):
        e, i = read_embedding_block(str(tmp_path / "one"), b)
        re_, ri = read_embedding_block(G61, b)
        assert e.shape == re_.shape and e.dtype == np.float32 and i.dtype == np.int64
        np.testing.assert_array_equal(i, ri)
        got.append(e)
        ref.append(np.asarray(re_))
    got, ref = np.concatenate(got), np.concatenate(ref)
    m = parity.assert_embeddings_match(got, ref, what="passages61")
    parity.assert_negative_control(got, ref)
    assert m["spread"]["raw_min"] > 0.01, m["spread"]
    for r in range(2):
        encode_passages(enc, tp, str(tmp_path / "two"), per_gpu_eval_batch_size=mk.BATCH, n_gpu=1, rank=r, world_size=2,
                        expect_per_block_passage_num=mk.PER_BLOCK)
    for b in range(4):
        e2, i2 = read_embedding_block(str(tmp_path / "two"), b)
        e1, i1 = read_embedding_block(str(tmp_path / "one"), b)
        np.testing.assert_array_equal(i2, i1)
        np.testing.assert_array_equal(e2, e1)


@pytest.mark.gpu
def test_encode_then_search_end_to_end(tmp_path):
    """Blocks written by encode_passages are searchable by search_one_by_one: a passage's own
    embedding retrieves it first."""
    import argparse
    from haconvdr_amd import synth
    from haconvdr_amd.encoder import ANCEEncoder
    from haconvdr_amd.index import FlatIPIndex
    from haconvdr_amd.passages import TokenizedPassages, encode_passages, read_embedding_block
    from haconvdr_amd.search import search_one_by_one
    enc = ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 2))
    tp = TokenizedPassages(os.path.join(G, "passages"))
    encode_passages(enc, tp, str(tmp_path), per_gpu_eval_batch_size=4, expect_per_block_passage_num=12)
    e0, _ = read_embedding_block(str(tmp_path), 0)
    D, I = search_one_by_one(argparse.Namespace(passage_block_num=5), str(tmp_path), FlatIPIndex(768), e0[:5], 3)
    assert list(I[:, 0]) == [0, 1, 2, 3, 4]


def _checkpoint_dir(path, n_layers=2, **cfg_over):
    """A HF-style ANCE checkpoint directory: pytorch_model.bin (+ the unused classifier head) and config.json."""
    import json
    import torch
    from haconvdr_amd import synth
    os.makedirs(path, exist_ok=True)
    sd = dict(synth.ance_state_dict(0xA11CE, n_layers))
    sd["classifier.dense.weight"] = np.zeros((768, 768), np.float32)
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, os.path.join(path, "pytorch_model.bin"))
    cfg = {"architectures": ["RobertaForSequenceClassification"], "model_type": "roberta", "num_hidden_layers": n_layers, "hidden_size": 768,
           "num_attention_heads": 12, "intermediate_size": 3072, "vocab_size": 50265, "max_position_embeddings": 514, "type_vocab_size": 1,
           "layer_norm_eps": 1e-05, "pad_token_id": 1, "bos_token_id": 0, "eos_token_id": 2, "hidden_act": "gelu", "finetuning_task": "MSMarco"}
    cfg.update(cfg_over)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    return path


@pytest.mark.gpu
def test_generate_new_ann_then_gen_metric_score_and_save(tmp_path, oracle):
    """The two entry points a user of the reference runs, end to end on the GPU: generate_new_ann(args) with the keys of
    Config/gen_doc_embeddings.toml (gen_doc_embeddings.py:190-212) writes the passage blocks — equal to the blocks the
    REFERENCE's pipeline wrote for the same 23 records — and gen_metric_score_and_save(args, index, Q, ids)
    (src/test_HAConvDR_topiocqa.py:355-372) searches them and writes the TREC run + metrics."""
    from types import SimpleNamespace
    from haconvdr_amd import synth
    from haconvdr_amd.index import build_index
    from haconvdr_amd.passages import generate_new_ann, read_embedding_block
    from haconvdr_amd.trec import gen_metric_score_and_save
    ckpt = _checkpoint_dir(str(tmp_path / "ad-hoc-ance-msmarco"))
    tok_dir = tmp_path / "tokenized"
    os.makedirs(tok_dir)
    for f in ("passages", "passages_meta"):
        os.symlink(os.path.join(G, f), tok_dir / f)
    gen_args = SimpleNamespace(model_type="ANCE", pretrained_passage_encoder=ckpt, max_seq_length=64, per_gpu_eval_batch_size=4, local_rank=-1,
                               disable_tqdm=True, n_gpu=1, tokenized_passage_collection_dir_path=str(tok_dir), data_output_path=str(tmp_path / "embeds"))
    assert generate_new_ann(gen_args) == 23
    e, i = read_embedding_block(gen_args.data_output_path, 0)
    ref_e = pickle.load(open(os.path.join(G, "passage_emb_block_0.pb"), "rb"))
    np.testing.assert_array_equal(i, pickle.load(open(os.path.join(G, "passage_embid_block_0.pb"), "rb")))
    cos = (e * ref_e).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(ref_e, axis=1))
    assert np.all(1 - cos < 1e-4)
    # retrieval over the freshly written blocks
    offset2pid = [1000 + 3 * (j // 2) for j in range(23)]            # pairs of offsets share a passage id: de-duplication
    with open(tmp_path / "offset2pid.pickle", "wb") as f:
        pickle.dump(offset2pid, f)
    q = synth.embeddings(4242, 6)
    qids = [f"conv{j // 2}_{j % 2 + 1}" for j in range(6)]
    with open(tmp_path / "qrels.txt", "w") as f:
        for j, qid in enumerate(qids):
            f.write(f"{qid} 0 {offset2pid[(5 * j) % 23]} 1\n")
    args = SimpleNamespace(n_gpu=1, use_gpu=True, passage_block_num=4, passage_embeddings_dir_path=gen_args.data_output_path, top_k=5,
                           passage_offset2pid_path=str(tmp_path / "offset2pid.pickle"), qrel_output_path=str(tmp_path), output_trec_file="run.trec",
                           trec_gold_qrel_file_path=str(tmp_path / "qrels.txt"), rel_threshold=1, test_file_path="unused")
    res = gen_metric_score_and_save(args, build_index(args), q, qids)
    assert set(res) == {"MRR", "NDCG@3", "Recall@10", "Recall@100"} and all(0.0 <= v <= 100.0 for v in res.values())
    lines = open(tmp_path / "run.trec").read().splitlines()
    assert len(lines) == 6 * 5
    oD, oI = oracle.flat_ip_search(e, q, 5)                           # exact search over the very rows that were written
    for j, qid in enumerate(qids):
        want, seen = [], set()
        for off, sc in zip(oI[j], oD[j].astype(np.float64).tolist()):
            if offset2pid[off] not in seen:
                seen.add(offset2pid[off])
                want.append((offset2pid[off], sc))
        want += [(0, 0)] * (5 - len(want))
        got = [ln.split(" ") for ln in lines[5 * j:5 * j + 5]]
        assert [g[0] for g in got] == [qid] * 5 and [g[3] for g in got] == list("12345") and [g[4] for g in got] == ["199", "198", "197", "196", "195"]
        assert [(g[2], g[5]) for g in got] == [(str(p), str(s)) for p, s in want] and all(g[1] == "Q0" and g[6] == "ance" for g in got)


@pytest.mark.gpu
def test_from_pretrained_reads_and_checks_config_json(tmp_path):
    from haconvdr_amd import synth
    from haconvdr_amd.encoder import ANCEEncoder
    from haconvdr_amd.passages import load_model
    from oracle import ance_oracle
    ids, lens = synth.token_batch(5, 3, 48, min_len=6)
    mask = (np.arange(48)[None, :] < lens[:, None]).astype(np.int32)
    # layer_norm_eps and pad_token_id come from config.json, not from defaults: an unusual eps must change the output
    # exactly as it changes the fp32 oracle's
    tok, enc = load_model("ANCE_Query", _checkpoint_dir(str(tmp_path / "a"), layer_norm_eps=1e-2))
    assert tok is None
    sd = synth.ance_state_dict(0xA11CE, 2)
    ref = ance_oracle.ance_forward(sd, ids, mask, eps=1e-2)
    out = enc(ids, mask)
    cos = (out * ref).sum(1) / (np.linalg.norm(out, axis=1) * np.linalg.norm(ref, axis=1))
    assert np.all(1 - cos < 1e-4)
    plain = ANCEEncoder.from_pretrained(_checkpoint_dir(str(tmp_path / "b")))(ids, mask)
    assert np.abs(plain - out).max() > 1e-3
    for bad in (dict(num_hidden_layers=3), dict(intermediate_size=4096), dict(vocab_size=30522), dict(hidden_act="relu"),
                dict(num_attention_heads=16), dict(max_position_embeddings=512)):
        with pytest.raises(ValueError):
            ANCEEncoder.from_pretrained(_checkpoint_dir(str(tmp_path / "c"), **bad))
    with pytest.raises(ValueError):
        load_model("BERT_Query", str(tmp_path / "a"))
