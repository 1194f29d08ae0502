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
