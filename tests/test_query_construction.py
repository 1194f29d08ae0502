"""f-4 query construction: the mirror of the reference's test-time datasets against goldens the
REFERENCE's own classes produced (tests/golden/make_golden_queries.py, stub tokenizer on both sides)."""
import argparse
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
QDIR = os.path.join(HERE, "golden", "queries")
GOLD = json.load(open(os.path.join(QDIR, "queries_golden.json")))
CASES = [k for k in GOLD if k != "padding_seq_to_same_length"]


def build(name):
    from haconvdr_amd import query_construction as qc
    from tests.golden.stub_tokenizer import StubTokenizer
    g = GOLD[name]
    args = argparse.Namespace(is_train=False, is_PRF=False, PRF_top=3, is_pseudo_prepos=False, hard_neg_type="bm25", **g["args"])
    cls = qc.Retrieval_topiocqa if g["dataset"] == "topiocqa" else qc.Retrieval_qrecc
    return cls(args, StubTokenizer(), os.path.join(QDIR, g["dataset"] + "_test.jsonl")), args


def test_padding_seq_to_same_length_vs_reference():
    from haconvdr_amd.query_construction import padding_seq_to_same_length
    g = GOLD["padding_seq_to_same_length"]
    for ids, want in zip(g["inputs"], g["outputs"]):
        got = padding_seq_to_same_length(list(ids), g["max_pad_length"])
        assert [list(got[0]), list(got[1])] == want
    assert padding_seq_to_same_length([5, 6], 4, pad_token=9) == ([5, 6, 9, 9], [1, 1, 0, 0])


@pytest.mark.parametrize("name", CASES)
def test_examples_equal_the_references(name):
    ds, args = build(name)
    want = GOLD[name]["examples"]
    assert len(ds) == len(want)
    for got, exp in zip(ds.examples, want):
        assert got == exp                        # sample id, every token id, every mask bit, the empty document fields
    # every sequence is fully padded to the configured length and its mask is a prefix mask (what the encoder requires)
    col = 3 if GOLD[name]["dataset"] == "topiocqa" else 1
    for ex in ds.examples:
        ids, mask = ex[col], ex[col + 1]
        n = sum(mask)
        assert len(ids) == len(mask) == args.max_concat_length and mask == [1] * n + [0] * (len(mask) - n) and n >= 1
        assert ids[0] == 0 and all(t == 0 for t in ids[n:])


@pytest.mark.parametrize("name", CASES)
def test_collated_batches_equal_the_references(name):
    import torch
    ds, args = build(name)
    batch = ds.get_collate_fn(args)([ds[i] for i in range(min(5, len(ds)))])
    want = GOLD[name]["first_batch"]
    assert list(batch) == list(want)             # same keys, same order
    for k, v in batch.items():
        if isinstance(want[k], list):
            assert v == want[k]
        else:
            assert isinstance(v, torch.Tensor) and str(v.dtype) == want[k]["dtype"] and list(v.shape) == want[k]["shape"]
            assert v.tolist() == want[k]["values"]


def test_training_and_prf_modes_are_refused():
    from haconvdr_amd import query_construction as qc
    from tests.golden.stub_tokenizer import StubTokenizer
    base = dict(use_PRL=True, max_query_length=32, max_doc_length=384, max_response_length=32, max_concat_length=512, PRF_top=3)
    for bad in (dict(is_train=True, is_PRF=False), dict(is_train=False, is_PRF=True)):
        with pytest.raises(NotImplementedError):
            qc.Retrieval_topiocqa(argparse.Namespace(**base, **bad), StubTokenizer(), os.path.join(QDIR, "topiocqa_test.jsonl"))
        with pytest.raises(NotImplementedError):
            qc.Retrieval_qrecc(argparse.Namespace(**base, **bad), StubTokenizer(), os.path.join(QDIR, "qrecc_test.jsonl"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,test_type", [("topiocqa_prl", "convqp"), ("qrecc_history", "convqa"), ("topiocqa_prl", "raw")])
def test_constructed_queries_through_the_encoder(name, test_type):
    """Dataset -> DataLoader (batch 4, the reference's default) -> get_test_query_embedding -> embeddings: the same as
    encoding the padded id / mask matrices directly, ids in file order (src/test_HAConvDR_topiocqa.py:175-219)."""
    import torch
    from torch.utils.data import DataLoader
    from haconvdr_amd import synth
    from haconvdr_amd.encoder import ANCEEncoder
    from haconvdr_amd.queries import get_test_query_embedding
    ds, args = build(name)
    enc = ANCEEncoder.from_state_dict(synth.ance_state_dict(0xA11CE, 2))
    loader = DataLoader(ds, batch_size=4, shuffle=False, collate_fn=ds.get_collate_fn(args))
    emb, ids = get_test_query_embedding(enc, loader, test_type)
    assert ids == [e[0] for e in ds.examples] and emb.shape == (len(ds), 768) and emb.dtype == np.float32
    col = {"convqp": 3, "convqa": 1, "raw": 1}[test_type]
    direct = enc(np.array([e[col] for e in ds.examples], np.int32), np.array([e[col + 1] for e in ds.examples], np.int32))
    np.testing.assert_array_equal(emb, direct)
