"""The reference's main() chain (src/test_HAConvDR_qrecc.py:375-383: build_faiss_index -> get_test_query_embedding(args)
-> search_one_by_one_with_faiss -> output_test_res), pinned by tests/golden/chain_qrecc.npz = what the REFERENCE produced
for the synthetic QReCC test file with a 2-layer synthetic checkpoint directory (tests/golden/make_golden_chain.py:
embeddings, embedding2id, merged (D, I), the TREC run file).

CPU: the oracle restates the chain (ance_oracle on the constructed queries, flat_ip_oracle + merge) and must reproduce the
fixture -> oracle pinned against the composition, not only against its parts.
GPU: haconvdr_amd.queries.get_test_query_embedding(args) / run_test(args) through the HIP path, two legs — encode leg within
the cosine bar, search + TREC leg fed the reference's embeddings byte-identical."""
import os
import time

import numpy as np
import pytest

from tests.golden import make_golden_chain as mk
from tests.golden.stub_tokenizer import StubTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "chain_qrecc.npz"))
TOPK = int(G["top_k"])


@pytest.fixture(scope="module")
def chain_dir(tmp_path_factory):
    """Checkpoint directory + passage blocks + offset2pid regenerated from the fixture's seeds."""
    tmp = str(tmp_path_factory.mktemp("chain"))
    sd = mk.write_checkpoint(os.path.join(tmp, "ckpt"), int(G["weights_seed"]), int(G["n_layers"]), float(G["layer_matrix_std"]))
    x, offset2pid = mk.write_corpus(os.path.join(tmp, "emb"), int(G["corpus_seed"]), int(G["corpus_rows"]), int(G["blocks"]))
    return tmp, sd, x, offset2pid


def _args(tmp):
    import torch
    args = mk.chain_args(tmp)
    args.device = torch.device("cuda:0")
    args.dataset = "qrecc"
    del args.trec_gold_qrel_file_path          # no gold qrel file in the fixture: the chain ends with the TREC file
    return args


def _cosd(a, b):
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


def _construct(tmp):
    """The queries as the product's dataset mirror builds them (equal to the reference's: test_query_construction.py)."""
    from haconvdr_amd import query_construction as qc
    args = mk.chain_args(tmp)
    ds = qc.Retrieval_qrecc(args, StubTokenizer(), args.test_file_path)
    batch = ds.get_collate_fn(args)([ds[i] for i in range(len(ds))])
    return batch["bt_sample_ids"], batch["bt_conv_qa"].numpy(), batch["bt_conv_qa_mask"].numpy()


def test_oracle_reproduces_the_reference_chain(chain_dir, oracle):
    from oracle import ance_oracle
    tmp, sd, x, offset2pid = chain_dir
    qids, ids, mask = _construct(tmp)
    assert [str(q) for q in G["embedding2id"]] == list(qids)
    emb = ance_oracle.ance_forward(sd, ids, mask)
    assert emb.shape == G["embeddings"].shape
    assert _cosd(emb, G["embeddings"]).max() < 1e-5          # fp32 both sides (sdpa vs explicit softmax)
    # search leg on the reference's own embeddings: ids and scores of the first top_k columns, bit for bit
    oD, oI = oracle.flat_ip_search(x, G["embeddings"], TOPK)
    np.testing.assert_array_equal(oI, G["ref_I"][:, :TOPK])
    np.testing.assert_array_equal(oD.astype(np.float64), G["ref_D"][:, :TOPK])


def test_trec_writer_reproduces_the_chain_file(chain_dir, tmp_path):
    """output_test_res fed the reference's merged matrices (2 * top_k columns, as the reference hands them over)."""
    import argparse
    from haconvdr_amd.trec import output_test_res
    _, _, _, offset2pid = chain_dir
    args = argparse.Namespace(top_k=TOPK, qrel_output_path=str(tmp_path), output_trec_file="run.trec")
    path = output_test_res([str(q) for q in G["embedding2id"]], G["ref_D"], G["ref_I"], offset2pid, args)
    assert open(path).read() == str(G["trec_text"])


def test_trec_writer_converts_a_large_offset_table_once(tmp_path):
    """offset2pid is a plain python list of 25M-54M entries in production; converting it per query row took
    0.4 s per row at 5M entries.  2M entries x 300 rows must take seconds, not minutes."""
    import argparse
    from haconvdr_amd.trec import output_test_res
    n, rows, k = 2_000_000, 300, 10
    offset2pid = list(range(7, 7 + n))
    rng = np.random.default_rng(3)
    I = rng.integers(0, n, size=(rows, k))
    D = np.sort(rng.random((rows, k)), axis=1)[:, ::-1].copy()
    args = argparse.Namespace(top_k=k, qrel_output_path=str(tmp_path), output_trec_file="big.trec")
    t0 = time.perf_counter()
    path = output_test_res([f"q{i}" for i in range(rows)], D, I, offset2pid, args)
    dt = time.perf_counter() - t0
    first = open(path).readline().split()
    assert first[0] == "q0" and int(first[2]) == int(I[0, 0]) + 7
    assert dt < 10.0, f"output_test_res took {dt:.1f} s for {rows} rows over a {n}-entry list"


def test_trec_metrics_read_both_qrel_forms(tmp_path):
    """test_HAConvDR_topiocqa.py parses with split(" ") (:299,:318), test_HAConvDR_qrecc.py with split() — its default
    qrel file is tab-separated (qrecc_qrel.tsv, :395).  Both forms must give the same metrics."""
    from haconvdr_amd.trec import print_trec_res
    rows = [("q1", "0", "a", "2"), ("q1", "0", "b", "1"), ("q1", "0", "x", "0"), ("q2", "0", "c", "1")]
    (tmp_path / "qrel.trec").write_text("".join(" ".join(r) + "\n" for r in rows))
    (tmp_path / "qrel.tsv").write_text("".join("\t".join(r) + "\n" for r in rows))
    lines = []
    for qid, docs in (("q1", ["x", "a", "b", "y"]), ("q2", ["z", "c"])):
        for i, d in enumerate(docs):
            lines.append(f"{qid} Q0 {d} {i + 1} {200 - i - 1} {1.0 - 0.1 * i} ance\n")
    (tmp_path / "run.trec").write_text("".join(lines))
    a = print_trec_res(str(tmp_path / "run.trec"), str(tmp_path / "qrel.trec"))
    b = print_trec_res(str(tmp_path / "run.trec"), str(tmp_path / "qrel.tsv"))
    assert a == b and a["MRR"] == 50.0 and a["Recall@10"] == 100.0
    (tmp_path / "run.tsv").write_text("".join(ln.replace(" ", "\t") for ln in lines))
    assert print_trec_res(str(tmp_path / "run.tsv"), str(tmp_path / "qrel.tsv")) == a


def test_trec_metrics_table_from_trec_evals_definitions(tmp_path):
    """print_trec_res against trec_eval's published definitions on a fixture worked out BY HAND (the literals below come from the
    formulas, not from this package) -- pytrec_eval itself is absent here, so this is what pins f-3's metric block
    (src/test_HAConvDR_topiocqa.py:288-353) short of the real thing.  What the cases discriminate:
      q1  ties in the run's score: trec_eval ranks by score descending, then by document id DESCENDING as a string ("d9" before "d10";
          a numeric or ascending tie-break would put the relevant d10 at rank 2 and give a reciprocal rank of 1/2 instead of 1/3);
      q2  a passage id on several lines (the reference's run dict keeps the LAST score, :325: `runs[query][passage] = rel` -- this is
          what the "0 ... 0" filler lines of a de-duplicated run do), graded judgements > 1 (ndcg_cut_3 takes the grade as the gain
          and the ideal ranking over ALL judged documents, retrieved or not), and rel_threshold binarising recall / MRR only;
      q3  no relevant document at all: every measure 0, and the query still counts in the averages;
      q4  in the run but not judged: skipped;  q5  judged but not in the run: ignored (pytrec_eval evaluates the run's queries)."""
    from haconvdr_amd.trec import print_trec_res
    qrel = [("q1", "d10", 1), ("q1", "d2", 0), ("q1", "d1", 1),
            ("q2", "p5", 3), ("q2", "p7", 1), ("q2", "p8", 2),
            ("q3", "a", 0), ("q3", "b", 0),
            ("q5", "zz", 1)]
    (tmp_path / "qrel.trec").write_text("".join(f"{q} 0 {p} {r}\n" for q, p, r in qrel))
    run = [("q1", "d9", 5), ("q1", "d10", 5), ("q1", "d2", 7), ("q1", "d1", 1),
           ("q2", "p5", 199), ("q2", "p7", 198), ("q2", "p5", 197), ("q2", "0", 196), ("q2", "0", 195),
           ("q3", "a", 3), ("q3", "b", 2),
           ("q4", "w", 9)]
    (tmp_path / "run.trec").write_text("".join(f"{q} Q0 {p} {i + 1} {s} {float(s)} ance\n" for i, (q, p, s) in enumerate(run)))
    # q1: ranking d2, d9, d10, d1 -> RR 1/3, recall 2/2, DCG@3 = 1 / log2(4), ideal 1 + 1 / log2(3)                    -> 0.30657360
    # q2: ranking p7 (198), p5 (197), 0 (195) -> RR 1, recall 2/3, DCG@3 = 1 + 3 / log2(3), ideal 3 + 2 / log2(3) + 1 / 2  -> 0.60749152
    # q3: zeros.   Averages over q1, q2, q3, in percent, 5 places:
    assert print_trec_res(str(tmp_path / "run.trec"), str(tmp_path / "qrel.trec")) == \
        {"MRR": 44.44444, "NDCG@3": 30.46884, "Recall@10": 55.55556, "Recall@100": 55.55556}
    # rel_threshold = 2: q1 has no relevant document left, q2's are p5 (rank 2) and p8 (not retrieved); NDCG keeps the grades
    assert print_trec_res(str(tmp_path / "run.trec"), str(tmp_path / "qrel.trec"), rel_threshold=2) == \
        {"MRR": 16.66667, "NDCG@3": 30.46884, "Recall@10": 16.66667, "Recall@100": 16.66667}


def test_get_args_defaults_follow_the_two_scripts():
    """src/test_HAConvDR_qrecc.py:386-414 / src/test_HAConvDR_topiocqa.py:386-414."""
    from haconvdr_amd import queries
    q = queries.get_args(["--output_trec_file", "r.trec"])
    assert (q.dataset, q.test_type, q.passage_block_num, q.max_concat_length, q.max_doc_length, q.max_response_length, q.top_k,
            q.per_gpu_test_batch_size) == ("qrecc", "convqa", 22, 256, 256, 64, 100, 4)
    assert q.trec_gold_qrel_file_path.endswith("qrecc_qrel.tsv")
    t = queries.get_args(["--dataset", "topiocqa"])
    assert (t.test_type, t.passage_block_num, t.max_concat_length, t.max_doc_length, t.max_response_length) == ("convqp", 26, 512, 384, 32)
    assert t.passage_offset2pid_path == "datasets/topiocqa/tokenized/offset2pid.pickle" and str(t.device) == "cuda:0"
    with pytest.raises(SystemExit):
        queries.get_args(["--use_gpu", ""])      # the reference's CPU branch has no counterpart


# ------------------------------------------------------------------------------------------------ GPU legs
@pytest.mark.gpu
def test_get_test_query_embedding_args_vs_reference_chain(chain_dir, monkeypatch):
    """Encode leg: get_test_query_embedding(args) with the reference's signature — checkpoint directory, test file,
    batch size rule — against the embeddings the reference's function returned; same embedding2id."""
    from haconvdr_amd import queries
    tmp = chain_dir[0]
    monkeypatch.setattr(queries, "_load_tokenizer", lambda path: StubTokenizer())
    args = _args(tmp)
    emb, e2id = queries.get_test_query_embedding(args)
    assert args.batch_size == int(G["batch_size"])                     # :173
    assert emb.dtype == np.float32 and emb.shape == G["embeddings"].shape
    assert [str(q) for q in G["embedding2id"]] == list(e2id)
    # bar 1e-3 (north_star) AND a tenth of the distance between two different queries of the fixture (content-sensitive
    # checkpoint: >= 0.16), centred cosine, relative L2; the same embeddings handed to the neighbouring queries must fail
    from tests import parity
    parity.assert_embeddings_match(emb, G["embeddings"], what="chain encode leg")
    parity.assert_negative_control(emb, G["embeddings"])


@pytest.mark.gpu
def test_search_and_trec_leg_byte_identical_to_reference_chain(chain_dir, tmp_path):
    """Search + TREC leg fed the REFERENCE's embeddings: gen_metric_score_and_save -> byte-identical TREC file;
    merged ids / scores equal the reference's first top_k columns."""
    from haconvdr_amd.index import build_index
    from haconvdr_amd.search import search_one_by_one
    from haconvdr_amd.trec import gen_metric_score_and_save
    tmp = chain_dir[0]
    args = _args(tmp)
    args.qrel_output_path = str(tmp_path)
    index = build_index(args)
    D, I = search_one_by_one(args, args.passage_embeddings_dir_path, index, G["embeddings"], args.top_k)
    assert D.dtype == np.float64 and I.dtype == np.int64
    np.testing.assert_array_equal(I, G["ref_I"][:, :TOPK])
    np.testing.assert_array_equal(D, G["ref_D"][:, :TOPK])
    path = gen_metric_score_and_save(args, index, G["embeddings"], [str(q) for q in G["embedding2id"]])
    assert open(path).read() == str(G["trec_text"])


@pytest.mark.gpu
def test_run_test_is_the_references_main(chain_dir, tmp_path, monkeypatch):
    """run_test(args) = main() (:375-383) end to end on the HIP path.  Its own embeddings differ from the reference's in
    the low bits (bf16 operands), so the TREC file is compared structurally: same queries in the same order, top_k ranks
    each, and the retrieved passage lists agree with the reference's except where scores are within the encoders' noise."""
    from haconvdr_amd import queries
    tmp = chain_dir[0]
    monkeypatch.setattr(queries, "_load_tokenizer", lambda path: StubTokenizer())
    args = _args(tmp)
    args.qrel_output_path = str(tmp_path)
    path = queries.run_test(args)
    got = [ln.split() for ln in open(path).read().splitlines()]
    want = [ln.split() for ln in str(G["trec_text"]).splitlines()]
    assert len(got) == len(want) and [g[0] for g in got] == [w[0] for w in want] and [g[3] for g in got] == [w[3] for w in want]
    # the score noise this encoder's rounding explains: for corpus rows of unit variance (out - ref) . x has standard deviation
    # |out - ref|; a returned score may differ from the reference's by six of those (the checkpoint is the content-sensitive one:
    # 1-cos ~ 1e-4 between the two encoders is |out - ref| ~ 0.4 on embeddings of norm 27.7)
    emb, e2id = queries.get_test_query_embedding(args)
    sigma = dict(zip(e2id, np.linalg.norm(emb.astype(np.float64) - G["embeddings"].astype(np.float64), axis=1)))
    assert max(sigma.values()) < 1.0, max(sigma.values())
    agree = 0
    for qid in dict.fromkeys(g[0] for g in got):
        a = [g[2] for g in got if g[0] == qid and g[2] != "0"]
        b = [w[2] for w in want if w[0] == qid and w[2] != "0"]
        agree += len(set(a[:10]) & set(b[:10])) >= 9
        sa = np.array([float(g[5]) for g in got if g[0] == qid and g[2] != "0"])
        sb = np.array([float(w[5]) for w in want if w[0] == qid and w[2] != "0"])
        n = min(len(sa), len(sb), 5)
        assert np.allclose(sa[:n], sb[:n], rtol=0, atol=6.0 * sigma[qid] + 1e-3), (qid, sa[:n], sb[:n], sigma[qid])
    assert agree >= 0.9 * len(set(g[0] for g in got))
    # with a gold qrel file the chain ends in the metric block (tab-separated, the QReCC form)
    args.trec_gold_qrel_file_path = str(tmp_path / "qrel.tsv")
    with open(args.trec_gold_qrel_file_path, "w") as f:
        for qid in dict.fromkeys(g[0] for g in got):
            top = next(g[2] for g in got if g[0] == qid)
            f.write(f"{qid}\t0\t{top}\t1\n")
    res = queries.run_test(args)
    assert res["MRR"] == 100.0 and res["Recall@10"] == 100.0


@pytest.mark.gpu
def test_topiocqa_chain_vs_oracle(chain_dir, tmp_path, monkeypatch):
    """The TopiOCQA form of the chain (src/test_HAConvDR_topiocqa.py: Retrieval_topiocqa, test_type convqp, max_concat_length
    512).  That script cannot be imported (it imports a class that does not exist), so no fixture of the reference's
    run exists; its dataset class is pinned by tests/golden/queries (test_query_construction.py) and the chain is checked
    against the oracle here: embeddings of the constructed queries, retrieval over the same blocks, TREC lines."""
    import torch
    from haconvdr_amd import queries, query_construction as qc
    from oracle import ance_oracle, oracle as orc
    tmp, sd, x, offset2pid = chain_dir
    monkeypatch.setattr(queries, "_load_tokenizer", lambda path: StubTokenizer())
    args = _args(tmp)
    args.dataset = "topiocqa"
    args.test_file_path = os.path.join(HERE, "golden", "queries", "topiocqa_test.jsonl")
    args.test_type, args.max_concat_length, args.max_doc_length, args.max_response_length = "convqp", 512, 384, 32
    args.qrel_output_path = str(tmp_path)
    emb, e2id = queries.get_test_query_embedding(args)
    ds = qc.Retrieval_topiocqa(args, StubTokenizer(), args.test_file_path)
    batch = ds.get_collate_fn(args)([ds[i] for i in range(len(ds))])
    assert list(batch["bt_sample_ids"]) == list(e2id) and emb.shape == (len(ds), 768)
    ref = ance_oracle.ance_forward(sd, batch["bt_conv_qp"].numpy(), batch["bt_conv_qp_mask"].numpy())
    from tests import parity
    parity.assert_embeddings_match(emb, ref, what="topiocqa chain vs oracle")
    parity.assert_negative_control(emb, ref)
    path = queries.run_test(args)
    lines = [ln.split() for ln in open(path).read().splitlines()]
    assert len(lines) == len(ds) * TOPK and lines[0][1] == "Q0" and lines[0][-1] == "ance"
    oD, oI = orc.flat_ip_search(x, emb, TOPK)                       # the product's own embeddings: ids must be the oracle's
    first_pid = {}
    for ln in lines:
        first_pid.setdefault(ln[0], ln[2])
    for row, qid in enumerate(dict.fromkeys(e2id)):
        assert first_pid[qid] == str(offset2pid[int(oI[row, 0])]), (qid, first_pid[qid])
