"""Query side of the test scripts: host-side mirror of ``get_test_query_embedding(args)`` and ``main()``
(src/test_HAConvDR_qrecc.py:165-219, :375-383 = src/test_HAConvDR_topiocqa.py:165-219, :375-383).

``get_test_query_embedding(args)`` has the reference's signature and does what it does — encoder, tokenizer and config
from ``args.pretrained_encoder_path`` (:168-170), the test dataset from ``args.test_file_path`` (:175), a DataLoader of
``per_gpu_test_batch_size * max(1, n_gpu)`` conversations per batch (:173-179), the ``test_type`` → key mapping
(:192-207) — and returns ``(embeddings float32 [nq, 768], embedding2id list)`` (:216-219).  What differs is how the hot
loop runs: the reference encodes each batch of 4 on its own and synchronises with ``.detach().cpu().numpy()`` (:211-212);
here batches accumulate into encoder calls of up to ``max_queries_per_call`` sequences and come back in one copy.

The two scripts differ in the dataset class only (``Retrieval_qrecc`` :175 / ``Retrieval_topiocqa``): ``args.dataset``
("qrecc", the default, or "topiocqa") selects it.  ``get_test_query_embedding_from_loader`` is the same loop for a
caller that already holds a model and an iterable of collated batches.

``run_test(args)`` is ``main()`` without the argument parsing: build_index → get_test_query_embedding →
gen_metric_score_and_save (:375-383); ``python -m haconvdr_amd.queries --dataset qrecc ...`` parses the reference's flags.
"""
import logging

import numpy as np

logger = logging.getLogger(__name__)

_KEYS = {  # test_type -> (ids key, mask key)   (:192-207)
    "rewrite": ("bt_rewrite", "bt_rewrite_mask"),
    "raw": ("bt_raw_query", "bt_raw_query_mask"),
    "convq": ("bt_conv_q", "bt_conv_q_mask"),
    "convqa": ("bt_conv_qa", "bt_conv_qa_mask"),
    "convqp": ("bt_conv_qp", "bt_conv_qp_mask"),
}


def get_test_query_embedding_from_loader(model, test_loader, test_type="convqa", device=0, max_queries_per_call=1024):
    """model: ANCEEncoder; test_loader: iterable of collated batches as the reference's datasets
    produce them (dict with ``bt_sample_ids`` and the id/mask LongTensors of the chosen test_type).
    Returns (np.float32 [nq, 768], list of sample ids) exactly like the reference (:216-219)."""
    import torch
    if test_type not in _KEYS:
        raise ValueError("test type:{}, has not been implemented.".format(test_type))   # the reference's message (:209)
    kid, kmask = _KEYS[test_type]
    dev = torch.device("cuda", device)
    embeddings, embedding2id = [], []
    pend_ids, pend_mask, pend_n = [], [], 0

    def flush():
        nonlocal pend_ids, pend_mask, pend_n
        if not pend_ids:
            return
        L = max(t.shape[1] for t in pend_ids)
        ids = torch.zeros((pend_n, L), dtype=torch.int64)
        mask = torch.zeros((pend_n, L), dtype=torch.int64)
        r = 0
        for a, m in zip(pend_ids, pend_mask):
            ids[r:r + a.shape[0], :a.shape[1]] = a
            mask[r:r + a.shape[0], :a.shape[1]] = m
            r += a.shape[0]
        out = model(ids.to(dev), mask.to(dev)).cpu().numpy()
        bad = np.flatnonzero(np.isnan(out).any(axis=1))
        if bad.size:   # the device path flags a sequence it cannot encode with a NaN row (include/haconvdr.h)
            raise ValueError(f"query {sum(len(e) for e in embeddings) + int(bad[0])}: attention mask is not a non-empty prefix mask "
                             "or a token id lies outside the vocabulary")
        embeddings.append(out)
        pend_ids, pend_mask, pend_n = [], [], 0

    for batch in test_loader:
        embedding2id.extend(batch["bt_sample_ids"])                     # :213
        a, m = torch.as_tensor(batch[kid]), torch.as_tensor(batch[kmask])
        pend_ids.append(a)
        pend_mask.append(m)
        pend_n += a.shape[0]
        if pend_n >= max_queries_per_call:
            flush()
    flush()
    if not embeddings:
        raise ValueError("need at least one array to concatenate")       # what np.concatenate([]) raises there (:216)
    return np.concatenate(embeddings, axis=0), embedding2id             # :216


def _load_tokenizer(path):
    """RobertaTokenizer.from_pretrained(args.pretrained_encoder_path, do_lower_case=True) (:169)."""
    from transformers import RobertaTokenizer
    return RobertaTokenizer.from_pretrained(path, do_lower_case=True)


def _device_ordinal(args):
    dev = getattr(args, "device", None)          # the reference's get_args sets torch.device("cuda:0") (:417-421)
    idx = getattr(dev, "index", None)
    return int(idx) if idx is not None else 0


def get_test_query_embedding(args, *legacy, **legacy_kw):
    """Mirror of get_test_query_embedding(args) (:165-219).  Called with (model, test_loader, ...) — the round-2
    form — it forwards to get_test_query_embedding_from_loader."""
    if legacy or legacy_kw or not hasattr(args, "pretrained_encoder_path"):
        return get_test_query_embedding_from_loader(args, *legacy, **legacy_kw)
    from torch.utils.data import DataLoader
    from . import query_construction as qc
    from .encoder import ANCEEncoder
    set_seed(args)                                                                        # :166
    tokenizer = _load_tokenizer(args.pretrained_encoder_path)                              # :169
    device = _device_ordinal(args)
    model = ANCEEncoder.from_pretrained(args.pretrained_encoder_path, device=device)       # :168,:170 (config.json is read there)
    args.batch_size = args.per_gpu_test_batch_size * max(1, args.n_gpu)                    # :173
    logger.info("Buidling test dataset...")
    dataset = getattr(args, "dataset", "qrecc")
    if dataset not in ("qrecc", "topiocqa"):
        raise ValueError(f"args.dataset = {dataset!r}: 'qrecc' (test_HAConvDR_qrecc.py) or 'topiocqa' (test_HAConvDR_topiocqa.py)")
    cls = qc.Retrieval_qrecc if dataset == "qrecc" else qc.Retrieval_topiocqa
    test_dataset = cls(args, tokenizer, args.test_file_path)                               # :175
    test_loader = DataLoader(test_dataset, batch_size=args.batch_size, shuffle=False,
                             collate_fn=test_dataset.get_collate_fn(args))                 # :176-179
    logger.info("Generating query embeddings for testing...")
    return get_test_query_embedding_from_loader(model, test_loader, args.test_type, device=device,
                                                max_queries_per_call=int(getattr(args, "max_queries_per_call", 1024)))


def set_seed(args):
    """src/utils.py:106-111 (nothing on the inference path draws random numbers; kept for the same side effects)."""
    import random
    import torch
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if args.n_gpu > 0:
        torch.cuda.manual_seed_all(args.seed)


def run_test(args):
    """main() (:375-383) after get_args: index, query embeddings, retrieval + TREC file (+ metrics when the gold qrel
    file exists).  Returns what output_test_res returns (the metric dict, or the TREC file's path without a qrel file)."""
    from .index import build_index
    from .trec import gen_metric_score_and_save
    set_seed(args)
    index = build_index(args)
    query_embeddings, query_embedding2id = get_test_query_embedding(args)
    res = gen_metric_score_and_save(args, index, query_embeddings, query_embedding2id)
    logger.info("Test finish!")
    return res


def get_args(argv=None):
    """The flags of both scripts' get_args (:386-425); defaults follow ``--dataset`` where the two differ
    (paths, test_type convqa / convqp, passage_block_num 22 / 26, max_concat_length 256 / 512, max_doc_length 256 / 384,
    max_response_length 64 / 32).
    The reference's ``type=bool`` flags (any non-empty string is True) are parsed as real booleans here."""
    import argparse
    import torch
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--dataset", choices=("qrecc", "topiocqa"), default="qrecc")
    ds = pre.parse_known_args(argv)[0].dataset
    q = ds == "qrecc"

    def flag(v):
        return str(v).lower() not in ("", "0", "false", "no")
    p = argparse.ArgumentParser(parents=[pre])
    p.add_argument("--test_file_path", type=str, default="../datasets/qrecc/test_with_gold_rel.json" if q else "../datasets/topiocqa/test_with_info.json")
    p.add_argument("--passage_collection_path", type=str, default="datasets/topiocqa/full_wiki_segments.tsv")   # topiocqa's flag, unused there too
    p.add_argument("--passage_embeddings_dir_path", type=str, default=None if q else "datasets/topiocqa/embeds")
    p.add_argument("--passage_offset2pid_path", type=str, default=None if q else "datasets/topiocqa/tokenized/offset2pid.pickle")
    p.add_argument("--pretrained_encoder_path", type=str, default="../output/qrecc/model/best_model" if q else
                   "../output/topiocqa/model/imp-bs32-convqp-goldPRL-Truepreposhard-Trueprepos-best-retriever")
    p.add_argument("--qrel_output_path", type=str, default=f"../output/{ds}")
    p.add_argument("--output_trec_file", type=str, default=None if q else "")
    p.add_argument("--trec_gold_qrel_file_path", type=str,
                   default="datasets/qrecc/new_preprocessed/qrecc_qrel.tsv" if q else "datasets/topiocqa/topiocqa_qrel.trec")
    p.add_argument("--test_type", type=str, default="convqa" if q else "convqp")
    p.add_argument("--use_PRL", type=flag, default=False)
    p.add_argument("--is_train", type=flag, default=False)
    p.add_argument("--is_PRF", type=flag, default=False)
    p.add_argument("--top_k", type=int, default=100)
    p.add_argument("--n_gpu", type=int, default=1)
    p.add_argument("--rel_threshold", type=int, default=1)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--per_gpu_test_batch_size", type=int, default=4)
    p.add_argument("--passage_block_num", type=int, default=22 if q else 26)
    p.add_argument("--disable_tqdm", type=flag, default=False)
    p.add_argument("--use_gpu", type=flag, default=True)
    p.add_argument("--max_query_length", type=int, default=32)
    p.add_argument("--max_doc_length", type=int, default=256 if q else 384)
    p.add_argument("--max_response_length", type=int, default=64 if q else 32)
    p.add_argument("--max_concat_length", type=int, default=256 if q else 512)
    args = p.parse_args(argv)
    if not args.use_gpu:
        raise SystemExit("haconvdr_amd is the GPU path: --use_gpu false has no counterpart (DESIGN.md 0)")
    args.device = torch.device("cuda:0")
    args.PRF_top, args.is_pseudo_prepos, args.hard_neg_type = 3, False, "bm25"   # read by the dataset classes beside the flags
    logger.info("---------------------The arguments are:---------------------")
    logger.info(args)
    return args


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    run_test(get_args())
