"""Result post-processing: host-side mirror of ``output_test_res`` and ``gen_metric_score_and_save``
(src/test_HAConvDR_topiocqa.py:222-286, :355-372).

Writes the same TREC run file, line for line: ``"{qid} Q0 {pid} {rank} {200-rank} {score} ance"``
(:282).  Quirks of the reference that are kept because downstream files depend on them:
per-query passage-id de-duplication leaves the unused tail slots at ``(0, 0)`` (:243-255), and a
query id that occurs twice reuses the first occurrence's list (:241-246).
``print_trec_res`` restates the metric block (:288-353) with trec_eval's published definitions, because
``pytrec_eval`` is not installable here: **parity unpinned** against it (no reference output to compare with; the unit
tests are hand-worked and a randomized cross-check against scikit-learn's ndcg_score).  Pass ``evaluate=`` to plug in pytrec_eval where it exists.
"""
import logging
import math
import os
import pickle

import numpy as np

logger = logging.getLogger(__name__)


def _pid_table(offset2pid):
    """offset2pid as the pickle held it (gen_tokenized_doc.py writes a plain python list: 25M entries for TopiOCQA, 54M for
    QReCC; a dict or an ndarray are accepted too) -> something cheap to index many times.  The list is converted ONCE per
    run, not once per query row."""
    if isinstance(offset2pid, (dict, np.ndarray)):
        return offset2pid
    return np.asarray(offset2pid)


def _pids_of(table, offsets):
    """passage ids of a row of corpus offsets (table from _pid_table)."""
    if isinstance(table, dict):
        return [table[int(o)] for o in offsets]
    return table[np.asarray(offsets, dtype=np.int64)].tolist()


def output_test_res(query_embedding2id, retrieved_scores_mat, retrieved_pid_mat, offset2pid, args, evaluate=None):
    """TREC run file of a retrieval result (src/test_HAConvDR_topiocqa.py:222-286).  Per query row: the first top_k
    (offset, score) columns, offsets mapped to passage ids, a passage id kept only where it first occurs; the survivors
    fill the ranks from 1, the ranks left over read ``0 ... 0``.  A query id met again writes over the head of the list
    its first occurrence made and inherits the rest.  One line per rank: qid Q0 pid rank 200-rank score ance."""
    k = args.top_k
    ranking = {}                                              # qid -> k (pid, score) slots, in first-seen order of the qids
    table = _pid_table(offset2pid)
    for row in range(len(retrieved_pid_mat)):
        scores = retrieved_scores_mat[row][:k].tolist()       # python floats: their repr is what lands in the file
        pids = _pids_of(table, retrieved_pid_mat[row][:k])
        first = np.sort(np.unique(np.asarray(pids), return_index=True)[1])
        slots = ranking.setdefault(query_embedding2id[row], [(0, 0)] * k)
        slots[:len(first)] = [(pids[i], scores[i]) for i in first]
    output_trec_file = os.path.join(args.qrel_output_path, args.output_trec_file)
    with open(output_trec_file, "w") as out:
        out.writelines(f"{qid} Q0 {pid} {rank} {200 - rank} {score} ance\n"
                       for qid, slots in ranking.items() for rank, (pid, score) in enumerate(slots, start=1))
    logger.info("output file write ok at %s", output_trec_file)
    if evaluate is None:
        if not getattr(args, "trec_gold_qrel_file_path", None):
            return output_trec_file
        evaluate = print_trec_res
    return evaluate(output_trec_file, args.trec_gold_qrel_file_path, getattr(args, "rel_threshold", 1))   # :284


def _ranked(run_q):
    """trec_eval's ranking of one query's run: score descending, ties by document id descending."""
    return [p for p, _ in sorted(run_q.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)]


def print_trec_res(run_file, qrel_file, rel_threshold=1):
    """Mirror of print_trec_res (:288-353): MRR, NDCG@3, Recall@10, Recall@100 in percent, rounded to 5
    places, averaged over the queries that have both a run and a judgement (pytrec_eval's behaviour).
    The run's score column is the integer ``200 - rank`` the reference parses (``int(line[4])``, :319).
    Definitions (trec_eval): recip_rank = 1 / rank of the first relevant document (0 if none);
    recall_k = relevant documents among the first k / relevant documents judged; ndcg_cut_3 = DCG@3 / ideal
    DCG@3 with gain = the graded judgement and discount log2(rank + 1).

    Field separation: test_HAConvDR_topiocqa.py splits on single blanks (``split(" ")``, :299,:318), test_HAConvDR_qrecc.py on
    any whitespace (``split()``; its default qrel file is the tab-separated qrecc_qrel.tsv, :395).  ``split()`` reads both:
    on a single-blank-separated line it yields the same fields (the last one without its newline, which int() ignores)."""
    qrels, qrels_ndcg, runs = {}, {}, {}
    with open(qrel_file) as f:
        for line in f:
            line = line.split()
            if len(line) < 4:
                continue
            query, passage, rel = line[0], line[2], int(line[3])
            qrels_ndcg.setdefault(query, {})[passage] = rel
            qrels.setdefault(query, {})[passage] = 1 if rel >= rel_threshold else 0
    with open(run_file) as f:
        for line in f:
            line = line.split()
            if len(line) < 5:
                continue
            runs.setdefault(line[0], {})[line[2]] = int(line[4])
    mrr, ndcg3, r10, r100 = [], [], [], []
    for query, run_q in runs.items():
        if query not in qrels:
            continue
        ranked = _ranked(run_q)
        rel_q = qrels[query]
        n_rel = sum(rel_q.values())
        first = next((i for i, p in enumerate(ranked) if rel_q.get(p, 0) > 0), None)
        mrr.append(0.0 if first is None else 1.0 / (first + 1))
        for k, acc in ((10, r10), (100, r100)):
            acc.append(sum(rel_q.get(p, 0) for p in ranked[:k]) / n_rel if n_rel else 0.0)
        gains = qrels_ndcg[query]
        dcg = sum(max(gains.get(p, 0), 0) / math.log2(i + 2) for i, p in enumerate(ranked[:3]))
        ideal = sum(g / math.log2(i + 2) for i, g in enumerate(sorted((g for g in gains.values() if g > 0), reverse=True)[:3]))
        ndcg3.append(dcg / ideal if ideal > 0 else 0.0)

    def avg(v):
        return round(sum(v) / len(v) * 100, 5) if v else 0.0
    res = {"MRR": avg(mrr), "NDCG@3": avg(ndcg3), "Recall@10": avg(r10), "Recall@100": avg(r100)}
    logger.info("---------------------Evaluation results:---------------------")
    logger.info(res)
    return res


def gen_metric_score_and_save(args, index, query_embeddings, query_embedding2id, evaluate=None):
    """Mirror of gen_metric_score_and_save(args, index, Q, ids) (:355-372)."""
    from .search import search_one_by_one
    retrieved_scores_mat, retrieved_pid_mat = search_one_by_one(args, args.passage_embeddings_dir_path, index,
                                                                query_embeddings, args.top_k)
    with open(args.passage_offset2pid_path, "rb") as f:
        offset2pid = pickle.load(f)
    return output_test_res(query_embedding2id, retrieved_scores_mat, retrieved_pid_mat, offset2pid, args, evaluate)
