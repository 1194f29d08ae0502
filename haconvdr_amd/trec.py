"""Result post-processing: host-side mirror of ``output_test_res`` and ``gen_metric_score_and_save``
(src/test_HAConvDR_topiocqa.py:222-286, :355-372).

Writes the same TREC run file, line for line: ``"{qid} Q0 {pid} {rank} {200-rank} {score} ance"``
(:282).  Quirks of the reference that are kept because downstream files depend on them:
per-query passage-id de-duplication leaves the unused tail slots at ``(0, 0)`` (:243-255), and a
query id that occurs twice reuses the first occurrence's list (:241-246).
``print_trec_res`` restates the metric block (:288-353) with trec_eval's published definitions, because
``pytrec_eval`` is not installable here: **parity unpinned** for it (no reference output to compare with; the
unit test is hand-worked).  Pass ``evaluate=`` to plug in pytrec_eval where it exists.
"""
import logging
import math
import os
import pickle

logger = logging.getLogger(__name__)


def output_test_res(query_embedding2id, retrieved_scores_mat, retrieved_pid_mat, offset2pid, args, evaluate=None):
    qids_to_ranked_candidate_passages = {}
    topN = args.top_k
    for query_idx in range(len(retrieved_pid_mat)):
        seen_pid = set()
        query_id = query_embedding2id[query_idx]
        selected_ann_idx = retrieved_pid_mat[query_idx][:topN]                    # :238
        selected_ann_score = retrieved_scores_mat[query_idx][:topN].tolist()      # :239
        rank = 0
        if query_id not in qids_to_ranked_candidate_passages:
            qids_to_ranked_candidate_passages[query_id] = [(0, 0)] * topN         # :244-246
        for idx, score in zip(selected_ann_idx, selected_ann_score):
            pred_pid = offset2pid[idx]                                            # :250
            if pred_pid not in seen_pid:
                qids_to_ranked_candidate_passages[query_id][rank] = (pred_pid, score)
                rank += 1
                seen_pid.add(pred_pid)
    output_trec_file = os.path.join(args.qrel_output_path, args.output_trec_file)
    with open(output_trec_file, "w") as g:
        for qid, passages in qids_to_ranked_candidate_passages.items():
            for i in range(topN):
                pid, score = passages[i]
                g.write(str(qid) + " Q0 " + str(pid) + " " + str(i + 1) + " " + str(-i - 1 + 200) + " " + str(score) + " ance\n")  # :282
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
    DCG@3 with gain = the graded judgement and discount log2(rank + 1)."""
    qrels, qrels_ndcg, runs = {}, {}, {}
    with open(qrel_file) as f:
        for line in f:
            line = line.split(" ")
            if len(line) < 4:
                continue
            query, passage, rel = line[0], line[2], int(line[3])
            qrels_ndcg.setdefault(query, {})[passage] = rel
            qrels.setdefault(query, {})[passage] = 1 if rel >= rel_threshold else 0
    with open(run_file) as f:
        for line in f:
            line = line.split(" ")
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
