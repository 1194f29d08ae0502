"""Result post-processing: host-side mirror of ``output_test_res`` and ``gen_metric_score_and_save``
(src/test_HAConvDR_topiocqa.py:222-286, :355-372).

Writes the same TREC run file, line for line: ``"{qid} Q0 {pid} {rank} {200-rank} {score} ance"``
(:282).  Quirks of the reference that are kept because downstream files depend on them:
per-query passage-id de-duplication leaves the unused tail slots at ``(0, 0)`` (:243-255), and a
query id that occurs twice reuses the first occurrence's list (:241-246).
The pytrec_eval metrics (:288-353) are not restated: that package is absent here, and scoring a
run file is not on the accelerated path; pass ``evaluate=`` to plug in any scorer.
"""
import logging
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
    if evaluate is not None:
        return evaluate(output_trec_file, args.trec_gold_qrel_file_path, getattr(args, "rel_threshold", 1))
    return output_trec_file


def gen_metric_score_and_save(args, index, query_embeddings, query_embedding2id, evaluate=None):
    """Mirror of gen_metric_score_and_save(args, index, Q, ids) (:355-372)."""
    from .search import search_one_by_one
    retrieved_scores_mat, retrieved_pid_mat = search_one_by_one(args, args.passage_embeddings_dir_path, index,
                                                                query_embeddings, args.top_k)
    with open(args.passage_offset2pid_path, "rb") as f:
        offset2pid = pickle.load(f)
    return output_test_res(query_embedding2id, retrieved_scores_mat, retrieved_pid_mat, offset2pid, args, evaluate)
