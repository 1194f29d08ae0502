"""Query-embedding loop: host-side mirror of ``get_test_query_embedding``'s hot loop
(src/test_HAConvDR_topiocqa.py:186-219 = src/test_HAConvDR_qrecc.py:186-219).

The reference builds tokenizer + dataset + DataLoader (CPU string work, out of scope) and then, per
batch of ``per_gpu_test_batch_size`` (default 4) conversations, moves ids/mask to the GPU, runs the
model and synchronises with ``.detach().cpu().numpy()`` (:206-212).  The mirror keeps the contract —
same batch dicts in, ``(embeddings float32 [nq,768], embedding2id list)`` out, same ``test_type`` → key
mapping (:192-207) — but accumulates batches into one large encoder call and copies back once.
"""
import numpy as np

_KEYS = {  # test_type -> (ids key, mask key)   (:192-207)
    "rewrite": ("bt_rewrite", "bt_rewrite_mask"),
    "raw": ("bt_raw_query", "bt_raw_query_mask"),
    "convq": ("bt_conv_q", "bt_conv_q_mask"),
    "convqa": ("bt_conv_qa", "bt_conv_qa_mask"),
    "convqp": ("bt_conv_qp", "bt_conv_qp_mask"),
}


def get_test_query_embedding(model, test_loader, test_type="convqa", device=0, max_queries_per_call=1024):
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
    return np.concatenate(embeddings, axis=0), embedding2id             # :216
