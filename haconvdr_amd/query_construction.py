"""Conversational query construction: host-side mirror of the test-time datasets the query
encoder is fed from — ``padding_seq_to_same_length`` (src/data.py:8-23), ``Retrieval_topiocqa`` (src/data.py:25-251)
and ``Retrieval_qrecc`` (src/data.py:381-506), as src/test_HAConvDR_{topiocqa,qrecc}.py:175-179 use them.

What they compute, over token-id lists the tokenizer returns (``tokenizer.encode(text, add_special_tokens=True,
max_length=..., truncation=...)`` is called with exactly the reference's arguments, so any tokenizer gives the same ids
on both sides):

  * TopiOCQA (``bt_conv_qp``): ``<s>current question</s>`` followed, for the earlier turns the relevance labels select
    (``use_PRL``: the turns labelled 1; otherwise every turn), newest first, by that turn's gold passage and its
    question, then by the conversation history itself (answer, question, … newest first).
  * QReCC (``bt_conv_qa``): ``<s>current question</s>`` followed by (``use_PRL``) answer + question of every relevant
    earlier turn, or by the history utterances newest first.

A piece that does not fit into ``max_concat_length`` is cut so that its own last token (the separator) still ends the
sequence, and construction stops; the result is padded with 0 / masked to ``max_concat_length``
(fully padded batches are what the reference feeds its encoder: SURVEY.md §8 a4).  Collated batches are the dicts
``haconvdr_amd.queries.get_test_query_embedding`` consumes.

Only evaluation-time construction is mirrored (``is_train`` false, ``is_PRF`` false = the defaults of the two test
scripts, :400-401): negatives sampling for training and pseudo-relevance feedback are outside the accelerated path and
raise NotImplementedError rather than being half-done.
"""
import json


def padding_seq_to_same_length(input_ids, max_pad_length, pad_token=0):
    """(ids cut or padded to max_pad_length, attention mask = ones over the kept tokens) — src/data.py:8-23."""
    kept = list(input_ids[:max_pad_length])
    fill = max_pad_length - len(kept)
    return kept + [pad_token] * fill, [1] * len(kept) + [0] * fill


def _take(concat, piece, limit):
    """Append ``piece`` to ``concat`` under the length budget ``limit``.  Returns True if it fitted whole; otherwise the
    head of the piece that still fits (less one slot) plus the piece's LAST token is appended — the sequence must end
    with the separator — and False tells the caller to stop (src/data.py:74-76, 134-139, 435-439: same slice
    expression, including its behaviour once the budget is already exhausted)."""
    if len(concat) + len(piece) > limit:
        concat += piece[:limit - len(concat) - 1] + [piece[-1]]
        return False
    concat.extend(piece)
    return True


def _require_eval_mode(args, who):
    if getattr(args, "is_train", False):
        raise NotImplementedError(f"{who}: training-time example construction (negative sampling) is not part of the accelerated path")
    if getattr(args, "is_PRF", False):
        raise NotImplementedError(f"{who}: pseudo-relevance-feedback expansion (is_PRF) is not part of the accelerated path")


def _to_long_tensors(collated, keep_as_is):
    import torch
    for key, value in collated.items():
        if key not in keep_as_is:
            collated[key] = torch.tensor(value, dtype=torch.long)
    return collated


class _Examples:
    def __len__(self):
        return len(self.examples)

    def __getitem__(self, item):
        return self.examples[item]


class Retrieval_topiocqa(_Examples):
    """One example per line of the TopiOCQA test file: [sample_id, question ids, mask, conv_qp ids, mask, [], [], [], []]
    (the four empty lists are the document fields training fills in)."""

    def __init__(self, args, tokenizer, filename, collection=None):
        _require_eval_mode(args, "Retrieval_topiocqa")
        with open(filename, encoding="utf-8") as f:
            lines = f.readlines()
        limit = args.max_concat_length
        self.examples = []
        for i, line in enumerate(lines):
            record = json.loads(line)
            history = record["cur_utt_text"].strip().split(" [SEP] ")       # q1, a1, q2, a2, ..., current question
            question_text, history = history[-1], history[:-1]
            labels = record["rel_label"]
            # fields the reference reads (a missing one is the reference's KeyError too)
            record["last_response"], record["pos_docs"][0], record["pos_docs_pids"][0]

            question = tokenizer.encode(question_text, add_special_tokens=True, max_length=args.max_query_length)
            conv_qp = list(question)

            def earlier_turn(index):            # the record of the turn whose label is rel_label[index]
                return json.loads(lines[i - (len(labels) - index)])

            if args.use_PRL and 1 in labels:
                chosen = [t for t in range(len(labels) - 1, -1, -1) if labels[t] == 1]
            elif not args.use_PRL:
                chosen = list(range(len(labels) - 1, -1, -1))
            else:
                chosen = []
            for t in chosen:                    # newest relevant turn first: its gold passage, then its question
                turn = earlier_turn(t)
                passage = tokenizer.encode(turn["pos_docs"][0], add_special_tokens=True, max_length=args.max_doc_length)
                if not _take(conv_qp, passage, limit):
                    break
                asked = tokenizer.encode(turn["cur_utt_text"].strip().split(" [SEP] ")[-1], add_special_tokens=True,
                                         max_length=args.max_query_length)
                if not _take(conv_qp, asked, limit):
                    break

            # the history itself, newest utterance first.  The reference walks it for three concatenations at once
            # (q-only, q+a, q+passages) and leaves the loop as soon as the q+a one overflows — which also ends the
            # q+passages one at that utterance (src/data.py:128-144); conv_qa is therefore tracked although unused.
            conv_qa = []
            for j in range(len(history) - 1, -1, -1):
                budget = args.max_response_length if j % 2 == 1 else args.max_query_length
                utt = tokenizer.encode(history[j], add_special_tokens=True, max_length=budget, truncation=True)
                if not _take(conv_qa, utt, limit):
                    break
                if not _take(conv_qp, utt, limit):
                    break

            question, question_mask = padding_seq_to_same_length(question, max_pad_length=args.max_query_length)
            conv_qp, conv_qp_mask = padding_seq_to_same_length(conv_qp, max_pad_length=limit)
            self.examples.append([record["sample_id"], question, question_mask, conv_qp, conv_qp_mask, [], [], [], []])

    @staticmethod
    def get_collate_fn(args):
        names = ("bt_sample_ids", "bt_raw_query", "bt_raw_query_mask", "bt_conv_qp", "bt_conv_qp_mask",
                 "bt_pos_docs", "bt_pos_docs_mask", "bt_neg_docs", "bt_neg_docs_mask")

        def collate_fn(batch):
            collated = {name: [example[k] for example in batch] for k, name in enumerate(names)}
            return _to_long_tensors(collated, {"bt_sample_ids", "bt_cur_utt_text", "bt_oracle_utt_text"})

        return collate_fn


class Retrieval_qrecc(_Examples):
    """One example per QReCC test line that has a gold passage: [sample_id, conv_qa ids, mask, [], [], [], []]."""

    def __init__(self, args, tokenizer, filename, collection=None):
        _require_eval_mode(args, "Retrieval_qrecc")
        with open(filename, encoding="utf-8") as f:
            lines = f.readlines()
        limit = args.max_concat_length
        self.examples = []
        for i, line in enumerate(lines):
            record = json.loads(line)
            history, labels = record["ctx_utts_text"], record["rel_label"]      # [q1, a1, q2, a2, ...]
            record["cur_response_text"]
            if len(record["pos_docs_text"]) == 0:
                continue                                                         # no gold passage: not evaluated (:400-401)
            conv_qa = list(tokenizer.encode(record["cur_utt_text"], add_special_tokens=True, max_length=args.max_query_length))
            if args.use_PRL:
                # answer + question of every relevant earlier turn, newest first; no length budget here — the padding
                # step cuts the tail (src/data.py:414-427)
                for t in range(len(labels) - 1, -1, -1):
                    if labels[t] == 0:
                        continue
                    turn = json.loads(lines[i - (len(labels) - t)])
                    asked = tokenizer.encode(turn["cur_utt_text"], add_special_tokens=True, max_length=args.max_query_length, truncation=True)
                    if len(turn["cur_response_text"]) > 0:
                        conv_qa.extend(tokenizer.encode(turn["cur_response_text"], add_special_tokens=True,
                                                        max_length=args.max_response_length, truncation=True))
                    conv_qa.extend(asked)
            else:
                for j in range(len(history) - 1, -1, -1):
                    budget = args.max_response_length if j % 2 == 1 else args.max_query_length
                    utt = tokenizer.encode(history[j], add_special_tokens=True, max_length=budget, truncation=True)
                    if not _take(conv_qa, utt, limit):
                        break
            conv_qa, conv_qa_mask = padding_seq_to_same_length(conv_qa, max_pad_length=limit)
            self.examples.append([record["sample_id"], conv_qa, conv_qa_mask, [], [], [], []])

    @staticmethod
    def get_collate_fn(args):
        names = ("bt_sample_ids", "bt_conv_qa", "bt_conv_qa_mask", "bt_pos_docs", "bt_pos_docs_mask", "bt_neg_docs", "bt_neg_docs_mask")

        def collate_fn(batch):
            collated = {name: [example[k] for example in batch] for k, name in enumerate(names)}
            return _to_long_tensors(collated, {"bt_sample_ids"})

        return collate_fn
