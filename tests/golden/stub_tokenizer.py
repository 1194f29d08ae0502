"""A deterministic stand-in for the RoBERTa BPE tokenizer (whose vocabulary files are not available offline), used
on BOTH sides of the query-construction goldens: the reference's datasets (make_golden_queries.py) and the build's
mirror (tests).  The construction logic under test only sees ``encode``'s output lists, so any tokenizer pins it.

encode(): one id per whitespace-separated word (a stable hash into [3, 50265)), wrapped in <s> = 0 ... </s> = 2 when
add_special_tokens; with max_length the sequence is cut to that many tokens keeping the closing </s> (what
transformers 4.2.0, the reference's pin, does for a bare max_length as well as for truncation=True)."""
import zlib


class StubTokenizer:
    bos_token_id, eos_token_id, pad_token_id = 0, 2, 1

    def __init__(self):
        self.calls = []

    def encode(self, text, add_special_tokens=False, max_length=None, truncation=None):
        self.calls.append((text, add_special_tokens, max_length, truncation))
        ids = [3 + zlib.crc32(w.encode("utf-8")) % 50262 for w in text.split()]
        if add_special_tokens:
            ids = [self.bos_token_id] + ids + [self.eos_token_id]
        if max_length is not None and len(ids) > max_length:
            ids = ids[:max_length - 1] + [ids[-1]] if add_special_tokens else ids[:max_length]
        return ids
