"""Build-owned deterministic synthetic data.

A counter-based generator (splitmix64 over ``seed``/``index``) whose every step is
integer arithmetic or a correctly-rounded IEEE basic operation (+, *, /, sqrt), so
the same seed yields the same bits on any host — fixtures under ``tests/golden``
store seeds, not megabytes.  No transcendental functions, no numpy Generator.

* ``embeddings``: rows shaped like the output of the ANCE head's LayerNorm
  (/root/reference/src/models.py:44 with unit affine): approximately Gaussian
  entries (Irwin–Hall, 12 uniform 16-bit terms), each row standardised to zero
  mean / unit population variance, so ‖x‖ = √d (27.71 for d = 768).
* ``normal``: N(0, std²) weights, the reference's init (models.py:37, std 0.02).
* ``token_batch``: token ids / lengths in the layout both reference pipelines
  use (``<s>`` … ``</s>``, padded with token id 0: gen_tokenized_doc.py:18,242;
  src/data.py:8).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream(seed, n, lane=0):
    """n 64-bit words of the stream (seed, lane)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(lane) * np.uint64(0xD6E8FEB86659FD93)))
        ctr = np.arange(n, dtype=np.uint64)
        return _splitmix64(base + ctr * np.uint64(0x9E3779B97F4A7C15))


def uniform_u32(seed, n):
    """n uint32 values."""
    return (_stream(seed, n) >> np.uint64(32)).astype(np.uint32)


def _irwin_hall12(seed, n):
    """Sum of twelve independent 16-bit uniforms per element, as int64 (mean 393210, sd ≈ 65536)."""
    acc = np.zeros(n, np.int64)
    for lane in range(3):
        w = _stream(seed, n, lane + 1)
        for s in (0, 16, 32, 48):
            acc += ((w >> np.uint64(s)) & np.uint64(0xFFFF)).astype(np.int64)
    return acc


def normal(seed, shape, std=1.0):
    """float32 N(0, std²) (Irwin–Hall approximation), reproducible bit-for-bit."""
    n = int(np.prod(shape))
    g = _irwin_hall12(seed, n)
    return (((g - 393210).astype(np.float64) / 65536.0) * float(std)).astype(np.float32).reshape(shape)


def embeddings(seed, n, d=768, chunk=1 << 16):
    """float32 [n, d] row-standardised pseudo-Gaussian vectors, ‖row‖ = √d."""
    out = np.empty((n, d), np.float32)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        m = hi - lo
        g = _irwin_hall12(seed + 0x51ED27 * (lo // chunk + 1), m * d).reshape(m, d)
        s = g.sum(1, dtype=np.int64)[:, None]
        ss = (g * g).sum(1, dtype=np.int64)[:, None]
        num = (d * g - s).astype(np.float64)
        den = np.sqrt((d * ss - s * s).astype(np.float64))
        out[lo:hi] = (num / den).astype(np.float32)
    return out


def token_batch(seed, batch, max_len, min_len=8, vocab=50265, fixed_len=None):
    """(ids int32 [B, L] padded with 0, lens int32 [B]).  ids[:,0]=0 (<s>), ids[:,len-1]=2 (</s>),
    body tokens uniform in [3, vocab)."""
    u = uniform_u32(seed, batch * max_len).reshape(batch, max_len)
    ids = (u % np.uint32(vocab - 3)).astype(np.int32) + 3
    if fixed_len is None:
        lens = (uniform_u32(seed + 1, batch) % np.uint32(max_len - min_len + 1)).astype(np.int32) + min_len
    else:
        lens = np.full(batch, fixed_len, np.int32)
    pos = np.arange(max_len)[None, :]
    ids[:, 0] = 0
    ids[pos == (lens[:, None] - 1)] = 2
    ids[pos >= lens[:, None]] = 0
    return ids, lens
