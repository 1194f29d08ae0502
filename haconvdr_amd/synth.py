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


def normal_fast(seed, shape, std=1.0):
    """float32 pseudo-normal from FOUR 16-bit uniforms of one splitmix64 word (Irwin–Hall n=4,
    bounded at ±3.46 sd).  3x cheaper than ``normal``; used for the 125 M synthetic encoder weights."""
    n = int(np.prod(shape))
    w = _stream(seed, n, 7)
    acc = np.zeros(n, np.int64)
    for s in (0, 16, 32, 48):
        acc += ((w >> np.uint64(s)) & np.uint64(0xFFFF)).astype(np.int64)
    # mean 4*32767.5 = 131070, sd = 65536/sqrt(3)
    return (((acc - 131070).astype(np.float64) * (np.sqrt(3.0) / 65536.0)) * float(std)).astype(np.float32).reshape(shape)


def _name_seed(seed, name):
    h = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        for ch in name.encode():
            h = _splitmix64(h ^ np.uint64(ch))
    return int(h)


def ance_state_dict(seed=0xA11CE, n_layers=12, hidden=768, ffn=3072, vocab=50265, max_pos=514, rich=True, layer_matrix_std=0.02):
    """Synthetic weights with the reference checkpoint's key names (``roberta.*``,
    ``embeddingHead.*``, ``norm.*``): dict name -> float32 ndarray.

    Matrices ~ N(0, 0.02²) (the reference's init, src/models.py:37).  rich=True also randomises
    biases (sd 0.02) and LayerNorm affine (gamma 1 ± 0.1, beta sd 0.05) so that every parameter
    influences the output — used by the parity tests; rich=False is the survey's bench recipe
    (biases 0, LN identity).

    layer_matrix_std: standard deviation of the six matrices of every encoder layer (embeddings and
    head keep 0.02).  At 0.02 a random RoBERTa maps all inputs onto nearly one direction (pairwise
    1−cos between different sequences' embeddings 1e-4 … 2e-3): parity asserts on such outputs cannot
    tell one sequence from another.  0.08 ("content-sensitive": attention logits of σ ≈ 5, different
    sequences ≥ 0.05 apart in 1−cos) is what the discriminative goldens use."""
    sd = {}

    def mat(name, shape):
        std = layer_matrix_std if name.startswith("roberta.encoder.layer.") else 0.02
        sd[name] = normal_fast(_name_seed(seed, name), shape, std)

    def vec(name, n, kind):
        if not rich:
            sd[name] = np.ones(n, np.float32) if kind == "gamma" else np.zeros(n, np.float32)
        elif kind == "gamma":
            sd[name] = (1.0 + normal_fast(_name_seed(seed, name), (n,), 0.1)).astype(np.float32)
        else:
            sd[name] = normal_fast(_name_seed(seed, name), (n,), 0.05 if kind == "beta" else 0.02)

    p = "roberta.embeddings."
    mat(p + "word_embeddings.weight", (vocab, hidden))
    mat(p + "position_embeddings.weight", (max_pos, hidden))
    mat(p + "token_type_embeddings.weight", (1, hidden))
    vec(p + "LayerNorm.weight", hidden, "gamma")
    vec(p + "LayerNorm.bias", hidden, "beta")
    for i in range(n_layers):
        q = f"roberta.encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            mat(q + f"attention.self.{nm}.weight", (hidden, hidden))
            vec(q + f"attention.self.{nm}.bias", hidden, "bias")
        mat(q + "attention.output.dense.weight", (hidden, hidden))
        vec(q + "attention.output.dense.bias", hidden, "bias")
        vec(q + "attention.output.LayerNorm.weight", hidden, "gamma")
        vec(q + "attention.output.LayerNorm.bias", hidden, "beta")
        mat(q + "intermediate.dense.weight", (ffn, hidden))
        vec(q + "intermediate.dense.bias", ffn, "bias")
        mat(q + "output.dense.weight", (hidden, ffn))
        vec(q + "output.dense.bias", hidden, "bias")
        vec(q + "output.LayerNorm.weight", hidden, "gamma")
        vec(q + "output.LayerNorm.bias", hidden, "beta")
    mat("embeddingHead.weight", (768, hidden))
    vec("embeddingHead.bias", 768, "bias")
    vec("norm.weight", 768, "gamma")
    vec("norm.bias", 768, "beta")
    return sd
