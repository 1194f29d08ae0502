/*
 * haconvdr.h — C ABI of the MI355X-native dense-retrieval hot path for HAConvDR.
 *
 * Drop-in boundary.  The reference has no FFI; its seams are two
 * duck-typed Python objects.  Each entry point below cites the reference call
 * site it replaces (paths are into the upstream repo fengranMark/HAConvDR):
 *
 *   Index   = the object returned by build_faiss_index()
 *             src/test_HAConvDR_topiocqa.py:39-71 (faiss.IndexFlatIP(768) :52,
 *             sharded over GPUs :55-66) and used by search_one_by_one_with_faiss()
 *             :74-162 through .add (:98) / .search (:102) / .reset (:122).
 *   Encoder = model(input_ids, attention_mask) -> float32 [B,768]
 *             src/models.py:39-64 (ANCE.forward/query_emb), called at
 *             src/test_HAConvDR_topiocqa.py:211 and gen_doc_embeddings.py:110.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function
 * returns 0 on success or a non-zero hac_status, never throws; the message of the
 * last failure on the calling thread is hac_last_error().  A handle is not
 * re-entrant; distinct handles may be used from distinct threads.  "_device"
 * variants take HIP device pointers and a hipStream_t (as void*), enqueue work on
 * that stream and return without synchronising; host variants are synchronous
 * like the faiss calls they replace.  There is NO CPU fallback: with no HIP
 * device every call fails with HAC_ERR_HIP.
 *
 * Canonical semantics (shared with oracle/flat_ip_oracle.c): score(q, x) is the
 * k-ordered fp32 fma chain acc = fmaf(x[k], q[k], acc), k = 0..d-1 (what the
 * gfx950 fp32 MFMA computes natively); results are ordered by (score descending,
 * row ascending); NaN scores are never returned; short result lists are padded
 * with D = -FLT_MAX, I = -1 (faiss's convention).
 */
#ifndef HACONVDR_H
#define HACONVDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    HAC_OK = 0,
    HAC_ERR_INVALID = 1,   /* bad argument (null handle, d not a multiple of 32, k out of range, ...) */
    HAC_ERR_HIP = 2,       /* a HIP runtime call failed / no device */
    HAC_ERR_OOM = 3,       /* device or pinned-host allocation failed */
    HAC_ERR_UNSUPPORTED = 4,
    HAC_ERR_INTERNAL = 5   /* the device detected a broken invariant of the library itself (a candidate loop ran past the pass
                              count no legal input reaches): the affected queries' lists are returned EMPTY (-FLT_MAX / -1), never
                              wrong, and the call (host entry points) or hac_index_last_status (after *_device calls) says so */
} hac_status;

#define HAC_MAX_K 2048      /* faiss-gpu 1.7.2 has the same top-k ceiling */
#define HAC_MAX_D 1024

/* ------------------------------------------------------------------ errors */
/* Message of the last failing call on this thread ("" if none).  Never NULL. */
const char *hac_last_error(void);
/* Library version string, e.g. "haconvdr-amd 0.5.0 (gfx950)". */
const char *hac_version(void);


/* ------------------------------------------------------------------- index */
typedef struct hac_index hac_index;

/* faiss.IndexFlatIP(d) [+ index_cpu_to_gpu_multiple(shard=True)]:
 * src/test_HAConvDR_topiocqa.py:52,55-66.  d must be a multiple of 32, <= HAC_MAX_D
 * (the reference fixes d = 768).  device_ids[0..n_dev) are HIP ordinals; with
 * n_dev > 1 the rows of every add() are split contiguously across the devices and
 * search() merges the per-device top-k (faiss IndexShards semantics; the devices are
 * searched concurrently, one host thread per device, rows stay numbered by insertion
 * order over the whole index).  The scalable path is one process per GPU
 * (haconvdr_amd.sharded). */
int hac_index_create(int d, const int *device_ids, int n_dev, hac_index **out);
void hac_index_destroy(hac_index *idx);

/* index.add(passage_embedding) — src/test_HAConvDR_topiocqa.py:98.
 * x: host pointer, float32 row-major [n, d]; copied (and re-tiled for the MFMA
 * scan) before return — the caller may free it (the reference does, :123). */
int hac_index_add(hac_index *idx, const float *x, int64_t n);
/* Same, x already in device memory of the index's (single) device. */
int hac_index_add_device(hac_index *idx, const float *x_dev, int64_t n, void *hip_stream);

/* D, I = index.search(query_embeddings, topN) — src/test_HAConvDR_topiocqa.py:102.
 * q: host float32 [nq, d]; D: host float32 [nq, k]; I: host int64 [nq, k]
 * (rows numbered by insertion order since the last reset, whatever the number of devices).
 * Synchronous.
 *
 * Result contract of every search entry point: score = k-ordered fp32 fma chain, order =
 * (score desc, row asc), NaN scores never returned, short lists padded -FLT_MAX / -1.
 * Where it is faster (k <= 192, d % 64 == 0, d >= 192; at once with >= 48 queries and >= 1e8 (query, row) pairs; for fewer --
 * down to ONE query on >= 750 k rows, >= 17 on >= 500 k, >= 2.5e7 pairs on >= 40 k -- when the fp16 image of the rows exists
 * already, or from the third such search after the last add / reset on, which builds it: +50 % of the corpus in HBM) the
 * rows are first screened on the fp16 matrix pipe under a proven error bound, the few
 * candidates are rescored exactly and each query's list is certified complete; queries
 * that cannot be certified are searched again by the exact fp32 kernels ("Prefilter with a certificate").
 * The results are the same bits either way.  hac_index_set_option(idx, "split", "0") disables
 * the screen, "1" applies it whenever the shape allows (tests), "auto" decides by size.
 * The host entry point reads the certificates back (and may retry many failures with three
 * fp16 products per score before the exact kernels); the *_device entry points leave the
 * decision on the device -- failed queries are compacted, searched again by the exact kernels
 * and scattered back by launches sized for every query and cut down by a device-side count --
 * and never synchronize or read back: they can be captured into a HIP graph once workspaces,
 * the segment table (re-uploaded by the first search after an add / reset, which does
 * synchronize once) and the fp16 image exist.  That image (+50 % of the corpus bytes in HBM)
 * is built by the first search that takes the screen and lives as long as the segment it belongs to (reset() keeps both;
 * destroying the handle frees them); an index that never takes it (small, searched once or twice per add, or split = "0")
 * never allocates it.  A search that is being captured never builds it (nor the rescoring's row-major copy): it runs what
 * the call before it ran.  With split = "auto", when the image does not fit beside the corpus the
 * exact fp32 kernels answer instead (same bits) and no search tries again until the next add / reset; split = "1"
 * reports HAC_ERR_OOM.
 * Before capturing a *_device search into a graph, run one search of the LARGEST shape (nq, k) you will capture: a call that
 * has to grow a workspace allocates, zero-fills it on the null stream and waits for that fill (once per growth).
 * Cost of the device-decided route: per chunk of 1024 queries a re-tiling of the failed queries, two memsets, a full-grid scan,
 * a select and a scatter are launched even when no query failed (their workgroups exit at once), and failed queries go
 * straight to the exact fp32 kernels, without the host route's three-product retry and threshold seeding: on tie-heavy or
 * degenerate corpora, where many certificates fail, hac_index_search (host route) is the faster entry point. */
int hac_index_search(hac_index *idx, const float *q, int64_t nq, int k, float *D, int64_t *I);
/* Device/stream variant (single-device index).  id_map_dev (optional, int64
 * [ntotal]) maps row -> external id, fusing `passage_embedding2id[I]` (:110). */
int hac_index_search_device(hac_index *idx, const float *q_dev, int64_t nq, int k, float *D_dev,
                            int64_t *I_dev, const int64_t *id_map_dev, void *hip_stream);
/* Top-k as packed 64-bit keys, [nq, k] sorted descending, 0 = empty slot:
 *   key = (orderable(score) << 32) | (0xFFFFFFFF - (pos_base + row)).
 * This is the unit exchanged between corpus shards (one all-gather of keys). */
int hac_index_search_keys_device(hac_index *idx, const float *q_dev, int64_t nq, int k,
                                 uint64_t *keys_dev, uint32_t pos_base, void *hip_stream);

/* index.reset() — src/test_HAConvDR_topiocqa.py:122.  Keeps allocations for reuse. */
int hac_index_reset(hac_index *idx);
/* index.ntotal */
int64_t hac_index_ntotal(const hac_index *idx);
/* Device-detected failures of searches enqueued so far (the host entry points report them themselves; the *_device entry
 * points cannot, they never read back).  Waits for the index's own streams only: synchronize YOUR stream first, then call
 * this.  Returns HAC_OK, or HAC_ERR_INTERNAL with the details in hac_last_error(); reading clears the device's error word
 * (hac_index_reset clears it too).  Every scan kernel bounds its candidate-compaction loop by the number of passes the
 * worst legal input needs; a workgroup that reaches the bound sets the word, drops what it still held, and the queries of
 * its tile come back with EMPTY lists: an invariant slip is an error code, not a hang and not wrong bits. */
int hac_index_last_status(hac_index *idx);

/* Tuning and test switches of a live handle: "split" = "0" | "1" | "auto", "split_terms" = "1" | "3",
 * "force_scan16" = "0" | "1", "scanq_nt" = "0".."4", "scanq_waves" = "4" | "8", "scan_no_p8" = "0" | "1",
 * "seed_groups_max" = cap of the prefilter's seeding pass in 64-row groups (integer >= 0; 0 = by size: 14 sqrt(groups),
 * or 1024 groups when the scan is cut into passes),
 * "split_decide" = "auto" (host entry point: host, *_device: device) | "host" | "device": who reads the certificates,
 * "scan_passes" = "auto" | "1" .. "5" (prefilter: passes of the main scan, the thresholds refreshed from everything found so
 * far between them; auto: 3 from 1.6M rows, 4 from 8.4M), "scan_pass_cuts" = "auto" | "a,b" (where the first two passes end, in
 * thousandths of the rows; auto: the first after ~6k 64-row groups, the others in geometric progression towards the corpus), "debug_max_pass" = integer >= 0 (tests: the pass bound of the
 * candidate loops; 0 = the bound no legal input reaches),
 * "scan_halfq" = "1" (default) | "0" (development: a prefilter search of one tile with <= 128 / <= 64 queries runs an instantiation
 * that leaves out the empty query tiles' matrix work -- "tiles=half" / "tiles=quarter" in the plan text; "0" pins the full-tile form; same bits),
 * "fp16_image" = "lazy" (default) | "eager": when the fp16 image of the rows (what the prefilter streams) is made: by the first
 * search that wants it, or by add() while it tiles the rows of a new segment (a resident index that will serve few queries per call
 * is then fast from its first search; best effort, +50 % of the corpus in HBM either way),
 * "rescore_rows" = "auto" | "0" | "1": the prefilter's exact rescoring reads ~130 scattered rows per query; out of the T64 tiles
 * that is one useful 16-byte piece per 64-byte sector.  Small indexes (auto: <= 12M rows) therefore keep their rows once more,
 * row-major, for the rescoring alone (+100 % of a small corpus, built lazily by the THIRD prefilter search after the last add /
 * reset of an index that has none -- an index searched once per block, the reference's add / search / reset loop, never pays for
 * it --, kept current across later adds, best effort: when the allocation fails the tiles are read as before); "1" builds it with the
 * first such search at any size, "0" never (and frees one that exists, as does growing past the size).  Same bits either way.
 * Any other name or value is HAC_ERR_INVALID (never a silent default).
 * The HAC_<NAME> environment variables give the defaults and are read once, in hac_index_create. */
int hac_index_set_option(hac_index *idx, const char *name, const char *value);

/* Profiling aid for bench.py: when enabled, the main scan kernel of every search is
 * bracketed by a hipEvent pair recorded on the launch stream (no host sync).
 * hac_index_profile_drain() waits for the recorded pairs, writes up to cap kernel
 * durations (ms, in launch order) to ms_out, sets *n_out and clears the record. */
int hac_index_set_profiling(hac_index *idx, int enable);
int hac_index_profile_drain(hac_index *idx, float *ms_out, int cap, int *n_out);
/* Human-readable description of the kernel configuration of the most recent search
 * (kernel name, grid, queries per workgroup, candidate slots, LDS bytes). */
const char *hac_index_last_plan(const hac_index *idx);

/* ------------------------------------------------------- top-k list merging */
/* K9: merge L per-block / per-shard key lists into one.  lists_dev: uint64
 * [L, nq, k] (each [nq,k] slab sorted descending, 0 = empty); out_dev: [nq, k].
 * Equals the reference's sequential `>=` block merge (:131-149) when slabs are in
 * block order and positions increase with the block index. */
int hac_merge_keys_device(int device, const uint64_t *lists_dev, int n_lists, int64_t nq, int k,
                          uint64_t *out_dev, void *hip_stream);
/* keys -> (D float32, I int64); id_map_dev optional (position -> external id). */
int hac_keys_to_results_device(int device, const uint64_t *keys_dev, int64_t n_keys,
                               const int64_t *id_map_dev, float *D_dev, int64_t *I_dev,
                               void *hip_stream);

/* ----------------------------------------------------------------- encoder */
/* model(input_ids, attention_mask) -> float32 [B,768]: ANCE.forward, src/models.py:39-64
 * (RobertaModel -> last_hidden_state[:,0] -> embeddingHead -> norm), called at
 * src/test_HAConvDR_topiocqa.py:211 and gen_doc_embeddings.py:110.  RoBERTa-base geometry. */
typedef struct hac_encoder hac_encoder;
typedef struct {
    int n_layers;      /* 12 for ANCE */
    int hidden;        /* 768 */
    int n_heads;       /* 12 */
    int ffn;           /* 3072 */
    int vocab;         /* 50265 */
    int max_pos;       /* 514 */
    int type_vocab;    /* 1 */
    int pad_token_id;  /* 1: position ids are cumsum(id != pad)*(id != pad) + pad (HF RoBERTa) */
    float ln_eps;      /* 1e-5 */
} hac_encoder_config;

int hac_encoder_create(const hac_encoder_config *cfg, int device, hac_encoder **out);
void hac_encoder_destroy(hac_encoder *enc);
/* One tensor of the checkpoint, by its state-dict name (ANCE.from_pretrained keys, :170):
 * "roberta.embeddings.*", "roberta.encoder.layer.N.*", "embeddingHead.*", "norm.*"
 * ("classifier.*" exists in the checkpoint but is unused by forward: do not pass it).
 * data: host float32, copied before return. */
int hac_encoder_set_weight(hac_encoder *enc, const char *name, const float *data, size_t count);
/* Checks that every tensor is present and packs the GEMM weights to bf16. */
int hac_encoder_finalize(hac_encoder *enc);
/* ids, mask: host int32 [B, L]; mask must be a prefix mask (first len >= 1 ones) as both
 * reference pipelines produce (gen_doc_embeddings.py:38-40, src/data.py:8-23), attended ids in
 * [0, vocab); out: host float32 [B, 768].  Synchronous.  HAC_ERR_INVALID names the first bad sequence. */
int hac_encoder_forward(hac_encoder *enc, const int32_t *ids, const int32_t *mask, int B, int L, float *out);
/* Device/stream variant: ids/mask device pointers of elem_bytes 4 (int32) or 8 (int64, what the
 * reference passes); out_dev float32 [B,768].  Work is enqueued on hip_stream.  A batch of more than
 * 524288 padded rows (B * roundup(L,32); option "max_tokens") runs as several sub-batches sized by the real lengths: that
 * costs ONE read-back of B ints, i.e. the call synchronizes hip_stream once (not capturable); smaller
 * batches never synchronize.  Errors the device finds are reported per sequence: a mask that is not
 * a non-empty prefix mask, or an attended token id outside [0, vocab) (nn.Embedding would raise),
 * yields a NaN row for THAT sequence only; the host variant turns such a row into HAC_ERR_INVALID. */
int hac_encoder_forward_device(hac_encoder *enc, const void *ids_dev, const void *mask_dev, int elem_bytes,
                               int B, int L, float *out_dev, void *hip_stream);
/* Tuning and test switches of a live handle: "gemm" = "auto" (by batch size) | "classic" (128^2 / 256^2 two-stage
 * kernels with separate LayerNorm passes) | "8phase" (the large-batch ping-pong kernel with folded LayerNorms whenever
 * the batch has a whole 256-row tile); "attn" = "stream" (persistent single-pass attention kernels, the default) |
 * "twopass" (one workgroup per (sequence, head), exact row maxima first: the cross-check of the tests);
 * "graph" = "auto" (default) | "off": batches of at most 16384 padded rows (the reference's own call shape is 4 queries per GPU,
 * src/test_HAConvDR_topiocqa.py:173,406: ~110 launches for ~0.4 TFLOP) are captured ONCE per (B, L, elem_bytes, options) into a HIP
 * graph over private input / output buffers and replayed -- per call one graph launch and three small device copies; the first call
 * of a shape runs plain launches (it sizes the workspaces), the second captures, later ones replay; results are the same bits as
 * with "off" (at most 64 shapes are kept per encoder; beyond that the cache starts over).  Not used while profiling is on or while
 * the caller's stream is itself capturing;
 * "ksplit" = "auto" (default) | "off": with few rows (the classic 128^2 kernels) the two residual GEMMs of a layer split their
 * K loop over 2..6 work items per tile where that pays (by a measured cost model of rows and K), the partial sums being added in a fixed order by
 * the LayerNorm pass behind them: deterministic, but the summation order -- hence the last bits of an embedding -- then depends
 * on the batch's row count (as it does between the "gemm" families); "off" restores one summation order for every small batch;
 * "ksplit_pin" = "a/b" (development, tools/ks_sweep.py: the slices of out-proj / FFN-down pinned; "0/0" = by the model);
 * "attn_qs_pin" = "0" | "1" | "2" | "4" | "8" | "16" (development, tools/opt_sweep.py: the streaming attention's query split pinned; "0" = by the rule);
 * "max_tokens" = packed rows per sub-batch (integer >= 4096); "g8_split" = bit mask 0..15 (development: which kernel
 * classes -- bit 0 QKV, 1 out-proj, 2 FFN-up, 3 FFN-down -- run the operand-split loop of the large-batch GEMM, default
 * 15; 0 = round 2's loop: same results bit for bit); "attn_qsplit" = "auto" (default) | "off" (with few sequences the streaming
 * attention kernel deals the query rows of a (sequence, head) item to 2..16 workgroups and runs both length classes in one launch;
 * same bits); "attn_pipe" = "auto" (default) | "off" | "all" (whole (sequence, head) items -- no query split, not the <s>-only last layer -- of sequences
 * longer than 256 rows go through the kernel with two query blocks per wave, the softmax of one woven into the MFMAs of the
 * other, followed by a fix-up pass of the one-block kernel over the items whose softmax reference has to move; in "auto" a layer whose items mostly needed
 * that pass in an earlier forward goes through the one-block kernel directly, with a retry every 64th forward; "off": the one-block kernel
 * everywhere; "all" (tests): the woven form for every whole item of either length class, always; same bits); "g8_stagger" = "auto" (default) | "off" (the workgroups of the large-batch
 * QKV and out-projection GEMMs start in four phases, one per pair of XCDs, so that their epilogues do not reach HBM all at once;
 * timing only, same bits).  Any other name or value is HAC_ERR_INVALID (never a
 * silent default).  HAC_ENC_GEMM gives the default of "gemm" and is read once, in hac_encoder_create.
 * "auto" decides ONCE per forward call, from the rows of the whole batch: every sub-batch of a call runs the same
 * GEMM family and tile size, so a sequence's embedding does not depend on the sub-batch it fell into. */
int hac_encoder_set_option(hac_encoder *enc, const char *name, const char *value);
/* What the most recent forward ran: "gemm=gemm8|classic256|classic128 attn=stream|twopass sub_batches=N rows=R graph=off|eager-first|replay ksplit=S_out/S_down attn_form=woven|single|twopass"
 * (tests and bench.py assert the kernel family they mean to check).  The GEMM family and, with the classic kernels, the tile
 * size are chosen once per call, from the rows of the whole batch: every sub-batch runs the same kernels. */
const char *hac_encoder_last_plan(hac_encoder *enc);

/* Profiling aid for bench.py: hipEvent pairs recorded on the launch stream (no host sync).  mask bit 0:
 * around the layer stack of each forward (sub-batch); bit 1+c: around every launch of kernel class c.
 * hac_encoder_profile_drain / _drain_class wait for the recorded pairs, write up to cap durations (ms, in
 * launch order) and clear the record. */
enum {
    /* kernels: large batches gemm8_kernel<EPI8_*> (gemm8.inc), small ones gemm_bf16_nt_kernel<EPI_*> */
    HAC_ENC_CLASS_QKV = 0,       /* <EPI8_QKV | EPI_QKV>:     [T,768] x [2304,768]^T (+ folded LayerNorm) + bias, Q scale, V regrouping */
    HAC_ENC_CLASS_ATTN = 1,      /* attention_stream_kernel<16|8> (both launches of a layer) */
    HAC_ENC_CLASS_OUTPROJ = 2,   /* <EPI8_RESID | EPI_RESID>: [T,768] x [768,768]^T + bias + residual (+ LayerNorm statistics) */
    HAC_ENC_CLASS_FFN_UP = 3,    /* <EPI8_GELU | EPI_GELU>:   [T,768] x [3072,768]^T (+ folded LayerNorm) + bias + erf GELU */
    HAC_ENC_CLASS_FFN_DOWN = 4,  /* <EPI8_RESID | EPI_RESID>: [T,3072] x [768,3072]^T + bias + residual (+ LayerNorm statistics) */
    HAC_ENC_CLASS_LN = 5,        /* LayerNorm passes that are their own kernel (ln_combine_kernel / ln_stats_rows_kernel) */
    HAC_ENC_NCLASS = 6
};
int hac_encoder_set_profiling(hac_encoder *enc, int mask);
int hac_encoder_profile_drain(hac_encoder *enc, float *ms_out, int cap, int *n_out);
int hac_encoder_profile_drain_class(hac_encoder *enc, int cls, float *ms_out, int cap, int *n_out);
/* The shader clock the part sustained inside the most recent large-batch FFN-up launch made while class profiling was on:
 * workgroup 0 of gemm8_kernel<EPI8_GELU> reads the shader-clock counter (s_memtime) and the constant 100-MHz counter
 * (s_memrealtime) at its first and last instruction (same CU both times: the counters of different XCDs are not aligned).
 * out[0] = shader clocks, out[1] = 100-MHz ticks between the two readings (0, 0 if no such launch happened); waits for the
 * stream of that launch.  clock = out[0] / out[1] x 100 MHz -- the figure fractions of a peak are normalised by, because boxes
 * of one pool hold different clocks under the same load.  Costs the kernel nothing (two scalar reads, four stores). */
int hac_encoder_last_clock(hac_encoder *enc, uint64_t out[2]);
/* Test aid: how often the woven attention kernel (attn_pipe.inc) handed an item to the fix-up pass in the most recent forward
 * (last sub-batch) -- a wave counts its item when one of its rows left the window in which a row's softmax reference stays 0
 * (a score above 2^63 relative to 1, or a first key block wholly below 2^-64); several waves of one item may count it: an upper
 * bound of the items, 0 iff none.  Waits for the device.  The results do not depend on it: the fix-up pass is the one-block
 * kernel, whose outputs the woven kernel's equal bit for bit wherever it does not flag. */
int hac_encoder_attention_redo(hac_encoder *enc, long long *items_out);

#ifdef __cplusplus
}
#endif
#endif /* HACONVDR_H */
