// ANCE / RoBERTa-base encoder forward for gfx950 (MI355X), behind hac_encoder_* (include/haconvdr.h).
//
// Reference path replaced: model(input_ids, attention_mask) — src/models.py:39-64
// (ANCE.forward -> query_emb -> RobertaModel -> [:,0] -> embeddingHead -> norm), called at
// src/test_HAConvDR_topiocqa.py:211 and gen_doc_embeddings.py:110.
//
// Design:
//   * varlen: only the first len_b tokens of a sequence are computed (the reference's output is
//     bit-identical whatever sits in masked positions); sequences are packed back to
//     back, each padded to a multiple of 32 rows so no MFMA tile straddles two sequences.
//   * GEMMs (QKV, attention-out, FFN up/down): bf16 operands, fp32 accumulate on
//     v_mfma_f32_32x32x16_bf16; persistent workgroups over 256x256x64 tiles (128x128 for small
//     batches) in XCD-aware runs; operand tiles arrive by LDS-DMA into a double-buffered
//     XOR-swizzled image (conflict-free ds_read_b128); LDS-transposed 16-byte epilogues (bias, query
//     scale log2(e)/8, V written in 16-key groups, erf GELU, residual add with deferred LayerNorm).
//   * the residual stream, LayerNorm, softmax and the final head stay fp32.
//   * attention: one workgroup = all query rows of one (sequence, head) with K and V resident in LDS
//     (LDS-DMA); S^T = K.Q^T so the query sits on the lane and softmax statistics are lane-local; the
//     S^T accumulator is, register for register, the B operand of O^T = V^T.P^T (no LDS, no shuffles).
#include <type_traits>
#include "hac_common.h"

#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace hac {
namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
#ifndef HAC_GEMM_S16
#define HAC_GEMM_S16 1
#endif

constexpr int H = 768;        // hidden size (RoBERTa-base / ANCE)
constexpr int NH = 12;        // heads
constexpr int DH = 64;        // head dim
constexpr int FF = 3072;      // FFN inner size
constexpr int SEQ_ALIGN = 32; // rows per sequence are padded to this
constexpr int MT = 256, BK = 64;   // packed rows are padded to MT (largest GEMM tile)

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------ sequence bookkeeping
struct SeqInfo {
    int *lens;    // [B] valid tokens
    int *len32;   // [B] rows occupied (multiple of 32)
    int *off;     // [B+1] first packed row of each sequence; off[B] = total rows
    int *pos;     // [B][L] position ids (HF rule), valid for t < len
    int *err;     // [B] per sequence: 1 mask is not a prefix mask, 2 mask is empty, 4 a token id outside [0, vocab)
    int *nb;      // [1] number of sequences (device copy; row count of the CLS-only tail)
    int *order;   // [B] the attention kernels' work list: sequences of 257+ rows, longest first, then the others, longest first
    int *ncls;    // [2] how many of each (empty sequences are in neither)
    int *desc;    // [B][4] (16-byte aligned) the work list's entries in full: (len, len32, first row, sequence) -- one scalar load per item (attn_pipe.inc)
};

// one workgroup per sequence: len = sum(mask), prefix check, HF position ids
//   pos = cumsum(id != pad) * (id != pad) + pad     (pad = 1)
template <typename IT>
__global__ __launch_bounds__(512) void seq_prep_kernel(const IT *__restrict__ ids, const IT *__restrict__ mask, int L, SeqInfo s,
                                                       int pad_id, int vocab) {
    __shared__ int wsum[8];
    __shared__ int wlen[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t = tid;
    int m = 0, np = 0;
    bool id_ok = true;
    if (tid == 0) s.err[b] = 0;
    if (t < L) {
        m = mask[(size_t)b * L + t] != 0;
        const IT id = ids[(size_t)b * L + t];
        np = id != (IT)pad_id;
        id_ok = id >= (IT)0 && id < (IT)vocab;   // nn.Embedding raises on such an id; here the sequence is flagged (NaN row)
    }
    // block reductions / scans over 512 threads (8 waves)
    int mlen = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mlen += __shfl_xor(mlen, o);
    int sc = np;  // inclusive scan within the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(sc, o);
        if (lane >= o) sc += v;
    }
    if (lane == 63) wsum[w] = sc;
    if (lane == 0) wlen[w] = mlen;
    __syncthreads();
    int base = 0, len = 0;
    for (int i = 0; i < 8; ++i) {
        if (i < w) base += wsum[i];
        len += wlen[i];
    }
    const int cum = sc + base;
    if (t < L) {
        s.pos[(size_t)b * L + t] = np ? cum + pad_id : pad_id;
        if ((m != 0) != (t < len)) atomicOr(s.err + b, 1);  // not a prefix mask
        if (t < len && !id_ok) atomicOr(s.err + b, 4);      // only attended tokens are looked up
    }
    if (tid == 0) {
        if (len <= 0) atomicOr(s.err + b, 2);
        s.lens[b] = len;
        s.len32[b] = (len + SEQ_ALIGN - 1) / SEQ_ALIGN * SEQ_ALIGN;
    }
}

__global__ __launch_bounds__(256) void seq_offsets_kernel(SeqInfo s, int B) {
    // exclusive scan of the padded lengths, one workgroup: every thread sums a run of consecutive sequences, the 256 run
    // totals are scanned through LDS (a single thread walking 512 sequences took 0.11 ms per forward)
    __shared__ int tot[256];
    const int tid = threadIdx.x;
    const int per = (B + 255) / 256, lo = min(B, tid * per), hi = min(B, lo + per);
    int sum = 0;
    for (int b = lo; b < hi; ++b) sum += s.len32[b];
    tot[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = tid >= o ? tot[tid - o] : 0;
        __syncthreads();
        tot[tid] += v;
        __syncthreads();
    }
    int acc = tot[tid] - sum;
    for (int b = lo; b < hi; ++b) {
        s.off[b] = acc;
        acc += s.len32[b];
    }
    if (tid == 255) {
        s.off[B] = tot[255];
        *s.nb = B;
    }
}

// The streaming attention kernels are persistent: workgroup g takes items g, g + G, ... of its length class.  With the
// sequences ordered by length inside a class every workgroup gets one item of each length stratum (and the short ones
// last), so the static deal is balanced; a counting sort over the 16 possible padded lengths, one workgroup.
__global__ __launch_bounds__(256) void attn_order_kernel(SeqInfo s, int B) {
    __shared__ int hist[17], start[17];
    const int tid = threadIdx.x;
    if (tid < 17) hist[tid] = 0;
    __syncthreads();
    for (int b = tid; b < B; b += 256) atomicAdd(&hist[min(s.len32[b] >> 5, 16)], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int k = 16; k >= 1; --k) {
            if (k == 8) s.ncls[0] = acc;      // 9..16 blocks of 32 rows: the long class
            start[k] = acc;
            acc += hist[k];
        }
        s.ncls[1] = acc - s.ncls[0];
    }
    __syncthreads();
    if (tid < 17) hist[tid] = 0;
    __syncthreads();
    for (int b = tid; b < B; b += 256) {
        const int k = min(s.len32[b] >> 5, 16);
        if (k > 0) {
            const int slot = start[k] + atomicAdd(&hist[k], 1);
            s.order[slot] = b;
            *reinterpret_cast<int4 *>(s.desc + 4 * slot) = make_int4(s.lens[b], s.len32[b], s.off[b], b);
        }
    }
}

// ------------------------------------------------------------------ LayerNorm over 768 (one wave per row)
__device__ __forceinline__ void ln768_store(const float (&v)[12], const float *__restrict__ gamma, const float *__restrict__ beta,
                                            float eps, int lane, float *__restrict__ out_f32, bf16 *__restrict__ out_bf) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 4;
        const f4v g = *reinterpret_cast<const f4v *>(gamma + c);
        const f4v bb = *reinterpret_cast<const f4v *>(beta + c);
        f4v o;
        o.x = (v[i * 4 + 0] - mean) * rstd * g.x + bb.x;
        o.y = (v[i * 4 + 1] - mean) * rstd * g.y + bb.y;
        o.z = (v[i * 4 + 2] - mean) * rstd * g.z + bb.z;
        o.w = (v[i * 4 + 3] - mean) * rstd * g.w + bb.w;
        if (out_f32) *reinterpret_cast<f4v *>(out_f32 + c) = o;
        if (out_bf) {
            bf16x4 ob;
            ob.x = (bf16)o.x;
            ob.y = (bf16)o.y;
            ob.z = (bf16)o.z;
            ob.w = (bf16)o.w;
            *reinterpret_cast<bf16x4 *>(out_bf + c) = ob;
        }
    }
}

// K1: LN(word[id] + pos[pos_id] + type[0]) -> packed rows; dead rows (len <= t < len32) are zeroed
template <typename IT>
__global__ __launch_bounds__(256) void embed_ln_kernel(const IT *__restrict__ ids, int L, SeqInfo s, const float *__restrict__ word,
                                                       const float *__restrict__ posw, const float *__restrict__ typew,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                       int vocab, float *__restrict__ x_f32, bf16 *__restrict__ x_bf) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (t >= s.len32[b]) return;
    const size_t row = (size_t)s.off[b] + t;
    float v[12];
    if (t < s.lens[b]) {
        long id = (long)ids[(size_t)b * L + t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // out-of-range ids were flagged by seq_prep_kernel; never read outside the table
        const int p = s.pos[(size_t)b * L + t];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = i * 256 + lane * 4;
            const f4v a = *reinterpret_cast<const f4v *>(word + id * H + c);
            const f4v pp = *reinterpret_cast<const f4v *>(posw + (size_t)p * H + c);
            const f4v ty = *reinterpret_cast<const f4v *>(typew + c);
            // same association as the reference: (inputs_embeds + token_type) + position
            v[i * 4 + 0] = (a.x + ty.x) + pp.x;
            v[i * 4 + 1] = (a.y + ty.y) + pp.y;
            v[i * 4 + 2] = (a.z + ty.z) + pp.z;
            v[i * 4 + 3] = (a.w + ty.w) + pp.w;
        }
        ln768_store(v, gamma, beta, eps, lane, x_f32 ? x_f32 + row * H : nullptr, x_bf + row * H);   // (no fp32 copy on the large-batch path: its residual stream is bf16)
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = i * 256 + lane * 4;
            if (x_f32) *reinterpret_cast<f4v *>(x_f32 + row * H + c) = (f4v){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<bf16x4 *>(x_bf + row * H + c) = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        }
    }
}

// rows [total, Mp) of the packed matrices: written by nobody, read by the last M tile of every GEMM
__global__ __launch_bounds__(192) void zero_tail_rows_kernel(float *__restrict__ x_f32, bf16 *__restrict__ x_bf, const int *__restrict__ total_rows, long Mp) {
    const long row = (long)*total_rows + blockIdx.x;
    if (row >= Mp) return;
    const int c = threadIdx.x * 4;
    if (x_f32) *reinterpret_cast<f4v *>(x_f32 + row * H + c) = (f4v){0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<bf16x4 *>(x_bf + row * H + c) = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
}

// K4/K6 tail: x = LN(y) for every packed row (y already holds dense + bias + residual)
__global__ __launch_bounds__(256) void ln_rows_kernel(const float *__restrict__ y, const int *__restrict__ total_rows,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                      float *__restrict__ x_f32, bf16 *__restrict__ x_bf) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (size_t)*total_rows) return;
    float v[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f4v a = *reinterpret_cast<const f4v *>(y + row * H + i * 256 + lane * 4);
        v[i * 4 + 0] = a.x;
        v[i * 4 + 1] = a.y;
        v[i * 4 + 2] = a.z;
        v[i * 4 + 3] = a.w;
    }
    ln768_store(v, gamma, beta, eps, lane, x_f32 + row * H, x_bf + row * H);
}

// Deferred form of the same LayerNorm: writes only the bf16 copy (the next GEMM's A operand) and the row
// statistics.  The fp32 normalized row is never stored: its one consumer, the residual add of the next
// EPI_RESID epilogue, recomputes (y - mean) * rstd * gamma + beta from y -- 0.40 GB less HBM traffic per
// LayerNorm at 131 k tokens (1.0 -> 0.6 GB).
// n_part > 0: the RESID GEMM in front ran split-K (GemmArgs::ksplit): y holds slice 0's rows, part the bare sums of slices
// 1 .. n_part; they are added here, in slice order, and the complete row is written back to y (it is the next residual).
__global__ __launch_bounds__(256) void ln_stats_rows_kernel(float *__restrict__ y, const int *__restrict__ total_rows,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                            float2 *__restrict__ stats, bf16 *__restrict__ x_bf,
                                                            const float *__restrict__ part = nullptr, int n_part = 0, size_t part_stride = 0) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (size_t)*total_rows) return;
    float v[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f4v a = *reinterpret_cast<const f4v *>(y + row * H + i * 256 + lane * 4);
        for (int sp = 0; sp < n_part; ++sp) a += *reinterpret_cast<const f4v *>(part + sp * part_stride + row * H + i * 256 + lane * 4);
        if (n_part) *reinterpret_cast<f4v *>(y + row * H + i * 256 + lane * 4) = a;
        v[i * 4 + 0] = a.x;
        v[i * 4 + 1] = a.y;
        v[i * 4 + 2] = a.z;
        v[i * 4 + 3] = a.w;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + eps);
    if (lane == 0) stats[row] = make_float2(mean, rstd);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 4;
        const f4v g = *reinterpret_cast<const f4v *>(gamma + c);
        const f4v bb = *reinterpret_cast<const f4v *>(beta + c);
        bf16x4 ob;
        ob.x = (bf16)((v[i * 4 + 0] - mean) * rstd * g.x + bb.x);
        ob.y = (bf16)((v[i * 4 + 1] - mean) * rstd * g.y + bb.y);
        ob.z = (bf16)((v[i * 4 + 2] - mean) * rstd * g.z + bb.z);
        ob.w = (bf16)((v[i * 4 + 3] - mean) * rstd * g.w + bb.w);
        *reinterpret_cast<bf16x4 *>(x_bf + row * H + c) = ob;
    }
}

// ------------------------------------------------------------------ bf16 GEMM  C = A[M,K] . W[N,K]^T  (+ fused epilogue)
enum { EPI_QKV = 0, EPI_RESID = 1, EPI_GELU = 2 };

struct GemmArgs {
    const bf16 *A;        // [Mp][K] row-major, Mp multiple of 128
    const bf16 *W;        // [N][K] row-major (torch Linear weight)
    const float *bias;    // [N]
    int N, K;
    const int *total_rows;  // device: rows in use; tiles starting beyond it exit
    // epilogue outputs
    bf16 *q, *k, *v16;    // EPI_QKV: q,k [Mp][768]; v16 [Mp/16][768][16] (16-key groups, see attention_kernel)
    const float *resid;   // EPI_RESID: [Mp][768] fp32 residual -- or, with rstats, the PRE-LayerNorm rows it is recomputed from
    const float2 *rstats; // EPI_RESID: per-row (mean, rstd) of resid, or null
    const float *rgamma, *rbeta;   // EPI_RESID with rstats: that LayerNorm's affine
    float *y;             // EPI_RESID: [Mp][768] fp32
    bf16 *h;              // EPI_GELU: [Mp][N] bf16
    // EPI_RESID, small batches: split-K.  With ksplit = S > 1 an output tile is S work items, slice s accumulating k-tiles
    // [s KT/S, (s+1) KT/S): slice 0 writes acc + bias + residual to y as always, slice s >= 1 its bare partial sums to
    // part[(s-1) * part_stride ..]; the LayerNorm pass that follows every RESID GEMM (ln_stats_rows_kernel) adds them up in a
    // fixed order (deterministic) and writes y back.  16 row tiles x 6 column tiles of a 4 x 512 batch are 96 work items for 256
    // CUs, each walking all of K = 3072: split four ways the same GEMM is 384 items a quarter as long.
    int ksplit;
    float *part;
    size_t part_stride;
};

// erf GELU (hidden_act = "gelu"): 0.5 x (1 + erf(x / sqrt 2)).  erf(z) = z P(z^2) on |z| <= 3, P of degree 8
// (weighted minimax fit, constrained to erf(3) = 1 so that the clamped tails are exactly x and 0): |erf error|
// <= 3.5e-5, |GELU error| <= 7.4e-5 absolute -- 30x below the bf16 grid the result is stored on (2^-9
// relative).  13 packed VALU per PAIR of elements and no transcendental; the rcp + exp form it replaces
// (Abramowitz-Stegun 7.1.26, 1.5e-7) cost twice that, and the FFN-up epilogue (400 M elements per layer at
// 131 k tokens, no MFMA running beside it) was as long as half its main loop.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v gelu_erf2(f2v x) {
    f2v z = x * 0.70710678118654752f;
    z.x = __builtin_amdgcn_fmed3f(z.x, -3.0f, 3.0f);
    z.y = __builtin_amdgcn_fmed3f(z.y, -3.0f, 3.0f);
    const f2v u = z * z;
    f2v p = u * 4.4868795e-08f + (-2.1085477e-06f);
    p = p * u + 4.3755896e-05f;
    p = p * u + (-0.0005348297f);
    p = p * u + 0.004356828f;
    p = p * u + (-0.02546209f);
    p = p * u + 0.11166332f;
    p = p * u + (-0.37576896f);
    p = p * u + 1.1283866f;
    const f2v hx = x * 0.5f;
    return hx * (p * z) + hx;
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2((f2v){x, x}).x; }

// TMT = 32-row MFMA tiles per wave along M.  TMT=2: 128x128 tile, 4 waves (2x2), 2 WG/CU (small M).
// TMT=4: 256x256 tile, 8 waves (2x4), wave tile 128x64, 1 WG/CU: twice the flops per byte staged — at
// 128^2 the kernel is bound by L2->LDS traffic (64 flop/B needs 39 TB/s).
// PERSISTENT: one workgroup per CU slot walks a list of output tiles; the first k-tile of the NEXT
// output tile is staged during the last k-step of the current one, so neither the first-load latency
// nor the epilogue's stores are exposed (K = 768 means only 12 k-steps per tile).
#ifndef HAC_RESID_SCALAR_LN
#define HAC_RESID_SCALAR_LN 1
#endif
template <int EPI, int TMT>
__global__ __launch_bounds__(TMT * 128, 2) void gemm_bf16_nt_kernel(GemmArgs g) {
    // LDS: two stages of {A tile BMx64, W tile BNx64} bf16 + one 4 KiB transpose patch per wave.
    // 16-byte chunk c of row r lives at r*128 + ((c ^ ((r >> 1) & 7)) << 4): conflict-free ds_read_b128.  Tiles arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows per
    // wave-instruction, no VGPRs, no ds_write); the DMA destination is lane-linear, so the swizzle is
    // applied to the per-lane SOURCE address (chunk (lane&7) ^ (lane>>3) of row lane>>3).
    constexpr int BM = 64 * TMT, BN = BM;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int WN = BN / 64;           // waves along N (2 or 4); 2 along M
    constexpr bool S16 = (TMT == 4) && HAC_GEMM_S16;   // 16x16x32 MFMA form for the big tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = g.K, KT = K / BK;
    const int nx = g.N / BN;
    const int S = (EPI == EPI_RESID && g.ksplit > 1) ? g.ksplit : 1, KTs = KT / S;   // split-K slices (RESID, small batches), k-tiles per slice
    const int n_tiles = ((*g.total_rows + BM - 1) / BM) * nx * S;
    // XCD-aware tile lists: workgroups are dealt round-robin over the 8 XCDs; XCD x owns the contiguous
    // run [x*n/8, (x+1)*n/8) of tiles (consecutive tiles share the A rows -> L2 reuse of activations)
    // and its workgroups take them round-robin.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (gridDim.x + 7 - xcd) >> 3;
    const int run_lo = (int)((long)n_tiles * xcd / 8), run_hi = (int)((long)n_tiles * (xcd + 1) / 8);
    int tile = run_lo + slot;
    if (tile >= run_hi) return;
    const int wm = w / WN, wn = w % WN;
    const int r = lane & 31, hh = lane >> 5;

    typedef const __attribute__((address_space(1))) void *gvp;
    typedef __attribute__((address_space(3))) void *lvp;
    // swizzle f(row) = (row >> 1) & 7: two 128-B rows share a 256-B bank row, so the rows of one
    // ds_read_b128 lane group (e.g. 0-3, 12-15, 20-27) need distinct f among rows of equal parity;
    // (row & 7) repeats (rows 12 and 20) and costs a 2-way conflict on every fragment read.
    // Row of DMA piece i of wave w: w*32 + i*8 + (lane>>3)  ->  f = ((i&1)*4 + (lane>>4)) & 7.
    const int srow = lane >> 3;
    const int sch_even = (lane & 7) ^ (srow >> 1), sch_odd = sch_even ^ 4;
    const size_t lane_src_e = (size_t)(w * 32 + srow) * K + sch_even * 8;
    const size_t lane_src_o = (size_t)(w * 32 + srow) * K + sch_odd * 8;
    auto stage = [&](int buf, int t, int kt) {   // kt: absolute k-tile
        const int tt = EPI == EPI_RESID ? t / S : t;
        const int m0s = (tt / nx) * BM, n0s = (tt % nx) * BN;
        const bf16 *gA = g.A + (size_t)m0s * K + kt * BK;
        const bf16 *gW = g.W + (size_t)n0s * K + kt * BK;
        unsigned char *sb = smem + buf * STAGE + w * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t ls = ((i & 1) ? lane_src_o : lane_src_e) + (size_t)i * 8 * K;
            __builtin_amdgcn_global_load_lds((gvp)(gA + ls), (lvp)(sb + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gvp)(gW + ls), (lvp)(sb + BM * 128 + i * 1024), 16, 0, 0);
        }
    };
    // fragment byte offsets inside a stage, per k-step: row*128 + ((2*ks+hh) ^ f(row))*16
    int aoff[TMT], woff[2];
#pragma unroll
    for (int t = 0; t < TMT; ++t) aoff[t] = (wm * (32 * TMT) + t * 32 + r) * 128;
#pragma unroll
    for (int t = 0; t < 2; ++t) woff[t] = BM * 128 + (wn * 64 + t * 32 + r) * 128;
    const int sw = (r >> 1) & 7;  // f(row) is the same for every fragment row of a lane (rows differ by multiples of 32)
    float *patch = reinterpret_cast<float *>(smem + 2 * STAGE) + w * 1024;  // [16 rows][64 cols] fp32, wave-private

    int cur = 0;
    stage(0, tile, EPI == EPI_RESID ? (tile % S) * KTs : 0);
    __syncthreads();
    for (; tile < run_hi; tile += per_xcd) {
        const int otile = EPI == EPI_RESID ? tile / S : tile, slice = EPI == EPI_RESID ? tile - otile * S : 0, kb = slice * KTs;
        const int m0 = (otile / nx) * BM, n0 = (otile % nx) * BN;
        const int next_tile = tile + per_xcd;
        // Two accumulator forms.  S16 (the 256^2 tile): v_mfma_f32_16x16x32_bf16, 8 x 4 tiles of 16x16 per wave;
        // same FLOPs, LDS traffic and matrix-pipe cycles as the 32x32x16 form, but the part holds a higher
        // clock on it (bare LDS-read + MFMA loop on random data: 1.79 vs 1.65 PF).
        f32x16 acc[S16 ? 1 : TMT][2];
        f32x4 acc16[S16 ? 2 * TMT : 1][4];
        if constexpr (S16) {
#pragma unroll
            for (int a = 0; a < 2 * TMT; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc16[a][b][e] = 0.f;
        } else {
#pragma unroll
            for (int a = 0; a < TMT; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        }

        for (int kt = 0; kt < KTs; ++kt) {
            if (kt + 1 < KTs) stage(cur ^ 1, tile, kb + kt + 1);
            else if (next_tile < run_hi) stage(cur ^ 1, next_tile, EPI == EPI_RESID ? (next_tile % S) * KTs : 0);
            const unsigned char *sc = smem + cur * STAGE;
            if constexpr (S16) {
                // lane (m = lane&15, kg = lane>>4): A/B fragment = row (tile*16 + m), chunk (4*ks32 + kg) ^ f(row);
                // f(row) = (m >> 1) & 7 for every tile (tile rows differ by multiples of 16)
                const int m16 = lane & 15, kg = lane >> 4, sw16 = (m16 >> 1) & 7;
                const unsigned char *ab = sc + (wm * (32 * TMT) + m16) * 128;
                const unsigned char *wb = sc + BM * 128 + (wn * 64 + m16) * 128;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int c = ((ks * 4 + kg) ^ sw16) << 4;
                    bf16x8 wf16[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) wf16[j] = *reinterpret_cast<const bf16x8 *>(wb + j * 2048 + c);
                    bf16x8 a_cur = *reinterpret_cast<const bf16x8 *>(ab + c), a_nxt = a_cur;
#pragma unroll
                    for (int i = 0; i < 2 * TMT; ++i) {
                        if (i + 1 < 2 * TMT) a_nxt = *reinterpret_cast<const bf16x8 *>(ab + (i + 1) * 2048 + c);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, wf16[j], acc16[i][j], 0, 0, 0);
                        a_cur = a_nxt;
                    }
                }
            } else {
            bf16x8 af[2][TMT], wf[2][2];
#pragma unroll
            for (int t = 0; t < TMT; ++t) af[0][t] = *reinterpret_cast<const bf16x8 *>(sc + aoff[t] + ((hh ^ sw) << 4));
#pragma unroll
            for (int t = 0; t < 2; ++t) wf[0][t] = *reinterpret_cast<const bf16x8 *>(sc + woff[t] + ((hh ^ sw) << 4));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) {
                    const int c = ((ks + 1) * 2 + hh) ^ sw;
#pragma unroll
                    for (int t = 0; t < TMT; ++t) af[(ks + 1) & 1][t] = *reinterpret_cast<const bf16x8 *>(sc + aoff[t] + (c << 4));
#pragma unroll
                    for (int t = 0; t < 2; ++t) wf[(ks + 1) & 1][t] = *reinterpret_cast<const bf16x8 *>(sc + woff[t] + (c << 4));
                }
#pragma unroll
                for (int a = 0; a < TMT; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][a], wf[ks & 1][b], acc[a][b], 0, 0, 0);
            }
            }
            __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) here: next stage landed, this one is free
            cur ^= 1;
        }

        // ---- epilogue.  acc[a][b][e] is element (m, n) with n = n0 + wn*64 + b*32 + r and
        // m = m0 + wm*32*TMT + a*32 + (e&3) + 8*(e>>2) + 4*hh: a lane owns ONE column and 16 scattered
        // rows, so storing from registers means 2- or 4-byte accesses.  Each wave instead transposes one
        // 16x64 sub-tile at a time through its private LDS patch and then touches global memory
        // row-wise with 16-byte accesses.  (The next tile's first k-step is already in LDS.)
        const int ncol0 = n0 + wn * 64;
        const bool partial = EPI == EPI_RESID && slice > 0;   // (workgroup-uniform) a split-K slice beyond the first: bare sums
        const float bias0 = partial ? 0.f : g.bias[ncol0 + r], bias1 = partial ? 0.f : g.bias[ncol0 + 32 + r];
        [[maybe_unused]] float *const y_out = partial ? g.part + (size_t)(slice - 1) * g.part_stride : g.y;
        if (EPI == EPI_QKV && n0 >= 2 * H) {
            // V goes out in 16-key groups ([Mp/16][768][16]) for the attention kernel's LDS-DMA: a lane holds
            // 4 consecutive tokens of one feature, i.e. 8 contiguous bytes of that layout.
            if constexpr (S16) {
                const int m16 = lane & 15, kg = lane >> 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = ncol0 + j * 16 + m16 - 2 * H;
                    const float bias = g.bias[ncol0 + j * 16 + m16];
                    const size_t mg = (size_t)(m0 + wm * (32 * TMT)) >> 4;
#pragma unroll
                    for (int i = 0; i < 2 * TMT; ++i) {   // tokens 16*i + 4*kg + (0..3) of the wave's rows: group i, slot 4*kg
                        bf16x4 o;
                        o.x = (bf16)(acc16[i][j][0] + bias);
                        o.y = (bf16)(acc16[i][j][1] + bias);
                        o.z = (bf16)(acc16[i][j][2] + bias);
                        o.w = (bf16)(acc16[i][j][3] + bias);
                        *reinterpret_cast<bf16x4 *>(g.v16 + ((mg + i) * H + n) * 16 + 4 * kg) = o;
                    }
                }
                continue;
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int n = ncol0 + b * 32 + r - 2 * H;
                const float bias = b ? bias1 : bias0;
#pragma unroll
                for (int a = 0; a < TMT; ++a) {
                    const size_t mg = (size_t)(m0 + wm * (32 * TMT) + a * 32) >> 4;   // first 16-token group of the MFMA tile
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        bf16x4 o;
                        o.x = (bf16)(acc[a][b][e4 * 4 + 0] + bias);
                        o.y = (bf16)(acc[a][b][e4 * 4 + 1] + bias);
                        o.z = (bf16)(acc[a][b][e4 * 4 + 2] + bias);
                        o.w = (bf16)(acc[a][b][e4 * 4 + 3] + bias);
                        // tokens 8*e4 + 4*hh + (0..3) of the tile: group e4>>1, slot 8*(e4&1) + 4*hh
                        *reinterpret_cast<bf16x4 *>(g.v16 + ((mg + (e4 >> 1)) * H + n) * 16 + 8 * (e4 & 1) + 4 * hh) = o;
                    }
                }
            }
            continue;
        }
        float bias16[4];
        if constexpr (S16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bias16[j] = partial ? 0.f : g.bias[ncol0 + j * 16 + (lane & 15)];
        }
#pragma unroll
        for (int a = 0; a < TMT; ++a) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {   // rows 16*half .. 16*half+15 of the 32-row MFMA tile
                // residual rows of this sub-tile are requested first: their L2 latency hides under the transpose
                f4v rs[4];
                if constexpr (EPI == EPI_RESID) {
                    const size_t mr = (size_t)m0 + wm * (32 * TMT) + a * 32 + half * 16;
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        rs[it] = partial ? (f4v){0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f4v *>(g.resid + (mr + it * 4 + (lane >> 4)) * H + ncol0 + (lane & 15) * 4);
                    if (g.rstats && !partial) {   // deferred LayerNorm of the residual rows (ln_stats_rows_kernel)
                        const f4v gam = *reinterpret_cast<const f4v *>(g.rgamma + ncol0 + (lane & 15) * 4);
                        const f4v bet = *reinterpret_cast<const f4v *>(g.rbeta + ncol0 + (lane & 15) * 4);
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const float2 st = g.rstats[mr + it * 4 + (lane >> 4)];
#if HAC_RESID_SCALAR_LN
                            // Component by component, each value pinned to a VGPR of its own, so that hipcc cannot form packed-fp32
                            // instructions here.  With the vector expression it emitted v_sub_f32 x 4, v_pk_mul_f32 ... op_sel:[0,1] (both
                            // halves times rstd, the high word of the statistics pair) and v_pk_fma_f32 right behind the loads' waits, and on
                            // MI355X the LOW half of such a packed result was now and then wrong for one 16-lane pass: one element in a few
                            // thousand forwards of 6-16 k rows (found by the encoder soak in round 4; a forward repeated on the same input
                            // differed in one sequence in ~5 % of the runs; LABNOTES, round 4, 2.4).  Not reproduced with scalar instructions.
                            float c0 = rs[it].x, c1 = rs[it].y, c2 = rs[it].z, c3 = rs[it].w, mean = st.x, rstd = st.y;
                            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(mean), "+v"(rstd));
                            c0 = fmaf((c0 - mean) * rstd, gam.x, bet.x);
                            c1 = fmaf((c1 - mean) * rstd, gam.y, bet.y);
                            c2 = fmaf((c2 - mean) * rstd, gam.z, bet.z);
                            c3 = fmaf((c3 - mean) * rstd, gam.w, bet.w);
                            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
                            rs[it] = (f4v){c0, c1, c2, c3};
#else
                            rs[it] = (rs[it] - st.x) * st.y * gam + bet;
#endif
                        }
                    }
                }
                if constexpr (S16) {   // 16x16 tiles: lane (column m16 of tile j, rows 4*kg .. 4*kg+3)
                    const int m16 = lane & 15, kg = lane >> 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) patch[(4 * kg + e) * 64 + 16 * j + m16] = acc16[2 * a + half][j][e] + bias16[j];
                } else {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = half * 8 + e8;
                    const int row = (e8 & 3) + 8 * (e8 >> 2) + 4 * hh;   // 0..15
                    patch[row * 64 + r] = acc[a][0][e] + bias0;
                    patch[row * 64 + 32 + r] = acc[a][1][e] + bias1;
                }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
                __builtin_amdgcn_wave_barrier();
                const size_t mrow = (size_t)m0 + wm * (32 * TMT) + a * 32 + half * 16;
                if constexpr (EPI == EPI_RESID) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 4 + (lane >> 4), c4 = (lane & 15) * 4;
                        const f4v v = *reinterpret_cast<const f4v *>(patch + row * 64 + c4);
                        const size_t off = (mrow + row) * H + ncol0 + c4;
                        *reinterpret_cast<f4v *>(y_out + off) = v + rs[it];
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
                        const f4v v0 = *reinterpret_cast<const f4v *>(patch + row * 64 + c8);
                        const f4v v1 = *reinterpret_cast<const f4v *>(patch + row * 64 + c8 + 4);
                        bf16x8 o;
                        if constexpr (EPI == EPI_GELU) {
                            const f2v g0 = gelu_erf2((f2v){v0.x, v0.y}), g1 = gelu_erf2((f2v){v0.z, v0.w});
                            const f2v g2 = gelu_erf2((f2v){v1.x, v1.y}), g3 = gelu_erf2((f2v){v1.z, v1.w});
                            o[0] = (bf16)g0.x; o[1] = (bf16)g0.y; o[2] = (bf16)g1.x; o[3] = (bf16)g1.y;
                            o[4] = (bf16)g2.x; o[5] = (bf16)g2.y; o[6] = (bf16)g3.x; o[7] = (bf16)g3.y;
                            *reinterpret_cast<bf16x8 *>(g.h + (mrow + row) * (size_t)g.N + ncol0 + c8) = o;
                        } else {  // Q (scaled by log2(e)/sqrt(64) in fp32, before the one rounding to bf16: the softmax runs in base 2) or K
                            const float sc = n0 < H ? 0.125f * 1.44269504088896341f : 1.0f;
                            o[0] = (bf16)(v0.x * sc); o[1] = (bf16)(v0.y * sc); o[2] = (bf16)(v0.z * sc); o[3] = (bf16)(v0.w * sc);
                            o[4] = (bf16)(v1.x * sc); o[5] = (bf16)(v1.y * sc); o[6] = (bf16)(v1.z * sc); o[7] = (bf16)(v1.w * sc);
                            bf16 *dst = n0 < H ? g.q : g.k;
                            const int nn = (n0 < H ? ncol0 : ncol0 - H) + c8;
                            *reinterpret_cast<bf16x8 *>(dst + (mrow + row) * H + nn) = o;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_s_waitcnt(0xC07F);  // reads done before the next sub-tile overwrites the patch
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

#include "gemm8.inc"

// ------------------------------------------------------------------ attention
struct AttnArgs {
    const bf16 *q, *k;   // q (pre-scaled by log2(e)/8), k: [Mp][768]
    const bf16 *v16;     // V in 16-key groups: [Mp/16][768][16]  (element (token m, feature n) at ((m>>4)*768 + n)*16 + (m&15))
    bf16 *ctx;           // [Mp][768]
    SeqInfo s;
    int cls_only;        // last layer: only the query block holding <s> is needed (models.py:56 takes [:,0])
    int qsplit;          // streaming kernel, small batches: an item's query rows are dealt to this many workgroups (1, 2, 4, 8 or 16), see there
    int one_class;       // streaming kernel, small batches: the 16-wave instantiation takes the short sequences too (one launch per layer)
    // the two-blocks-per-wave kernel (attn_pipe.inc) computes every item with the reference at 0 and flags those that need it moved:
    int *redo_flags;     // [items of both classes] 1: compute again (set by attention_pipe_kernel, taken and cleared by the fix-up pass)
    int *redo_count;     // [1] how many were flagged in this layer (zeroed per forward)
    int fixup;           // attention_stream_kernel: take only flagged items
};
// How far (log2 domain) a score may lie from a row's reference before the reference is moved: the streaming kernels' shared rule.
constexpr float ATT_TAU = 64.0f;

// The two-pass attention kernel of round 1, kept behind hac_encoder_set_option("attn", "twopass") as the tests' cross-check of
// the streaming kernel below (exact row maxima first, one workgroup per item).
// Workgroup = WAVES waves = ALL query rows of one (sequence, head); a wave owns U 32-row query blocks (instantiated with
// U = 1: four waves per SIMD hide more latency than sharing each K and V fragment between two MFMAs saves, -7 %).
// Both the head's K (len32 x 64) and V (64 x len32) live in
// LDS for the whole workgroup (<= 128 KiB), brought in once by LDS-DMA:
//   K image   row*128 + ((chunk ^ ((row>>1)&7)) << 4)           (the GEMM's XOR swizzle)
//   V image   1-KiB pieces (16-key group G, d-tile t): lane l = 32*hh + r holds V[keys 16G+8hh..+7][d = 32t+r],
//             i.e. a piece IS the A operand of one MFMA and one contiguous KiB of the v16 tensor.
// Sequences of <= 256 rows take the 8-wave instantiation (<= 64 KiB: two workgroups per CU), longer
// ones the 16-wave instantiation; both are launched over all sequences and a workgroup whose sequence
// belongs to the other class leaves at once.
// History: with V^T fragments fetched per wave from L2 (8-byte loads at a row stride) the kernel ran at
// the texture-address rate, 0.55 ms per layer at B=256, L=512, and 0.40 ms with those loads removed.
//
// Index algebra (v_mfma_f32_32x32x16_bf16; lane = 32*hh + r):
//   S^T = K.Q^T   A row m <- key 32kb + pi(m), pi = swap bits 2 and 3;  B column n <- query r.
//                 accumulator e of lane (r, hh)  <->  row m = (e&3) + 8(e>>2) + 4hh.
//   O^T = V^T.P^T the B operand of k-step s2 is accumulator registers 8*s2 .. 8*s2+7: slot j is row
//                 m = 16*s2 + 8(j>>2) + 4hh + (j&3), i.e. key pi(m) = 16*s2 + 8hh + j  -- eight consecutive
//                 keys, which is what a V piece holds.  Output: lane (query r, hh), accumulator e <-> d =
//                 32t + (e&3) + 8(e>>2) + 4hh: four consecutive features per lane, 1/l is lane-local.
template <int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64, WAVES * U == 8 ? 2 : 1) void attention_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y, head = blockIdx.x;
    const int len32 = a.s.len32[b];
    if ((len32 > 256) != (WAVES * U == 16)) return;  // the other instantiation's sequence (whole workgroup leaves)
    const int len = a.s.lens[b];
    const size_t base = (size_t)a.s.off[b];
    const int r = lane & 31, hh = lane >> 5;
    const int nkb = len32 >> 5;
    const int q0 = w * 32 * U;
    const bool active = q0 < len32 && !(a.cls_only && w != 0);
    unsigned char *vimg = smem + len32 * 128;

    typedef const __attribute__((address_space(1))) void *gvp;
    typedef __attribute__((address_space(3))) void *lvp;
    // Q^T fragments (B operand of S^T): lane (q = r, half hh) holds q[16*ks + 8*hh .. +8)
    bf16x8 qf[U][4];
    if (active) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bf16 *qrow = a.q + (base + q0 + u * 32 + r) * H + head * DH + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[u][ks] = *reinterpret_cast<const bf16x8 *>(qrow + ks * 16);
        }
    }
    {   // K rows -> LDS, 8 rows (1 KiB) per wave-instruction.  Rows of this wave's pieces are r8*8 + (lane>>3)
        // with r8 = w (mod 2), so the swizzle term is ((w&1)*4 + (lane>>4)) & 7.
        const int srow = lane >> 3, schunk = (lane & 7) ^ ((((w & 1) << 2) + (srow >> 1)) & 7);
        const bf16 *src = a.k + (base + srow) * H + head * DH + schunk * 8;
        for (int r8 = w; r8 * 8 < len32; r8 += WAVES)
            __builtin_amdgcn_global_load_lds((gvp)(src + (size_t)r8 * 8 * H), (lvp)(smem + r8 * 1024), 16, 0, 0);
    }
    __syncthreads();  // hipcc drains the LDS-DMA (and the Q loads) here: K is resident
    {   // V pieces -> LDS while pass 1 runs on K
        const bf16 *src = a.v16 + ((base >> 4) * H + head * DH) * 16 + r * 16 + hh * 8;
        for (int p = w; p * 8 < len32; p += WAVES)
            __builtin_amdgcn_global_load_lds((gvp)(src + ((size_t)(p >> 1) * H + (p & 1) * 32) * 16), (lvp)(vimg + p * 1024), 16, 0, 0);
    }
    const int pr = (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1);  // pi(r)
    const unsigned char *krow = smem + pr * 128;
    const int sw = (pr >> 1) & 7;
    // key held by accumulator register e of this lane, relative to its key block
    auto key_of = [&](int e) { return (e & 3) + 4 * ((e >> 2) & 1) + 8 * hh + 16 * (e >> 3); };

    // pass 1: row maxima
    float mxs[U] = {};
    if (active) {
        float mx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) mx[u] = -INFINITY;
        for (int kb = 0; kb < nkb; ++kb) {
            f32x16 s[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[u][e] = 0.f;
            const unsigned char *kp = krow + kb * 4096;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(kp + (((2 * ks + hh) ^ sw) << 4));
#pragma unroll
                for (int u = 0; u < U; ++u) s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[u][ks], s[u], 0, 0, 0);
            }
            if (kb * 32 + 32 <= len) {   // whole block valid (wave-uniform): no per-key masking
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int u = 0; u < U; ++u) mx[u] = fmaxf(mx[u], s[u][e]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kb * 32 + key_of(e) < len) {
#pragma unroll
                        for (int u = 0; u < U; ++u) mx[u] = fmaxf(mx[u], s[u][e]);
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) mxs[u] = fmaxf(mx[u], __shfl_xor(mx[u], 32));
    }
    __syncthreads();  // V is resident (drains this wave's DMA, then meets the others)
    if (!active) return;

    // pass 2: P = exp(S - max), l = sum P, O^T = V^T.P^T
    float lsum[U] = {};
    f32x16 o[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[u][t][e] = 0.f;
    const unsigned char *vlane = vimg + lane * 16;
    // The scores are already in the log2 domain (Q carries log2(e)/8) and the S^T accumulator starts at
    // -max, so P = 2^acc with no VALU between the MFMA and the v_exp.  K fragments are read one key
    // block ahead (two waves per SIMD do not hide an LDS round trip in front of every MFMA pair).
    bf16x8 kc[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kc[ks] = *reinterpret_cast<const bf16x8 *>(krow + (((2 * ks + hh) ^ sw) << 4));
    auto step = [&](int kb, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        f32x16 s[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[u][e] = -mxs[u];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[ks], qf[u][ks], s[u], 0, 0, 0);
        const unsigned char *vp = vlane + kb * 4096;
        bf16x8 vf[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int t = 0; t < 2; ++t) vf[s2][t] = *reinterpret_cast<const bf16x8 *>(vp + (s2 * 2 + t) * 1024);
        const unsigned char *kp = krow + min(kb + 1, nkb - 1) * 4096;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kc[ks] = *reinterpret_cast<const bf16x8 *>(kp + (((2 * ks + hh) ^ sw) << 4));
        // raw v_exp_f32: arguments are <= 0 (up to rounding), results in (0,1]; libm's exp2f wraps every
        // call in range checks and ldexp, 5 VALU instead of 1
        bf16x8 pf[U][2];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool valid = !MASKED || kb * 32 + key_of(e) < len;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float p = valid ? __builtin_amdgcn_exp2f(s[u][e]) : 0.f;
                lsum[u] += p;
                pf[u][e >> 3][e & 7] = (bf16)p;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < U; ++u) o[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][t], pf[u][s2], o[u][t], 0, 0, 0);
    };
    const int nfull = len >> 5;  // key blocks without padding keys (nkb - nfull is 0 or 1)
    for (int kb = 0; kb < nfull; ++kb) step(kb, std::false_type{});
    if (nfull < nkb) step(nfull, std::true_type{});

#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (q0 + u * 32 >= len32 || (a.cls_only && u)) break;  // past the sequence (computed on foreign rows) or not needed
        const float inv = 1.0f / (lsum[u] + __shfl_xor(lsum[u], 32));
        bf16 *crow = a.ctx + (base + q0 + u * 32 + r) * H + head * DH + 4 * hh;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 ov;
                ov.x = (bf16)(o[u][t][g4 * 4 + 0] * inv);
                ov.y = (bf16)(o[u][t][g4 * 4 + 1] * inv);
                ov.z = (bf16)(o[u][t][g4 * 4 + 2] * inv);
                ov.w = (bf16)(o[u][t][g4 * 4 + 3] * inv);
                *reinterpret_cast<bf16x4 *>(crow + 32 * t + 8 * g4) = ov;
            }
    }
}

// ------------------------------------------------------------------ attention, streaming form
// The two-pass kernel above keeps a head's whole K and V in LDS: one workgroup per CU for 257+ rows, and (in-kernel stamps,
// L = 512) 10 k of a workgroup's 49 k cycles go to waiting for K with nothing else on the CU -- every CU loading at once, at the
// chip's HBM rate; an item moves 256 KiB (Q, K, V in, context out), 40 % of the kernel's time at that rate.  Here the
// workgroups are persistent and K/V arrive as a continuous stream of chunks through a 3-stage LDS ring that runs on across
// the (sequence, head) items, two chunks ahead of the arithmetic, one K piece and one V piece per wave and chunk; the next
// item's Q rows are staged through LDS the same way.  K passes through once: the softmax is the online form (running
// reference m per query row, P = 2^(s - m) with the accumulator started at -m; m is only raised, exactly, when a block's
// maximum exceeds it by more than TAU = ATT_TAU = 64 (in the log2 domain) -- then l and O are rescaled -- so P <= 2^64 and the common step has
// no rescale).
//   WAVES = 16: sequences of 257..512 rows, chunks of 128 keys, 160 KiB of LDS, one workgroup per CU;
//   WAVES = 8 : up to 256 rows, chunks of 64 keys, 80 KiB, two workgroups per CU.
// Index algebra, LDS images of a K row block and of a V piece: as in attention_kernel.  vmcnt is in order; every wave
// issues the same DMA instructions per chunk (pieces past the sequence are redirected to a valid piece of the right
// parity, the exhausted stream re-reads its last chunk), so the waits can be counted: at a chunk's barrier the two youngest
// DMAs (the next chunk's) may stay in flight; at an item's first chunk everything but this wave's own context stores is
// waited for (the item's Q pieces were issued at least one chunk earlier).
#ifdef ATT_STAMP   // development: s_memtime stamps of one workgroup's 4th item (tools/probes/attn_stream_probe.hip reads them)
__device__ unsigned long long g_att_stamps[16 * 32];
#define ATTS_T(i) do { if (blockIdx.x == 5 && n_done == 3 && lane == 0) g_att_stamps[w * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATTS_T(i) do { } while (0)
#endif
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 4) void attention_stream_kernel(AttnArgs a) {
    constexpr int CHUNK = WAVES * 8;             // keys per ring stage
    constexpr int STEPS = CHUNK / 32;            // 32-key steps per chunk
    constexpr int NS = 3;                        // ring stages: the stream runs NS - 1 chunks ahead
    constexpr int STAGE = CHUNK * 256;           // bytes: K rows (CHUNK x 128) | V pieces (CHUNK x 128)
    constexpr int QBYTES = WAVES * 32 * 128;     // the item's Q rows
    // How far a score may exceed the reference before the reference is moved.  P = 2^(s - m) <= 2^TAU is held in bf16 (8-bit
    // exponent) and summed in fp32: 512 keys x 2^64 x |v| stays far inside the range, the relative precision of P does not depend
    // on its scale, and a row's first block always contains P = 1 -- so a generous margin costs nothing and makes the rescale a
    // rare event on any realistic logits (with 8 it ran on most steps once the logits' standard deviation reached ~5).
    constexpr float TAU = ATT_TAU;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const ring = smem + QBYTES;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int G = gridDim.x;
    typedef const __attribute__((address_space(1))) void *gvp;
    typedef __attribute__((address_space(3))) void *lvp;
    // the sequence table is read with scalar loads, by hand: inside these loops hipcc would use vector loads and wait
    // vmcnt(0) for each, draining the DMA stream
    auto sload = [&](const int *p, int i) {
        int v;
        asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p), "s"(i * 4) : "memory");
        return v;
    };
    // items of this instantiation's length class: (position in the class's part of the work list, head)
    const int n_long = sload(a.s.ncls, 0) + (WAVES == 16 && a.one_class ? sload(a.s.ncls, 1) : 0);   // (the order list holds the long class first)
    const int list0 = WAVES == 16 ? 0 : n_long;
    // Small batches (the reference's 4 queries per call are 48 items on 256 CUs): an item's query rows are dealt to QS workgroups,
    // WAVES / QS waves of each take 32 rows apiece and the others only help moving K and V.  Fewer waves per SIMD run their steps
    // faster (the step is bound by the SIMD's VALU issue), more CUs work: 4 x 512 tokens 19 -> ~8 us per layer.  Same arithmetic per row.
    const int QS = a.cls_only ? 1 : min(a.qsplit, WAVES);
    const int n_items = (WAVES == 16 ? n_long : sload(a.s.ncls, 1)) * NH * QS;
    // Fix-up mode (behind attention_pipe_kernel, which computes every item with the reference at 0 and flags the ones whose
    // reference has to move): only flagged items are taken; each is visited by exactly one workgroup, which clears its flag.
    if (a.fixup && sload(a.redo_count, 0) == 0) return;
    auto next_item = [&](int t) {
        t += G;
        if (a.fixup)
            while (t < n_items && sload(a.redo_flags, list0 * NH + t) == 0) t += G;
        return t;
    };
    struct Item { int len, len32, head, nch, q0; size_t base; };
    auto describe = [&](int t) {
        const int part = t % QS, th = t / QS;
        const int pos = th / NH;
        const int b = sload(a.s.order, list0 + pos);
        Item it;
        it.head = th - pos * NH;
        it.q0 = part * (WAVES * 32 / QS);
        it.len = sload(a.s.lens, b);
        it.len32 = sload(a.s.len32, b);
        it.base = (size_t)sload(a.s.off, b);
        it.nch = (it.len32 + CHUNK - 1) / CHUNK;
        return it;
    };
    // DMA source = workgroup-uniform base (SGPRs) + a 32-bit lane offset, recomputed at each use (kept live across the
    // arithmetic it would be spilled, and a scratch reload is a vmcnt(0)).  A K or Q piece is 8 rows, 8p + (lane>>3): the
    // GEMM swizzle term of its rows is ((p&1)*4 + (lane>>4)) & 7, and every piece of this wave has the parity of w.
    auto k_lane = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        return (unsigned)((l >> 3) * H + ((l & 7) ^ ((((w & 1) << 2) + (l >> 4)) & 7)) * 8);
    };
    auto v_lane = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        return (unsigned)((l & 31) * 16 + (l >> 5) * 8);
    };
    auto issue_q = [&](const Item &it) {   // 4 pieces per wave into the Q region (slot = piece), rows from the item's first query row on
        const int p_last = max((int)(w & 1), ((it.len32 - it.q0) >> 3) - 2 + (w & 1));   // (a part past the sequence's end loads rows it never uses)
        const bf16 *src = a.q + (it.base + min(it.q0, it.len32 - 32)) * H + it.head * DH;
        const unsigned kl = k_lane();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = w + WAVES * j;
            __builtin_amdgcn_global_load_lds((gvp)(src + (size_t)min(p, p_last) * 8 * H + kl), (lvp)(smem + p * 1024), 16, 0, 0);
        }
    };
    // ---- the K/V stream (runs NS - 1 chunks ahead of the arithmetic, across items)
    int s_t, s_c = 0, s_stage = 0;
    Item s_it;
    bool s_more = true;
    auto stream_issue = [&]() {
        const int p = min(s_c * WAVES + w, (s_it.len32 >> 3) - 2 + (w & 1));          // K piece (8 rows)
        const int gi = min(s_c * (CHUNK / 16) + (w >> 1), (s_it.len32 >> 4) - 1);      // V piece: 16-key group gi, d-tile w & 1
        unsigned char *dst = ring + s_stage * STAGE + w * 1024;
        __builtin_amdgcn_global_load_lds((gvp)(a.k + s_it.base * H + s_it.head * DH + (size_t)p * 8 * H + k_lane()), (lvp)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gvp)(a.v16 + (((s_it.base >> 4) + gi) * H + s_it.head * DH + (w & 1) * 32) * 16 + v_lane()),
                                         (lvp)(dst + CHUNK * 128), 16, 0, 0);
        s_stage = s_stage + 1 == NS ? 0 : s_stage + 1;
        if (s_more && ++s_c == s_it.nch) {
            const int t2 = next_item(s_t);
            if (t2 < n_items) {
                s_t = t2;
                s_it = describe(t2);
                s_c = 0;
            } else {
                s_more = false;   // exhausted: the remaining issues re-read this chunk into the free stage (the counts stay exact)
                s_c = s_it.nch - 1;
            }
        }
    };
    const int pr = (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1);  // pi(r)
    const int sw = (pr >> 1) & 7;
    auto key_of = [&](int e) { return (e & 3) + 4 * ((e >> 2) & 1) + 8 * hh + 16 * (e >> 3); };

    int t = next_item((int)blockIdx.x - G);
    if (t >= n_items) return;
    Item cur = describe(t);
    s_t = t;
    s_it = cur;
    issue_q(cur);
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) stream_issue();
    bool stored = false;   // this wave's 4 context stores are the youngest entries of its queue
    int c_stage = 0;
    bf16x8 qf[4];
    for ([[maybe_unused]] int n_done = 0;; ++n_done) {   // (read by the stamp macro only)
        ATTS_T(0);
        const int tn = next_item(t);
        const bool has_next = tn < n_items;
        const Item nxt = describe(has_next ? tn : t);
        const int len = cur.len, nkb = cur.len32 >> 5;
        const int nfull = len >> 5;            // key blocks without padding keys (nkb - nfull is 0 or 1)
        const bool active = w < WAVES / QS && cur.q0 + w * 32 < cur.len32 && !(a.cls_only && w != 0);
        float m_ref = 0.f, lsum = 0.f;
        f32x16 o[2], negm;
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[e] = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[tt][e] = 0.f;
        auto step = [&](const unsigned char *stage, int kk, int kb, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            const unsigned char *kp = stage + kk * 4096 + pr * 128;
            bf16x8 kf[4];   // all four reads in flight before the first MFMA (left alone hipcc funnels them through one register quad)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8 *>(kp + (((2 * ks + hh) ^ sw) << 4));
            // the accumulator starts at -m: negm is a register tuple kept across the steps (C operand of the first MFMA), rewritten
            // only when the reference moves -- not 16 v_mov per step
            f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], negm, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 DS reads
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMAs
            // this lane's 16 keys of the block, relative to the reference
            float mloc = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (!MASKED || kb * 32 + key_of(e) < len) mloc = fmaxf(mloc, s[e]);
            // (round 6) the reference starts at 0 and moves only when it has to: a block's maximum more than TAU above it or -- first
            // block -- more than TAU below.  Rows whose scores stay within +-ATT_TAU keep m = 0, which the two-blocks-per-wave
            // kernel (attn_pipe.inc) exploits; the rule is shared so that the two kernels stay bit-identical.
            const bool first = kb == 0;
            if (__builtin_amdgcn_ballot_w64(mloc > TAU || (first && mloc < -TAU)) != 0) {   // (wave-uniform) move the reference, exactly
                const float mrow = fmaxf(mloc, __shfl_xor(mloc, 32));       // the query row's maximum over the block
                const float delta = (mrow > TAU || (first && mrow < -TAU)) ? mrow : 0.f;
                const float sc = first ? 1.f : __builtin_amdgcn_exp2f(-delta);   // (first block: l and O are still zero)
                m_ref += delta;
#pragma unroll
                for (int e = 0; e < 16; ++e) negm[e] = -m_ref;
                lsum *= sc;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[tt][e] *= sc;
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] -= delta;
            }
            bf16x8 pf[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool valid = !MASKED || kb * 32 + key_of(e) < len;
                const float p = valid ? __builtin_amdgcn_exp2f(s[e]) : 0.f;
                lsum += p;
                pf[e >> 3][e & 7] = (bf16)p;
            }
            // V fragments only now: four waves per SIMD hide the LDS round trip, and the registers of s are free again
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char *vp = stage + CHUNK * 128 + kk * 4096 + lane * 16;
            bf16x8 vf[2][2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) vf[s2][tt] = *reinterpret_cast<const bf16x8 *>(vp + (s2 * 2 + tt) * 1024);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][tt], pf[s2], o[tt], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        };
#pragma nounroll
        for (int c = 0; c < cur.nch; ++c) {
            ATTS_T(1 + 4 * c);
            if (c == 0) {
                if (stored) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (NS - 2)) : "memory");
            }
            __builtin_amdgcn_s_barrier();      // this chunk has landed (every wave's pieces); the stage two back is free
            __builtin_amdgcn_sched_barrier(0);
            ATTS_T(2 + 4 * c);
            stream_issue();
            if (c == 0) {
                if (active) {                  // Q^T fragments (B operand of S^T): lane (q = r, half hh) holds q[16*ks + 8*hh .. +8)
                    const unsigned char *qrow = smem + (w * 32 + r) * 128;
                    const int qsw = (r >> 1) & 7;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(qrow + (((2 * ks + hh) ^ qsw) << 4));
                }
                if (cur.nch == 1) {            // no later barrier of this item: make one, the Q region is rewritten below
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (has_next) issue_q(nxt);
                }
            } else if (c == 1) {
                if (has_next) issue_q(nxt);    // every wave has read its Q fragments (it passed this chunk's barrier)
            }
            ATTS_T(3 + 4 * c);
            if (active) {
                const unsigned char *stage = ring + c_stage * STAGE;
                const int kb_hi = min(nfull, (c + 1) * STEPS);   // blocks without padding keys: the common step
#pragma nounroll
                for (int kb = c * STEPS; kb < kb_hi; ++kb) step(stage, kb - c * STEPS, kb, std::false_type{});
                if (c == cur.nch - 1 && nfull < nkb) step(stage, nfull - c * STEPS, nfull, std::true_type{});
            }
            c_stage = c_stage + 1 == NS ? 0 : c_stage + 1;
            ATTS_T(4 + 4 * c);
        }
        if (active) {
            // Context row: lane (r, hh) holds features 32t + 8g + 4hh + (0..3), i.e. half of each 16-byte group g.  One
            // v_permlane32_swap per packed register gives lane (r, 0) the whole of the even groups and lane (r, 1) the odd
            // ones: 4 stores of 16 bytes (each a 32-byte run per row) instead of 8 of 8.  Issued by instruction: the wait at the
            // next item's first chunk counts them.
            const float inv = 1.0f / (lsum + __shfl_xor(lsum, 32));
            bf16 *crow = a.ctx + (cur.base + cur.q0 + w * 32 + r) * H + cur.head * DH + 8 * hh;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    unsigned pe[2], po[2];   // groups 2j (even) and 2j + 1 (odd), this lane's four features of each, packed
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        pe[h2] = pack_bf16(o[tt][(2 * j) * 4 + 2 * h2] * inv, o[tt][(2 * j) * 4 + 2 * h2 + 1] * inv);
                        po[h2] = pack_bf16(o[tt][(2 * j + 1) * 4 + 2 * h2] * inv, o[tt][(2 * j + 1) * 4 + 2 * h2 + 1] * inv);
                        // lanes 32..63 of pe <-> lanes 0..31 of po: the lower half-wave now has [own even | partner's even] in
                        // (pe, po), the upper one [partner's odd | own odd]
                        const auto sw2 = __builtin_amdgcn_permlane32_swap(pe[h2], po[h2], false, false);
                        pe[h2] = sw2[0];
                        po[h2] = sw2[1];
                    }
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 ov = {pe[0], pe[1], po[0], po[1]};
                    // s_nop: a store of more than 64 bits may not be followed at once by a VALU write of its data registers
                    // (hipcc's hazard recognizer inserts the wait states for its own stores; it cannot see into this one)
                    // (temporal on purpose: with the nt bit the out-projection behind it misses what it finds cached today, +0.6 ms per forward)
                    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" ::"v"(crow), "v"(ov), "n"((32 * tt + 16 * j) * 2) : "memory");
                }
        }
        ATTS_T(20);
        stored = active;
        if (a.fixup && w == 0) {       // (rare path: the drained queue makes the counted waits that follow hold trivially)
            if (lane == 0) a.redo_flags[list0 * NH + t] = 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!has_next) break;
        cur = nxt;
        t = tn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stream's last read-ahead writes LDS: it must land before the workgroup ends
}

#include "attn_pipe.inc"

// Last layer: everything after attention is only needed for the <s> row of each sequence
// (masked_mean_or_first with use_mean=False, src/models.py:52-56): gather those B rows into compact
// matrices and run out-proj, LN, FFN, LN on B rows instead of T.
__global__ __launch_bounds__(256) void gather_cls_kernel(const bf16 *__restrict__ ctx, const float *__restrict__ x, const bf16 *__restrict__ x16, const float2 *__restrict__ xstats,
                                                         const float *__restrict__ xgamma, const float *__restrict__ xbeta, SeqInfo s, int B,
                                                         bf16 *__restrict__ ctx_c, float *__restrict__ x_c) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    if (b < B) {
        const size_t row = (size_t)s.off[b];
        float mean = 0.f, rstd = 1.f;
        if (xstats) {   // x holds pre-LayerNorm rows (deferred LN): normalize the <s> row here
            const float2 st = xstats[row];
            mean = st.x;
            rstd = st.y;
        }
        for (int i = tid; i < H; i += 256) {
            ctx_c[(size_t)b * H + i] = ctx[row * H + i];
            const float v = x16 ? (float)x16[row * H + i] : x[row * H + i];   // the large-batch path keeps the residual stream in bf16
            x_c[(size_t)b * H + i] = xstats ? (v - mean) * rstd * xgamma[i] + xbeta[i] : v;
        }
    } else {  // padding rows of the compact matrices feed the GEMM tiles: keep them finite
        for (int i = tid; i < H; i += 256) {
            ctx_c[(size_t)b * H + i] = (bf16)0.f;
            x_c[(size_t)b * H + i] = 0.f;
        }
    }
}

// ------------------------------------------------------------------ ANCE head: out[b] = LN(W_h . x[row_b] + b_h)   (fp32)
constexpr int CLS_SB = 8;     // sequences per workgroup of the <s>-row projections: a weight row is read once per 8 sequences
constexpr int CLS_NS = 64;    // output features per workgroup (16 per wave): grid = (ceil(B / 8), 768 / 64)
// The two 768 x 768 projections that run on the <s> rows only (the ANCE head and the last layer's queries) are tiny (1.2 GFLOP
// per 1000 sequences) but were 0.43 - 0.45 ms per launch as one workgroup per 8 sequences walking all 768 weight rows (64
// workgroups, 192 dependent iterations per wave: rocprofv3, r03).  Dealt as (8 sequences) x (64 features) they take tens of
// microseconds.  A sequence's arithmetic (products, their order, the reduction) does not depend on its neighbours.

// ANCE head, first half: e[b][n] = W_h[n,:] . x[b,:] + b_h[n] in fp32 (models.py:43)
// ns: output features per workgroup (gridDim.y = H / ns): CLS_NS for large batches; 8 for a handful of sequences, where 12
// workgroups of 64 features each took 40 us of a 1.5 ms forward
__global__ __launch_bounds__(256) void cls_head_proj_kernel(const float *__restrict__ x, SeqInfo s, int compact, int B, const float *__restrict__ Wh,
                                                            const float *__restrict__ bh, float *__restrict__ e_out, int ns) {
    __shared__ float xs[CLS_SB][H];
    const int b0 = blockIdx.x * CLS_SB, n0 = blockIdx.y * ns, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb = min(CLS_SB, B - b0);
    for (int j = 0; j < CLS_SB; ++j) {
        const float *xr = x + (size_t)(compact ? b0 + j : s.off[min(b0 + j, B - 1)]) * H;
        for (int i = tid; i < H; i += 256) xs[j][i] = j < nb ? xr[i] : 0.f;
    }
    __syncthreads();
    for (int n = n0 + w; n < n0 + ns; n += 4) {
        const float *wr = Wh + (size_t)n * H;
        float wv[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) wv[i] = wr[lane + 64 * i];
        const float bias = bh[n];
#pragma unroll
        for (int j = 0; j < CLS_SB; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) acc = fmaf(wv[i], xs[j][lane + 64 * i], acc);
            acc = wave_sum(acc);
            if (lane == 0 && j < nb) e_out[(size_t)(b0 + j) * H + n] = acc + bias;
        }
    }
}

// ANCE head, second half: out[b] = LayerNorm_768(e[b]) (models.py:44); a sequence the device could not encode gets a NaN row
__global__ __launch_bounds__(256) void cls_head_norm_kernel(const float *__restrict__ e_in, SeqInfo s, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps, float *__restrict__ out) {
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *er = e_in + (size_t)b * H;
    float v[3];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i] = er[tid + 256 * i];
        sum += v[i];
    }
    sum = wave_sum(sum);
    if (lane == 0) red[w] = sum;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + w] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) * (1.0f / H) + eps);
    const bool bad = s.err[b] != 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = tid + 256 * i;
        out[(size_t)b * H + c] = bad ? NAN : (v[i] - mean) * rstd * gamma[c] + beta[c];   // unsupported mask or token id of THIS sequence: fail loudly, never guess
    }
}

// Last layer, large-batch path: the query projection of the <s> rows only (the other rows' queries are never used: the
// attention kernel runs the first query block of each sequence and only its row 0 is kept).  Same arithmetic as the QKV GEMM's
// epilogue -- q = rstd (y . W'q^T - mean wsum) + cvec on the bf16 row y and the folded weights -- in fp32, written into row
// off[b] of the big Q matrix.
__global__ __launch_bounds__(256) void cls_q_kernel(const bf16 *__restrict__ yb, const float2 *__restrict__ stats, SeqInfo s, int B,
                                                    const bf16 *__restrict__ Wq, const float *__restrict__ wsum, const float *__restrict__ cvec,
                                                    bf16 *__restrict__ q) {
    __shared__ float xs[CLS_SB][H];
    __shared__ float2 st[CLS_SB];
    const int b0 = blockIdx.x * CLS_SB, n0 = blockIdx.y * CLS_NS, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb = min(CLS_SB, B - b0);
    for (int j = 0; j < CLS_SB; ++j) {
        const size_t row = (size_t)s.off[min(b0 + j, B - 1)];
        for (int i = tid; i < H; i += 256) xs[j][i] = j < nb ? (float)yb[row * H + i] : 0.f;
        if (tid == 0) st[j] = stats[row];
    }
    __syncthreads();
    for (int n = n0 + w; n < n0 + CLS_NS; n += 4) {
        const bf16 *wr = Wq + (size_t)n * H;
        float wv[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) wv[i] = (float)wr[lane + 64 * i];
        const float ws = wsum[n], cv = cvec[n];
#pragma unroll
        for (int j = 0; j < CLS_SB; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) acc = fmaf(wv[i], xs[j][lane + 64 * i], acc);
            acc = wave_sum(acc);
            if (lane == 0 && j < nb) q[(size_t)s.off[b0 + j] * H + n] = (bf16)(st[j].y * (acc - st[j].x * ws) + cv);
        }
    }
}

__global__ void f32_to_bf16_kernel(const float *__restrict__ src, bf16 *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (bf16)src[i];
}

}  // namespace
}  // namespace hac

// =============================================================================
// host side + C ABI
// =============================================================================
using namespace hac;

namespace {

struct LayerW {
    bf16 *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
    float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
    // large-batch path (gemm8.inc): the preceding LayerNorm folded into the weights that consume its output
    bf16 *wqkv8 = nullptr, *w18 = nullptr;      // [2304][768] (q rows also carry log2(e)/8), [3072][768]
    float *fold = nullptr;                      // wsum_qkv[2304] | cvec_qkv[2304] | wsum_1[3072] | cvec_1[3072]
};

}  // namespace

struct hac_encoder {
    hac_encoder_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<std::string, float *> raw;       // device fp32 copies of the checkpoint tensors
    std::map<std::string, size_t> raw_count;
    std::vector<LayerW> layers;
    float *word = nullptr, *posw = nullptr, *typew = nullptr, *embg = nullptr, *embb = nullptr;
    float *wh = nullptr, *bh = nullptr, *ng = nullptr, *nb = nullptr;
    bool finalized = false;
    // workspace
    GrowBuf ws_x, ws_xb, ws_q, ws_k, ws_vt, ws_ctx, ws_y, ws_h, ws_seq, ws_ids, ws_mask, ws_out, ws_cls, ws_stats;
    GrowBuf ws_ksplit;                    // classic path, small batches: split-K partial sums of the RESID GEMMs
    GrowBuf ws_yb, ws_part, ws_idstats;   // gemm8 path: bf16 copy of the attention-block rows, row-sum partials, (0, 1) statistics
    size_t idstats_rows = 0;
    int attn_mode = 0;                    // 0: streaming single-pass attention; 1: two-pass kernels (cross-check)
    int attn_pipe = -1;                   // streaming attention of whole items (no query split, not the <s>-only layer): two query blocks per wave, woven (attn_pipe.inc); 0: the one-block kernel everywhere
    int plan_attn_pipe = 0;               // what the most recent forward's layers used
    int gemm_mode = -1;                   // -1: by size, 0: classic kernels only, 1: gemm8 whenever the batch has a full tile (tests)
    // gemm8 loop form per class (bit 0 QKV, 1 out-proj, 2 FFN-up, 3 FFN-down; 1 = SPLIT, 0 = round 2's loop, kept for A/B runs).
    // A/B in one process on the 1000 x 512 forward (tools/ab_encoder.py), two boxes: SPLIT -1 .. -3.4 % on FFN-down (K = 3072),
    // +-1 % (inside the run-to-run spread) on the K = 768 GEMMs; layer stack 93.4 - 94.9 ms with every class split vs 95.3 - 96.7 with none
    int g8_split = 15;
    int attn_qsplit = -1;  // -1 auto (small batches: query rows of an item dealt to 2 or 4 workgroups), 0 off
    int g8_stagger = -1;   // -1 auto (phased workgroup starts of the K = 768 RESID / QKV classes on long tile runs), 0 off
    void *h_pin = nullptr;
    size_t h_pin_bytes = 0;
    int *h_len = nullptr;      // pinned: padded lengths of a forward that runs as several sub-batches
    size_t h_len_cap = 0;
    // profiling (bench): hipEvent pairs on the launch stream.  Bit 0 of prof_mask: around the layer stack of each
    // forward (pool 0); bit 1+c: around every launch of kernel class c (pool 1+c), see HAC_ENC_CLASS_* in the header.
    unsigned prof_mask = 0;
    struct EvPool {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
        size_t used = 0;
    } pools[1 + HAC_ENC_NCLASS];
    long max_tokens = 524288;  // packed rows per sub-batch (workspaces: ~9 GB).  1000 x 512 then runs as ONE pass: no host read-back of the
                               // lengths at all, -0.6 % against two passes of 262144 with round 3's kernels (196608: +0.9 %)
    int n_cu = 256;
    // what the most recent forward ran (hac_encoder_last_plan)
    const char *plan_gemm = "none";
    int plan_sub_batches = 0;
    long plan_rows = 0;
    const char *plan_graph = "off";
    char last_plan[160] = "none";
    // Small batches (the reference's own call shape is 4 queries per GPU, test_HAConvDR_topiocqa.py:173,406) are bound by
    // launches, not arithmetic: ~110 kernels for ~0.4 TFLOP.  Their forward is captured ONCE per (B, L, options) into a HIP
    // graph over private input / output buffers and replayed: one graph launch + three small copies per call.
    int graph_mode = -1;                  // -1: small batches without profiling, 0: never
    int attn_qs_pin = 0;                  // development ("attn_qs_pin" = 1 | 2 | 4 | 8 | 16; 0: by the rule)
    int ks_pin_out = 0, ks_pin_down = 0;  // development ("ksplit_pin" = "a/b"): the slices of out-proj / FFN-down pinned (0: by the model)
    int ksplit_mode = -1;                 // -1: split-K of the small-batch RESID GEMMs by tile count, 0: never (tests that compare batches of different sizes bit for bit)
    int plan_ks_out = 1, plan_ks_down = 1;
    struct GraphEntry {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        uint64_t sig = 0;                 // digest of every workspace pointer the captured launches hold
        int seen = 0;                     // eager passes so far (the first call sizes the workspaces)
        const char *gemm = "none";
        long rows = 0;
        int ks_out = 1, ks_down = 1;      // the plan of the captured forward (a replay runs no host-side planning)
        int attn_pipe = 0;
    };
    std::map<uint64_t, GraphEntry> graphs;
    hipEvent_t graph_done = nullptr;      // recorded behind every replay on the caller's stream: an exec is destroyed only after it
    bool graph_done_armed = false;
    GrowBuf ws_gids, ws_gmask, ws_gout;
    GrowBuf ws_identgb;                   // [2][768]: gamma = 1, beta = 0
    GrowBuf ws_redo;                      // [16] per-layer counts | [B * 12] item flags of the attention fix-up pass (attn_pipe.inc)
    // Layers whose items mostly fail the woven kernel's check (near one-hot attention: a property of the weights more than of the batch) would
    // pay for both kernels every time (measured with logits x 100: 28.4 against 14.4 ms per forward).  The per-layer counts of a forward come
    // back through a pinned copy and an event nobody waits for; a later forward that finds them routes such layers through the one-block kernel
    // at once (same bits either way) and tries the woven form again every 64th forward.
    int *h_redo = nullptr;                // pinned [16]
    hipEvent_t redo_ev = nullptr;
    bool redo_pending = false;
    long redo_items = 0;                  // items per layer (B x 12) of the forward whose counts are pending
    unsigned pipe_skip_mask = 0;          // bit l: layer l goes straight to the one-block kernel
    unsigned forwards_since_retry = 0;
    GrowBuf ws_clk;                       // [4] u64: hac_encoder_last_clock
    bool clk_valid = false;
    hipEvent_t clk_ev = nullptr;          // recorded behind the launch that wrote ws_clk: what hac_encoder_last_clock waits for (not the whole device)
};

namespace {

int enc_fail_missing(const std::string &name) { return fail(HAC_ERR_INVALID, "encoder weight '%s' was never set", name.c_str()); }

int get_raw(hac_encoder *e, const std::string &name, size_t count, float **out) {
    auto it = e->raw.find(name);
    if (it == e->raw.end()) return enc_fail_missing(name);
    if (e->raw_count[name] != count) return fail(HAC_ERR_INVALID, "encoder weight '%s' has %zu elements, expected %zu", name.c_str(), e->raw_count[name], count);
    *out = it->second;
    return HAC_OK;
}

int to_bf16(hac_encoder *e, const float *src, size_t n, bf16 **out) {
    HAC_HIP(hipMalloc((void **)out, n * sizeof(bf16)));
    f32_to_bf16_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream>>>(src, *out, n);
    HAC_HIP(hipGetLastError());
    return HAC_OK;
}

int prof_begin(hac_encoder *e, int pool, hipStream_t st) {
    if (!((e->prof_mask >> pool) & 1u)) return HAC_OK;
    auto &pl = e->pools[pool];
    if (pl.used == pl.ev.size()) {
        hipEvent_t a0, a1;
        HAC_HIP(hipEventCreate(&a0));
        HAC_HIP(hipEventCreate(&a1));
        pl.ev.emplace_back(a0, a1);
    }
    HAC_HIP(hipEventRecord(pl.ev[pl.used].first, st));
    return HAC_OK;
}
int prof_end(hac_encoder *e, int pool, hipStream_t st) {
    if (!((e->prof_mask >> pool) & 1u)) return HAC_OK;
    auto &pl = e->pools[pool];
    HAC_HIP(hipEventRecord(pl.ev[pl.used].second, st));
    ++pl.used;
    return HAC_OK;
}

// carve the sequence bookkeeping of a (sub-)batch out of ws_seq
int seq_layout(hac_encoder *e, int B, int L, SeqInfo &s) {
    const size_t seq_ints = (size_t)4 * B + (size_t)5 * B + 8 + (size_t)B * L;
    HAC_TRY(e->ws_seq.reserve(seq_ints * 4));
    int *p = (int *)e->ws_seq.p;
    s.desc = p;                 // 4 B entries, 16-byte aligned (hipMalloc's alignment)
    p += (size_t)4 * B;
    s.lens = p;
    s.len32 = p + B;
    s.off = p + 2 * B;          // B+1 entries
    s.nb = p + 3 * B + 2;
    s.err = p + 3 * B + 4;      // B entries
    s.order = p + 4 * B + 4;    // B entries
    s.ncls = p + 5 * B + 4;     // 2 entries
    s.pos = p + 5 * B + 6;
    return HAC_OK;
}

// rows_hint: an upper bound of the packed rows of this sub-batch when the caller knows one (sum of its
// sequences' padded lengths), 0 = every sequence may be full length
// family: the GEMM family of the whole hac_encoder_forward* call -- FAM_CLASSIC128 / FAM_CLASSIC256 / FAM_GEMM8 -- or -1:
// decide here, from this (only) sub-batch's rows
enum { FAM_CLASSIC128 = 0, FAM_GEMM8 = 1, FAM_CLASSIC256 = 2 };
// 256^2 tiles (gemm8's folded-LayerNorm, bf16-residual path) from this many out-proj tiles on: 9472 rows.  Measured at L = 512
// (tools/ab_option.py gemm classic 8phase): 16 x 512 classic 2.58 ms against 2.68, 20 x 512 2.91 against 2.75, 24 x 512 3.83 / 3.26.
constexpr long FAMILY_BIG_TILES = 111;
int pick_family(const hac_encoder *e, long rows) {
    const long Mp = (rows + MT - 1) / MT * MT;
    const bool big = (Mp / 256) * (H / 256) >= FAMILY_BIG_TILES;
    if (e->gemm_mode == 1 || (e->gemm_mode < 0 && big)) return FAM_GEMM8;
    return big ? FAM_CLASSIC256 : FAM_CLASSIC128;
}
template <typename IT>
int run_forward(hac_encoder *e, const IT *ids, const IT *mask, int B, int L, float *out_dev, hipStream_t st, long rows_hint = 0, int family = -1,
                long rows_plan = 0) {
    const hac_encoder_config &c = e->cfg;
    const int L32 = (L + SEQ_ALIGN - 1) / SEQ_ALIGN * SEQ_ALIGN;
    bool fw_capturing = false;           // (inside a stream capture -- the small-batch graphs -- an event query is an error that kills the capture)
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
        else fw_capturing = cs != hipStreamCaptureStatusNone;
    }
    if (e->redo_pending && !fw_capturing) {
        const hipError_t qe = hipEventQuery(e->redo_ev);
        if (qe == hipSuccess) {          // the fix-up counts of an earlier forward have arrived (never waited for)
            e->redo_pending = false;
            for (int l = 0; l < 16; ++l)     // (a wave counts its item once, up to 8 waves per item: "most items" = more than two counts per item)
                if ((long)e->h_redo[l] > 2 * e->redo_items) e->pipe_skip_mask |= 1u << l;
        } else {
            (void)hipGetLastError();     // (not ready: the query's hipErrorNotReady must not become the next check's error)
        }
    }
    if (++e->forwards_since_retry >= 64) {
        e->forwards_since_retry = 0;
        e->pipe_skip_mask = 0;
    }
    const long rows_max = rows_hint > 0 ? std::min<long>(rows_hint, (long)B * L32) : (long)B * L32;
    const long Mp = (rows_max + MT - 1) / MT * MT;
    HAC_TRY(e->ws_x.reserve((size_t)Mp * H * 4));
    HAC_TRY(e->ws_y.reserve((size_t)Mp * H * 4));
    HAC_TRY(e->ws_stats.reserve((size_t)Mp * 8 * 2));
    HAC_TRY(e->ws_xb.reserve((size_t)Mp * H * 2));
    HAC_TRY(e->ws_q.reserve((size_t)Mp * H * 2));
    HAC_TRY(e->ws_k.reserve((size_t)(Mp + 64) * H * 2));
    HAC_TRY(e->ws_vt.reserve((size_t)H * Mp * 2));
    HAC_TRY(e->ws_ctx.reserve((size_t)Mp * H * 2));
    HAC_TRY(e->ws_h.reserve((size_t)Mp * FF * 2));
    SeqInfo s;
    HAC_TRY(seq_layout(e, B, L, s));
    seq_prep_kernel<IT><<<dim3(B), dim3(512), 0, st>>>(ids, mask, L, s, c.pad_token_id, c.vocab);
    seq_offsets_kernel<<<dim3(1), dim3(256), 0, st>>>(s, B);
    attn_order_kernel<<<dim3(1), dim3(256), 0, st>>>(s, B);
    float *x = (float *)e->ws_x.p, *y = (float *)e->ws_y.p;
    float2 *statsA = (float2 *)e->ws_stats.p, *statsF = statsA + Mp;
    bf16 *xb = (bf16 *)e->ws_xb.p, *q = (bf16 *)e->ws_q.p, *k = (bf16 *)e->ws_k.p, *vt = (bf16 *)e->ws_vt.p;
    bf16 *ctx = (bf16 *)e->ws_ctx.p, *h = (bf16 *)e->ws_h.p;
    // tile choice: 256^2 tiles once they fill the chip, 128^2 tiles for small batches; persistent grids.
    // Large batches: the ping-pong GEMM with the LayerNorms folded into the consuming weights (gemm8.inc).  The family -- and
    // with the classic kernels the tile -- is chosen ONCE per hac_encoder_forward* call, from the whole batch
    // (forward_batched): a small tail sub-batch of a large forward runs the same kernels as the others, so a sequence's
    // embedding does not depend on which sub-batch it fell into.
    if (family < 0) family = pick_family(e, rows_max);
    const bool g8 = family == FAM_GEMM8;
    const bool big = family == FAM_CLASSIC256 || (g8 && (Mp / 256) * (H / 256) >= FAMILY_BIG_TILES);
    // (that path's residual stream is bf16 from the embedding rows on: no fp32 copy of them, 3 KB per token less to write)
    embed_ln_kernel<IT><<<dim3(L32 / 4, B), dim3(256), 0, st>>>(ids, L, s, e->word, e->posw, e->typew, e->embg, e->embb, c.ln_eps, c.vocab, g8 ? nullptr : x, xb);
    HAC_HIP(hipGetLastError());
    const int *total = s.off + B;
    // dead tail rows of the last M tile (fewer than MT) feed the GEMMs: keep them finite
    zero_tail_rows_kernel<<<dim3(MT), dim3(192), 0, st>>>(g8 ? nullptr : x, xb, total, Mp);
    HAC_HIP(hipGetLastError());
    const int bt = big ? 256 : 128;
    const size_t lds = (size_t)4 * bt * 128 + (size_t)(big ? 8 : 4) * 4096;   // 2 stages + per-wave patches
    // Small batches (128^2 tiles): the two RESID GEMMs have only Mp / 128 x 6 output tiles -- 96 for the reference's 4 x 512 query
    // batch, on 256 CUs -- and walk their whole K behind one exposed load latency per k-tile, so their K loop is split into S
    // slices (GemmArgs::ksplit; ln_stats_rows_kernel adds the slices up).  How far: a slice more saves k-tiles at ~0.6 us each
    // and costs one more fp32 copy of the rows written and read back, ~1.3 us per 1000 rows; below 8 (K = 3072) / 4 (K = 768)
    // k-tiles per item nothing is gained, and more items than workgroup slots (2 per CU) is a second round of them.  The minimum
    // of that model is within 0.3 % of the best of all 28 (S_out, S_down) pairs at 1 x 256, 4 x 256, 4 x 512, 8 x 384 and 8 x 512
    // (tools/ks_sweep.py on the whole forward, profiles/r05_ksplit_sweep.txt; tools/probes/gemm_small_probe.hip for the kernels
    // alone).  Round 4's rule -- split until 1.5 items per CU exist -- went 2-3 x too far on the smallest batches: 1 x 256 ran
    // 4 / 16 slices (0.962 ms) where 3 / 6 is 0.893.
    // The split -- i.e. the summation order -- is decided ONCE per hac_encoder_forward* call, from the rows of its first
    // sub-batch (rows_plan), like the family: a sequence's embedding must not depend on the sub-batch it fell into.
    const long Mp_plan = rows_plan > 0 ? (rows_plan + MT - 1) / MT * MT : Mp;
    auto pick_ksplit = [&](int K) {
        if (g8 || big || e->ksplit_mode == 0) return 1;
        const int KT = K / 64, min_kt = KT >= 48 ? 8 : 4;
        const int pin = K == H ? e->ks_pin_out : e->ks_pin_down;
        if (pin > 0 && KT % pin == 0 && KT / pin >= 3) return pin;
        const long tiles = (Mp_plan / 128) * (H / 128);
        int S = 1;
        double best = KT * 0.6;
        for (int cand : {2, 3, 4, 6}) {
            if (KT % cand || KT / cand < min_kt || tiles * cand > 2L * e->n_cu) continue;
            const double cost = (double)KT / cand * 0.6 + (cand - 1) * (double)Mp_plan * 1.3e-3;
            if (cost < best) {
                best = cost;
                S = cand;
            }
        }
        return S;
    };
    const int ks_out = pick_ksplit(H), ks_down = pick_ksplit(FF);
    e->plan_ks_out = ks_out;
    e->plan_ks_down = ks_down;
    const size_t part_stride = (size_t)Mp * H;
    if (std::max(ks_out, ks_down) > 1) HAC_TRY(e->ws_ksplit.reserve((size_t)(std::max(ks_out, ks_down) - 1) * part_stride * 4));
    float *kpart = (float *)e->ws_ksplit.p;
    const dim3 blk(big ? 512 : 256);
    const unsigned n_wg = (unsigned)(e->n_cu * (big ? 1 : 2));
#define HAC_GEMM(EPI, CLS)                                                            \
    do {                                                                              \
        HAC_TRY(prof_begin(e, 1 + (CLS), st));                                        \
        if (big) gemm_bf16_nt_kernel<EPI, 4><<<dim3(n_wg), blk, lds, st>>>(g);         \
        else gemm_bf16_nt_kernel<EPI, 2><<<dim3(n_wg), blk, lds, st>>>(g);             \
        HAC_TRY(prof_end(e, 1 + (CLS), st));                                          \
    } while (0)
    bf16 *yAb = nullptr;
    float2 *part = nullptr, *idstats = nullptr;
    if (g8) {
        HAC_TRY(e->ws_yb.reserve((size_t)Mp * H * 2));
        HAC_TRY(e->ws_part.reserve((size_t)Mp * (H / 64) * 8));
        if (e->idstats_rows < (size_t)Mp) {
            HAC_TRY(e->ws_idstats.reserve((size_t)Mp * 8));
            e->idstats_rows = e->ws_idstats.cap / 8;
            fill_identity_stats_kernel<<<dim3((unsigned)((e->idstats_rows + 255) / 256)), dim3(256), 0, st>>>((float2 *)e->ws_idstats.p, e->idstats_rows);
            HAC_HIP(hipGetLastError());
        }
        yAb = (bf16 *)e->ws_yb.p;
        part = (float2 *)e->ws_part.p;
        idstats = (float2 *)e->ws_idstats.p;
        if (!e->ws_identgb.p) {   // gamma = 1 | beta = 0: residual rows that are final (layer 0's embedding rows) take the same epilogue
            HAC_TRY(e->ws_identgb.reserve((size_t)2 * H * 4));
            std::vector<float> gb((size_t)2 * H, 0.f);
            std::fill(gb.begin(), gb.begin() + H, 1.f);
            HAC_HIP(hipMemcpy(e->ws_identgb.p, gb.data(), gb.size() * 4, hipMemcpyHostToDevice));
        }
    }
    HAC_TRY(prof_begin(e, 0, st));
    // compact buffers of the CLS-only tail of the last layer
    const long Mc = ((long)B + MT - 1) / MT * MT;
    HAC_TRY(e->ws_cls.reserve((size_t)Mc * (H * 2 + H * 4 * 3 + H * 2 + FF * 2)));
    bf16 *ctx_c = (bf16 *)e->ws_cls.p;
    float *x_c = (float *)(ctx_c + Mc * H);
    float *y_c = x_c + Mc * H;
    float *x2_c = y_c + Mc * H;
    bf16 *xb_c = (bf16 *)(x2_c + Mc * H);
    bf16 *h_c = xb_c + Mc * H;
    for (int li = 0; li < c.n_layers; ++li) {
        const LayerW &w = e->layers[li];
        const bool last = (li == c.n_layers - 1);
        GemmArgs g{};
        g.total_rows = total;
        Gemm8Args g8a{};
        g8a.total_rows = total;
        g8a.n_groups = 1;
        const int ng_up = 2;   // FFN-up: W1' is 4.7 MB against 4 MB of L2 per XCD; each XCD owns half of its column tiles (measured: -2 %)
        const dim3 grid8((unsigned)e->n_cu), blk8(512);
        // Phased starts (Gemm8Args::stagger): workgroups that start together stay together -- every tile takes the same time --
        // and reach their epilogues, the phases that write (and, RESID, read) the residual stream, all at once.  Delaying the
        // XCDs by 0..3 quarter-steps at the start spreads those bursts: QKV -3.8 % and out-proj -3.5 % at 512000 rows
        // (tools/probes/gemm8_stagger_probe.hip: 1.786 -> 1.718 ms, 0.803 -> 0.775 ms); the FFN classes did not move (their
        // tile times are longer and already drift apart) and stay unphased.  Only on runs of >= 16 tiles per workgroup: the
        // delay (<= 3 steps of ~1 us x stagger) is paid once per launch.
        auto stagger8 = [&](int N, int n_tile0, int steps) {
            const long tiles = (Mp / 256) * (long)(N / 256 - n_tile0);
            g8a.stagger = (e->g8_stagger != 0 && tiles >= 16L * e->n_cu) ? steps : 0;
            g8a.stagger_mode = 0;
        };
        // gemm8.inc's two loop forms: SPLIT (operand-split DMA roles, 160 KiB) / round 2's (128 KiB); one bit of g8_split per class
        auto launch8 = [&](auto epi, int cls_bit) {
            constexpr int EPI = decltype(epi)::value;
            if ((e->g8_split >> cls_bit) & 1) gemm8_kernel<EPI, true><<<grid8, blk8, 163840, st>>>(g8a);
            else gemm8_kernel<EPI, false><<<grid8, blk8, 131072, st>>>(g8a);
        };
        constexpr std::integral_constant<int, EPI8_QKV> epi_qkv{};
        constexpr std::integral_constant<int, EPI8_RESID> epi_resid{};
        constexpr std::integral_constant<int, EPI8_GELU> epi_gelu{};
        // QKV
        if (g8) {
            // A = the previous layer's un-normalized output rows (bf16) + their statistics; layer 0: the normalized embedding rows
            HAC_TRY(prof_begin(e, 1 + HAC_ENC_CLASS_QKV, st));
            g8a.A = xb; g8a.K = H; g8a.astats = li ? statsF : idstats;
            g8a.W = w.wqkv8; g8a.N = 3 * H; g8a.wsum = w.fold; g8a.cvec = w.fold + 3 * H; g8a.q = q; g8a.k = k; g8a.v16 = vt;
            // last layer: keys and values of every row, queries of the <s> rows only (a third of the GEMM: 0.6 ms per 1000 x 512 forward)
            g8a.n_tile0 = last ? H / 256 : 0;
            stagger8(3 * H, g8a.n_tile0, 6);
            launch8(epi_qkv, 0);
            g8a.n_tile0 = 0;
            g8a.stagger = 0;
            if (last) cls_q_kernel<<<dim3((unsigned)((B + CLS_SB - 1) / CLS_SB), H / CLS_NS), dim3(256), 0, st>>>(xb, g8a.astats, s, B, w.wqkv8, w.fold, w.fold + 3 * H, q);
            HAC_TRY(prof_end(e, 1 + HAC_ENC_CLASS_QKV, st));
        } else {
            g.A = xb; g.W = w.wqkv; g.bias = w.bqkv; g.N = 3 * H; g.K = H; g.q = q; g.k = k; g.v16 = vt;
            HAC_GEMM(EPI_QKV, HAC_ENC_CLASS_QKV);
        }
        // (streaming kernels: few sequences -> an item's query rows go to 2 or 4 workgroups while B * NH * qsplit items still fit the CUs)
        int att_qs = 1;
        while (e->attn_qsplit != 0 && att_qs < 16 && (long)B * NH * att_qs * 2 <= e->n_cu) att_qs *= 2;
        if (e->attn_qs_pin > 0) att_qs = e->attn_qs_pin;   // development (tools/ks_sweep.py attn)
        const bool att_one = att_qs > 1 && L32 > 256;   // few sequences: one launch (an empty second one is 5 us of a ~100-us layer)
        AttnArgs a{q, k, vt, ctx, s, last ? 1 : 0, att_qs, att_one ? 1 : 0, nullptr, nullptr, 0};
        if (li == 0) e->plan_attn_pipe = 0;
        // (measured, 512 sequences of one length, woven / one-block: 512 rows 0.475 / 0.589 ms, 448 rows 0.413 / 0.507, 384 rows 0.313 / 0.373;
        // 256 rows 0.174 / 0.174, 128 rows 0.086 / 0.086 -- the 8-wave instantiation wins wherever it applies, the 4-wave one ties: the woven
        // form takes the long class (sequences of more than 256 rows), the one-block kernel the short class; same bits either way)
        const bool att_pipe = e->attn_mode == 0 && e->attn_pipe != 0 && att_qs == 1 && !last && (L32 > 256 || e->attn_pipe > 0) &&
                              !(e->attn_pipe < 0 && ((e->pipe_skip_mask >> li) & 1u));      // ("all" pins the woven form: tests of the fix-up pass)
        if (att_pipe) {
            if (!e->plan_attn_pipe) {      // the forward's first woven layer: workspace, per-layer counts to zero (the flags are zero whenever no pass is pending)
                HAC_TRY(e->ws_redo.reserve(((size_t)B * NH + 16) * 4));
                HAC_HIP(hipMemsetAsync(e->ws_redo.p, 0, 64, st));
            }
            a.redo_count = (int *)e->ws_redo.p + (li & 15);
            a.redo_flags = (int *)e->ws_redo.p + 16;
        }
        // sequences of <= 256 rows: 4-wave workgroups; longer ones: 8-wave workgroups (each skips the other's)
        HAC_TRY(prof_begin(e, 1 + HAC_ENC_CLASS_ATTN, st));
        if (e->attn_mode == 0) {           // persistent streaming kernels, one launch per length class
            if (att_pipe) {   // whole items: the woven two-blocks-per-wave form, then the items it flagged through the one-block kernels (bit-identical results)
                const bool both = e->attn_pipe > 0;   // "all" (tests): the short class through the 4-wave instantiation too
                if (L32 > 256) attention_pipe_kernel<8><<<dim3(e->n_cu), dim3(512), 163840, st>>>(a);
                if (both) attention_pipe_kernel<4><<<dim3(2 * e->n_cu), dim3(256), 81920, st>>>(a);
                else attention_stream_kernel<8><<<dim3(2 * e->n_cu), dim3(512), 81920, st>>>(a);
                a.fixup = 1;
                if (L32 > 256) attention_stream_kernel<16><<<dim3(e->n_cu), dim3(1024), 163840, st>>>(a);
                if (both) attention_stream_kernel<8><<<dim3(2 * e->n_cu), dim3(512), 81920, st>>>(a);
                e->plan_attn_pipe = 1;
            } else {
                if (L32 > 256) attention_stream_kernel<16><<<dim3(e->n_cu), dim3(1024), 163840, st>>>(a);
                if (!att_one) attention_stream_kernel<8><<<dim3(2 * e->n_cu), dim3(512), 81920, st>>>(a);
            }
        } else {                           // two-pass kernels, one workgroup per (sequence, head): kept as a cross-check
            attention_kernel<8, 1><<<dim3(NH, B), dim3(512), (size_t)(L32 < 256 ? L32 : 256) * 256, st>>>(a);
            if (L32 > 256) attention_kernel<16, 1><<<dim3(NH, B), dim3(1024), (size_t)L32 * 256, st>>>(a);
        }
        HAC_TRY(prof_end(e, 1 + HAC_ENC_CLASS_ATTN, st));
        // Residual stream between layers: layer 0 reads the embedding rows x (normalized); afterwards the
        // stream lives as pre-LayerNorm rows + (mean, rstd): yF/statsF after a layer's FFN, yA/statsA after its
        // attention block.  yF shares x's buffer (x is dead once layer 0's out-projection has read it).
        const bool defer_in = li > 0;
        const float *ln2g_prev = defer_in ? e->layers[li - 1].ln2g : nullptr, *ln2b_prev = defer_in ? e->layers[li - 1].ln2b : nullptr;
        if (!last && g8) {
            // attention output projection + residual -> yA (bf16) and row-sum partials of its fp32 values -> (mean, rstd)
            g8a.A = ctx; g8a.W = w.wo; g8a.N = H; g8a.K = H; g8a.cvec = w.bo; g8a.resid = xb; g8a.yb = yAb; g8a.part = part;
            g8a.rstats = defer_in ? statsF : idstats;
            g8a.rgamma = defer_in ? ln2g_prev : (const float *)e->ws_identgb.p;
            g8a.rbeta = defer_in ? ln2b_prev : (const float *)e->ws_identgb.p + H;
            HAC_TRY(prof_begin(e, 1 + HAC_ENC_CLASS_OUTPROJ, st));
            stagger8(H, 0, 8);
            launch8(epi_resid, 1);
            g8a.stagger = 0;
            HAC_TRY(prof_end(e, 1 + HAC_ENC_CLASS_OUTPROJ, st));
            HAC_TRY(prof_begin(e, 1 + HAC_ENC_CLASS_LN, st));
            ln_combine_kernel<<<dim3((unsigned)(Mp / 256)), dim3(256), 0, st>>>(part, H / 64, H, total, c.ln_eps, statsA);
            HAC_TRY(prof_end(e, 1 + HAC_ENC_CLASS_LN, st));
            // FFN up: A = bf16(yA), the attention LayerNorm folded into W1
            g8a.A = yAb; g8a.astats = statsA; g8a.W = w.w18; g8a.N = FF; g8a.K = H; g8a.wsum = w.fold + 6 * H; g8a.cvec = w.fold + 6 * H + FF; g8a.h = h;
            g8a.n_groups = ng_up;
            if (e->prof_mask >> 1) {   // class profiling on: this launch also reads the clock counters (hac_encoder_last_clock)
                HAC_TRY(e->ws_clk.reserve(32));
                g8a.clk = (unsigned long long *)e->ws_clk.p;
                e->clk_valid = true;
            }
            HAC_TRY(prof_begin(e, 1 + HAC_ENC_CLASS_FFN_UP, st));
            launch8(epi_gelu, 2);
            if (g8a.clk) {      // (class profiling is never on inside a captured forward: graph_usable)
                if (!e->clk_ev) HAC_HIP(hipEventCreateWithFlags(&e->clk_ev, hipEventDisableTiming));
                HAC_HIP(hipEventRecord(e->clk_ev, st));
            }
            g8a.clk = nullptr;
            HAC_TRY(prof_end(e, 1 + HAC_ENC_CLASS_FFN_UP, st));
            g8a.n_groups = 1;
            // FFN down + residual LN1(yA) -> yF (bf16, in xb's buffer: the next layer's A operand and residual), partials
            g8a.A = h; g8a.W = w.w2; g8a.N = H; g8a.K = FF; g8a.cvec = w.b2; g8a.resid = yAb; g8a.yb = xb;
            g8a.rstats = statsA; g8a.rgamma = w.ln1g; g8a.rbeta = w.ln1b;
            HAC_TRY(prof_begin(e, 1 + HAC_ENC_CLASS_FFN_DOWN, st));
            launch8(epi_resid, 3);
            HAC_TRY(prof_end(e, 1 + HAC_ENC_CLASS_FFN_DOWN, st));
            HAC_TRY(prof_begin(e, 1 + HAC_ENC_CLASS_LN, st));
            ln_combine_kernel<<<dim3((unsigned)(Mp / 256)), dim3(256), 0, st>>>(part, H / 64, H, total, c.ln_eps, statsF);
            HAC_TRY(prof_end(e, 1 + HAC_ENC_CLASS_LN, st));
        } else if (!last) {
            // attention output projection + residual, LN statistics
            g.A = ctx; g.W = w.wo; g.bias = w.bo; g.N = H; g.K = H; g.resid = x; g.y = y;
            g.rstats = defer_in ? statsF : nullptr; g.rgamma = ln2g_prev; g.rbeta = ln2b_prev;
            g.ksplit = ks_out; g.part = kpart; g.part_stride = part_stride;
            HAC_GEMM(EPI_RESID, HAC_ENC_CLASS_OUTPROJ);
            g.ksplit = 1;
            HAC_TRY(prof_begin(e, 1 + HAC_ENC_CLASS_LN, st));
            ln_stats_rows_kernel<<<dim3((unsigned)(Mp / 4)), dim3(256), 0, st>>>(y, total, w.ln1g, w.ln1b, c.ln_eps, statsA, xb, kpart, ks_out - 1, part_stride);
            HAC_TRY(prof_end(e, 1 + HAC_ENC_CLASS_LN, st));

            // FFN
            g.A = xb; g.W = w.w1; g.bias = w.b1; g.N = FF; g.K = H; g.h = h;
            HAC_GEMM(EPI_GELU, HAC_ENC_CLASS_FFN_UP);
            g.A = h; g.W = w.w2; g.bias = w.b2; g.N = H; g.K = FF; g.resid = y; g.y = x;
            g.rstats = statsA; g.rgamma = w.ln1g; g.rbeta = w.ln1b;
            g.ksplit = ks_down; g.part = kpart; g.part_stride = part_stride;
            HAC_GEMM(EPI_RESID, HAC_ENC_CLASS_FFN_DOWN);
            g.ksplit = 1;
            HAC_TRY(prof_begin(e, 1 + HAC_ENC_CLASS_LN, st));
            ln_stats_rows_kernel<<<dim3((unsigned)(Mp / 4)), dim3(256), 0, st>>>(x, total, w.ln2g, w.ln2b, c.ln_eps, statsF, xb, kpart, ks_down - 1, part_stride);
            HAC_TRY(prof_end(e, 1 + HAC_ENC_CLASS_LN, st));

        } else {
            // only the <s> row of every sequence continues (B rows instead of T): same kernels, compact matrices
            gather_cls_kernel<<<dim3((unsigned)Mc), dim3(256), 0, st>>>(ctx, x, g8 ? xb : nullptr, defer_in ? statsF : nullptr, ln2g_prev, ln2b_prev, s, B, ctx_c, x_c);
            const size_t lds_s = (size_t)4 * 128 * 128 + (size_t)4 * 4096;
            const dim3 grid_s((unsigned)(e->n_cu * 2)), blk_s(256);
            g.total_rows = s.nb;
            g.A = ctx_c; g.W = w.wo; g.bias = w.bo; g.N = H; g.K = H; g.resid = x_c; g.y = y_c; g.rstats = nullptr;
            gemm_bf16_nt_kernel<EPI_RESID, 2><<<grid_s, blk_s, lds_s, st>>>(g);
            ln_rows_kernel<<<dim3((unsigned)(Mc / 4)), dim3(256), 0, st>>>(y_c, s.nb, w.ln1g, w.ln1b, c.ln_eps, x2_c, xb_c);
            g.A = xb_c; g.W = w.w1; g.bias = w.b1; g.N = FF; g.K = H; g.h = h_c;
            gemm_bf16_nt_kernel<EPI_GELU, 2><<<grid_s, blk_s, lds_s, st>>>(g);
            g.A = h_c; g.W = w.w2; g.bias = w.b2; g.N = H; g.K = FF; g.resid = x2_c; g.y = y_c;
            gemm_bf16_nt_kernel<EPI_RESID, 2><<<grid_s, blk_s, lds_s, st>>>(g);
            ln_rows_kernel<<<dim3((unsigned)(Mc / 4)), dim3(256), 0, st>>>(y_c, s.nb, w.ln2g, w.ln2b, c.ln_eps, x_c, xb_c);
        }
        HAC_HIP(hipGetLastError());
    }
#undef HAC_GEMM
    HAC_TRY(prof_end(e, 0, st));
    // ANCE head on the compact <s> rows: projection (y_c is free again), then LayerNorm_768 and the per-sequence error flag
    const int head_ns = B <= 64 ? 8 : CLS_NS;
    cls_head_proj_kernel<<<dim3((unsigned)((B + CLS_SB - 1) / CLS_SB), H / head_ns), dim3(256), 0, st>>>(x_c, s, 1, B, e->wh, e->bh, y_c, head_ns);
    cls_head_norm_kernel<<<dim3((unsigned)B), dim3(256), 0, st>>>(y_c, s, e->ng, e->nb, 1e-5f, out_dev);
    HAC_HIP(hipGetLastError());
    if (e->plan_attn_pipe && e->attn_pipe < 0 && !e->redo_pending) {
        if (!fw_capturing) {
            if (!e->h_redo) HAC_HIP(hipHostMalloc((void **)&e->h_redo, 64, hipHostMallocDefault));
            if (!e->redo_ev) HAC_HIP(hipEventCreateWithFlags(&e->redo_ev, hipEventDisableTiming));
            HAC_HIP(hipMemcpyAsync(e->h_redo, e->ws_redo.p, 64, hipMemcpyDeviceToHost, st));
            HAC_HIP(hipEventRecord(e->redo_ev, st));
            e->redo_pending = true;
            e->redo_items = (long)B * NH;
        }
    }
    e->plan_sub_batches += 1;
    e->plan_rows += Mp;
    e->plan_gemm = g8 ? "gemm8" : (big ? "classic256" : "classic128");
    return HAC_OK;
}

// ---- small batches: capture once, replay (see hac_encoder::graph_mode)
constexpr long GRAPH_MAX_ROWS = 16384;   // beyond this a forward is milliseconds of kernels: launches no longer matter
uint64_t ws_signature(const hac_encoder *e) {
    uint64_t h = 1469598103934665603ull;
    for (const GrowBuf *b : {&e->ws_x, &e->ws_xb, &e->ws_q, &e->ws_k, &e->ws_vt, &e->ws_ctx, &e->ws_y, &e->ws_h, &e->ws_seq, &e->ws_cls, &e->ws_stats,
                             &e->ws_yb, &e->ws_part, &e->ws_idstats, &e->ws_gids, &e->ws_gmask, &e->ws_gout, &e->ws_ksplit, &e->ws_redo})
        h = (h ^ (uint64_t)(uintptr_t)b->p) * 1099511628211ull;
    return h;
}
// (an executable graph may still be running on the stream of the call that replayed it last: wait for that replay's event
// before destroying anything -- HIP does not document destroying an in-flight exec as safe)
void graphs_quiesce(hac_encoder *e) {
    if (e->graph_done_armed) (void)hipEventSynchronize(e->graph_done);
    e->graph_done_armed = false;
}
void drop_graphs(hac_encoder *e) {
    if (e->graphs.empty()) return;
    graphs_quiesce(e);
    for (auto &kv : e->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    e->graphs.clear();
}
constexpr size_t GRAPH_MAX_SHAPES = 64;   // captured (B, L, options) shapes kept per encoder
bool graph_eligible(const hac_encoder *e, int B, int L, hipStream_t st) {
    if (e->graph_mode == 0 || e->prof_mask != 0) return false;
    const int L32 = (L + SEQ_ALIGN - 1) / SEQ_ALIGN * SEQ_ALIGN;
    if ((long)B * L32 > GRAPH_MAX_ROWS || (long)B * L32 > e->max_tokens) return false;   // (beyond max_tokens the call is cut into sub-batches)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;   // the caller captures: plain launches
    return true;
}
template <typename IT>
int forward_graph(hac_encoder *e, const IT *ids, const IT *mask, int B, int L, float *out_dev, hipStream_t st) {
    const size_t n_in = (size_t)B * L * sizeof(IT), n_out = (size_t)B * H * 4;
    HAC_TRY(e->ws_gids.reserve(n_in));
    HAC_TRY(e->ws_gmask.reserve(n_in));
    HAC_TRY(e->ws_gout.reserve(n_out));
    const uint64_t key = ((uint64_t)B << 40) | ((uint64_t)L << 24) | ((uint64_t)sizeof(IT) << 16) | ((uint64_t)(e->attn_mode & 1) << 8) |
                         ((uint64_t)((e->gemm_mode + 1) & 3) << 4) | (uint64_t)(e->g8_split & 15) | ((uint64_t)(e->ksplit_mode & 1) << 12) |
                         ((uint64_t)(e->attn_qsplit & 1) << 13) | ((uint64_t)(e->g8_stagger & 1) << 14) | ((uint64_t)((e->attn_pipe + 1) & 3) << 15);
    // (a caller that pads every batch to its own longest sequence can show hundreds of shapes: the cache is bounded, and starting
    // over costs each live shape one plain forward and one capture)
    if (e->graphs.size() >= GRAPH_MAX_SHAPES && e->graphs.find(key) == e->graphs.end()) drop_graphs(e);
    hac_encoder::GraphEntry &ge = e->graphs[key];
    const IT *gids = (const IT *)e->ws_gids.p, *gmask = (const IT *)e->ws_gmask.p;
    float *gout = (float *)e->ws_gout.p;
    HAC_HIP(hipMemcpyAsync(e->ws_gids.p, ids, n_in, hipMemcpyDeviceToDevice, st));
    HAC_HIP(hipMemcpyAsync(e->ws_gmask.p, mask, n_in, hipMemcpyDeviceToDevice, st));
    e->plan_sub_batches = 1;
    if (ge.seen == 0) {
        // first call of this shape: plain launches through the private buffers -- sizes every workspace (no allocation may
        // happen inside a capture) and is a valid forward by itself
        e->plan_rows = 0;
        HAC_TRY(run_forward<IT>(e, gids, gmask, B, L, gout, st));
        ge.seen = 1;
        ge.gemm = e->plan_gemm;
        ge.rows = e->plan_rows;
        ge.ks_out = e->plan_ks_out;
        ge.ks_down = e->plan_ks_down;
        ge.attn_pipe = e->plan_attn_pipe;
        e->plan_sub_batches = 1;
        e->plan_graph = "eager-first";
    } else {
        if (!ge.exec || ge.sig != ws_signature(e)) {
            // (a larger forward in between may have regrown a workspace: the captured launches hold the old pointers)
            if (ge.exec) graphs_quiesce(e);
            if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
            if (ge.graph) (void)hipGraphDestroy(ge.graph);
            ge.exec = nullptr;
            ge.graph = nullptr;
            HAC_HIP(hipStreamSynchronize(e->stream));
            HAC_HIP(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
            e->plan_rows = 0;
            const int rc = run_forward<IT>(e, gids, gmask, B, L, gout, e->stream);
            const hipError_t ce = hipStreamEndCapture(e->stream, &ge.graph);
            e->plan_sub_batches = 1;
            if (rc != HAC_OK) return rc;
            if (ce != hipSuccess || !ge.graph) return fail(HAC_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
            HAC_HIP(hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
            ge.sig = ws_signature(e);
            ge.gemm = e->plan_gemm;
            ge.rows = e->plan_rows;
            ge.ks_out = e->plan_ks_out;
            ge.ks_down = e->plan_ks_down;
            ge.attn_pipe = e->plan_attn_pipe;
        ge.attn_pipe = e->plan_attn_pipe;
        }
        HAC_HIP(hipGraphLaunch(ge.exec, st));
        if (!e->graph_done) HAC_HIP(hipEventCreateWithFlags(&e->graph_done, hipEventDisableTiming));
        HAC_HIP(hipEventRecord(e->graph_done, st));
        e->graph_done_armed = true;
        e->plan_gemm = ge.gemm;
        e->plan_rows = ge.rows;
        e->plan_ks_out = ge.ks_out;
        e->plan_ks_down = ge.ks_down;
        e->plan_attn_pipe = ge.attn_pipe;
        e->plan_graph = "replay";
    }
    HAC_HIP(hipMemcpyAsync(out_dev, gout, n_out, hipMemcpyDeviceToDevice, st));
    return HAC_OK;
}

template <typename IT>
int forward_batched(hac_encoder *e, const IT *ids, const IT *mask, int B, int L, float *out_dev, hipStream_t st) {
    const int L32 = (L + SEQ_ALIGN - 1) / SEQ_ALIGN * SEQ_ALIGN;
    e->plan_sub_batches = 0;
    e->plan_rows = 0;
    e->plan_graph = "off";
    if (graph_eligible(e, B, L, st)) return forward_graph<IT>(e, ids, mask, B, L, out_dev, st);
    if ((long)B * L32 <= e->max_tokens) return run_forward<IT>(e, ids, mask, B, L, out_dev, st);
    // More rows than one pass holds if every sequence were full length: size the sub-batches by the REAL padded
    // lengths (one seq_prep over the whole batch and a B-int read-back), so that varlen batches fill the
    // max_tokens-row GEMMs instead of running them half empty.
    SeqInfo s;
    HAC_TRY(seq_layout(e, B, L, s));
    seq_prep_kernel<IT><<<dim3(B), dim3(512), 0, st>>>(ids, mask, L, s, e->cfg.pad_token_id, e->cfg.vocab);
    HAC_HIP(hipGetLastError());
    // the one host read-back of a large forward: B ints through pinned memory, on the caller's stream
    if (e->h_len_cap < (size_t)B) {
        if (e->h_len) (void)hipHostFree(e->h_len);
        e->h_len = nullptr;
        e->h_len_cap = 0;
        hipError_t err = hipHostMalloc((void **)&e->h_len, (size_t)B * 4 * 2, hipHostMallocDefault);
        if (err != hipSuccess) return fail(HAC_ERR_OOM, "hipHostMalloc failed: %s", hipGetErrorString(err));
        e->h_len_cap = (size_t)B * 2;
    }
    HAC_HIP(hipMemcpyAsync(e->h_len, s.len32, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    HAC_HIP(hipStreamSynchronize(st));
    const int *len32 = e->h_len;
    // one GEMM family for the whole call: by the rows of the whole batch (at least one sub-batch is max_tokens rows long)
    long rows_all = 0;
    for (int b = 0; b < B; ++b) rows_all += std::min<long>(std::max(len32[b], SEQ_ALIGN), L32);
    const long rows_first = std::min<long>(rows_all, e->max_tokens);
    const int family = pick_family(e, rows_first);   // GEMM family AND classic tile size: one choice for every sub-batch of the call
    for (int b0 = 0; b0 < B;) {
        long rows = 0;
        int nb = 0;
        while (b0 + nb < B) {
            const long r = std::min<long>(std::max(len32[(size_t)b0 + nb], SEQ_ALIGN), L32);   // invalid masks are reported by the sub-batch itself
            if (nb > 0 && rows + r > e->max_tokens) break;
            rows += r;
            ++nb;
        }
        HAC_TRY(run_forward<IT>(e, ids + (size_t)b0 * L, mask + (size_t)b0 * L, nb, L, out_dev + (size_t)b0 * H, st, rows, family, rows_first));
        b0 += nb;
    }
    return HAC_OK;
}

}  // namespace

extern "C" {

int hac_encoder_create(const hac_encoder_config *cfg, int device, hac_encoder **out) {
    if (!out || !cfg) return fail(HAC_ERR_INVALID, "hac_encoder_create: null argument");
    *out = nullptr;
    if (cfg->hidden != H || cfg->n_heads != NH || cfg->ffn != FF)
        return fail(HAC_ERR_UNSUPPORTED, "only the RoBERTa-base geometry of ANCE is built (hidden 768, 12 heads, ffn 3072); got %d/%d/%d", cfg->hidden,
                    cfg->n_heads, cfg->ffn);
    if (cfg->n_layers < 1 || cfg->n_layers > 48 || cfg->vocab < 1 || cfg->max_pos < 3 || cfg->type_vocab < 1)
        return fail(HAC_ERR_INVALID, "bad encoder config");
    int n_visible = 0;
    if (hipGetDeviceCount(&n_visible) != hipSuccess || n_visible <= 0)
        return fail(HAC_ERR_HIP, "no HIP device visible: the haconvdr encoder has no CPU fallback");
    if (device < 0 || device >= n_visible) return fail(HAC_ERR_INVALID, "device id %d out of range (visible: %d)", device, n_visible);
    hac_encoder *e = new hac_encoder();
    e->cfg = *cfg;
    e->device = device;
    DeviceGuard g(device);
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
        delete e;
        return fail(HAC_ERR_HIP, "hipStreamCreate failed");
    }
    (void)hipFuncSetAttribute((const void *)attention_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void *)attention_kernel<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void *)attention_stream_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)hipFuncSetAttribute((const void *)attention_stream_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void *)attention_pipe_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)hipFuncSetAttribute((const void *)attention_pipe_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void *)gemm_bf16_nt_kernel<EPI_QKV, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)hipFuncSetAttribute((const void *)gemm_bf16_nt_kernel<EPI_RESID, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)hipFuncSetAttribute((const void *)gemm_bf16_nt_kernel<EPI_GELU, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)hipFuncSetAttribute((const void *)gemm_bf16_nt_kernel<EPI_QKV, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void *)gemm_bf16_nt_kernel<EPI_RESID, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void *)gemm_bf16_nt_kernel<EPI_GELU, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void *)gemm8_kernel<EPI8_QKV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void *)gemm8_kernel<EPI8_QKV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void *)gemm8_kernel<EPI8_RESID, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void *)gemm8_kernel<EPI8_RESID, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void *)gemm8_kernel<EPI8_GELU, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void *)gemm8_kernel<EPI8_GELU, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    if (const char *m = getenv("HAC_ENC_GEMM")) {   // classic | 8phase | auto
        const std::string v(m);
        if (v != "auto" && v != "classic" && v != "8phase") {
            delete e;
            return fail(HAC_ERR_INVALID, "HAC_ENC_GEMM = '%s': auto | classic | 8phase", m);
        }
        e->gemm_mode = v == "classic" ? 0 : (v == "8phase" ? 1 : -1);
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->n_cu = prop.multiProcessorCount;
    }
    *out = e;
    return HAC_OK;
}

void hac_encoder_destroy(hac_encoder *e) {
    if (!e) return;
    DeviceGuard g(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto &kv : e->raw) (void)hipFree(kv.second);
    for (auto &l : e->layers)
        for (bf16 *p : {l.wqkv, l.wo, l.w1, l.w2})
            if (p) (void)hipFree(p);
    for (auto &l : e->layers) {
        if (l.bqkv) (void)hipFree(l.bqkv);
        if (l.wqkv8) (void)hipFree(l.wqkv8);
        if (l.w18) (void)hipFree(l.w18);
        if (l.fold) (void)hipFree(l.fold);
    }
    drop_graphs(e);
    if (e->graph_done) (void)hipEventDestroy(e->graph_done);
    if (e->clk_ev) (void)hipEventDestroy(e->clk_ev);
    if (e->redo_ev) {
        if (e->redo_pending) (void)hipEventSynchronize(e->redo_ev);     // (the pinned words are the target of a copy that may still be in flight)
        (void)hipEventDestroy(e->redo_ev);
    }
    if (e->h_redo) (void)hipHostFree(e->h_redo);
    for (GrowBuf *b : {&e->ws_x, &e->ws_xb, &e->ws_q, &e->ws_k, &e->ws_vt, &e->ws_ctx, &e->ws_y, &e->ws_h, &e->ws_seq, &e->ws_ids, &e->ws_mask, &e->ws_out, &e->ws_cls, &e->ws_stats, &e->ws_yb, &e->ws_part, &e->ws_idstats,
                       &e->ws_gids, &e->ws_gmask, &e->ws_gout, &e->ws_ksplit, &e->ws_identgb, &e->ws_clk, &e->ws_redo})
        b->release();
    if (e->h_pin) (void)hipHostFree(e->h_pin);
    if (e->h_len) (void)hipHostFree(e->h_len);
    for (auto &pl : e->pools)
        for (auto &ev : pl.ev) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int hac_encoder_set_weight(hac_encoder *e, const char *name, const float *data, size_t count) {
    if (!e || !name || !data || count == 0) return fail(HAC_ERR_INVALID, "set_weight: bad arguments");
    DeviceGuard g(e->device);
    std::string key(name);
    auto it = e->raw.find(key);
    if (it != e->raw.end()) {
        (void)hipFree(it->second);
        e->raw.erase(it);
    }
    float *d = nullptr;
    HAC_HIP(hipMalloc((void **)&d, count * 4));
    HAC_HIP(hipMemcpy(d, data, count * 4, hipMemcpyHostToDevice));
    e->raw[key] = d;
    e->raw_count[key] = count;
    e->finalized = false;
    return HAC_OK;
}

int hac_encoder_finalize(hac_encoder *e) {
    if (!e) return fail(HAC_ERR_INVALID, "null encoder");
    DeviceGuard g(e->device);
    drop_graphs(e);   // captured launches hold the old weights' pointers
    const hac_encoder_config &c = e->cfg;
    const std::string p = "roberta.embeddings.";
    HAC_TRY(get_raw(e, p + "word_embeddings.weight", (size_t)c.vocab * H, &e->word));
    HAC_TRY(get_raw(e, p + "position_embeddings.weight", (size_t)c.max_pos * H, &e->posw));
    HAC_TRY(get_raw(e, p + "token_type_embeddings.weight", (size_t)c.type_vocab * H, &e->typew));
    HAC_TRY(get_raw(e, p + "LayerNorm.weight", H, &e->embg));
    HAC_TRY(get_raw(e, p + "LayerNorm.bias", H, &e->embb));
    HAC_TRY(get_raw(e, "embeddingHead.weight", (size_t)H * H, &e->wh));
    HAC_TRY(get_raw(e, "embeddingHead.bias", H, &e->bh));
    HAC_TRY(get_raw(e, "norm.weight", H, &e->ng));
    HAC_TRY(get_raw(e, "norm.bias", H, &e->nb));
    for (auto &l : e->layers) {
        for (bf16 *pp : {l.wqkv, l.wo, l.w1, l.w2, l.wqkv8, l.w18})
            if (pp) (void)hipFree(pp);
        if (l.bqkv) (void)hipFree(l.bqkv);
        if (l.fold) (void)hipFree(l.fold);
    }
    e->layers.assign(c.n_layers, LayerW());
    for (int i = 0; i < c.n_layers; ++i) {
        const std::string q = "roberta.encoder.layer." + std::to_string(i) + ".";
        LayerW &l = e->layers[i];
        float *wq, *wk, *wv, *bq, *bk, *bv, *t;
        HAC_TRY(get_raw(e, q + "attention.self.query.weight", (size_t)H * H, &wq));
        HAC_TRY(get_raw(e, q + "attention.self.key.weight", (size_t)H * H, &wk));
        HAC_TRY(get_raw(e, q + "attention.self.value.weight", (size_t)H * H, &wv));
        HAC_TRY(get_raw(e, q + "attention.self.query.bias", H, &bq));
        HAC_TRY(get_raw(e, q + "attention.self.key.bias", H, &bk));
        HAC_TRY(get_raw(e, q + "attention.self.value.bias", H, &bv));
        HAC_HIP(hipMalloc((void **)&l.wqkv, (size_t)3 * H * H * sizeof(bf16)));
        HAC_HIP(hipMalloc((void **)&l.bqkv, (size_t)3 * H * 4));
        const float *ws[3] = {wq, wk, wv};
        const float *bs[3] = {bq, bk, bv};
        for (int j = 0; j < 3; ++j) {
            f32_to_bf16_kernel<<<dim3((unsigned)(((size_t)H * H + 255) / 256)), dim3(256), 0, e->stream>>>(ws[j], l.wqkv + (size_t)j * H * H, (size_t)H * H);
            HAC_HIP(hipMemcpyAsync(l.bqkv + j * H, bs[j], H * 4, hipMemcpyDeviceToDevice, e->stream));
        }
        HAC_TRY(get_raw(e, q + "attention.output.dense.weight", (size_t)H * H, &t));
        HAC_TRY(to_bf16(e, t, (size_t)H * H, &l.wo));
        HAC_TRY(get_raw(e, q + "attention.output.dense.bias", H, &l.bo));
        HAC_TRY(get_raw(e, q + "attention.output.LayerNorm.weight", H, &l.ln1g));
        HAC_TRY(get_raw(e, q + "attention.output.LayerNorm.bias", H, &l.ln1b));
        HAC_TRY(get_raw(e, q + "intermediate.dense.weight", (size_t)FF * H, &t));
        HAC_TRY(to_bf16(e, t, (size_t)FF * H, &l.w1));
        HAC_TRY(get_raw(e, q + "intermediate.dense.bias", FF, &l.b1));
        HAC_TRY(get_raw(e, q + "output.dense.weight", (size_t)H * FF, &t));
        HAC_TRY(to_bf16(e, t, (size_t)H * FF, &l.w2));
        HAC_TRY(get_raw(e, q + "output.dense.bias", H, &l.b2));
        HAC_TRY(get_raw(e, q + "output.LayerNorm.weight", H, &l.ln2g));
        HAC_TRY(get_raw(e, q + "output.LayerNorm.bias", H, &l.ln2b));
        // large-batch path: fold the LayerNorm in front of QKV (the previous layer's output LayerNorm; layer 0 reads the
        // already normalized embedding rows) and the one in front of FFN-up (this layer's attention LayerNorm)
        HAC_HIP(hipMalloc((void **)&l.wqkv8, (size_t)3 * H * H * sizeof(bf16)));
        HAC_HIP(hipMalloc((void **)&l.w18, (size_t)FF * H * sizeof(bf16)));
        HAC_HIP(hipMalloc((void **)&l.fold, (size_t)(2 * 3 * H + 2 * FF) * 4));
        const float *pg = i ? e->layers[i - 1].ln2g : nullptr, *pb = i ? e->layers[i - 1].ln2b : nullptr;
        const float qscale = 0.125f * 1.44269504088896341f;   // softmax scale / ln 2: the attention kernel works in base 2
        for (int j = 0; j < 3; ++j)
            fold_ln_kernel<<<dim3(H), dim3(256), 0, e->stream>>>(ws[j], bs[j], pg, pb, H, j == 0 ? qscale : 1.0f, l.wqkv8 + (size_t)j * H * H,
                                                                 l.fold + j * H, l.fold + 3 * H + j * H);
        float *w1f;
        HAC_TRY(get_raw(e, q + "intermediate.dense.weight", (size_t)FF * H, &w1f));
        fold_ln_kernel<<<dim3(FF), dim3(256), 0, e->stream>>>(w1f, l.b1, l.ln1g, l.ln1b, H, 1.0f, l.w18, l.fold + 6 * H, l.fold + 6 * H + FF);
        HAC_HIP(hipGetLastError());
    }
    HAC_HIP(hipStreamSynchronize(e->stream));
    e->finalized = true;
    return HAC_OK;
}

int hac_encoder_forward_device(hac_encoder *e, const void *ids_dev, const void *mask_dev, int elem_bytes, int B, int L,
                               float *out_dev, void *hip_stream) {
    if (!e) return fail(HAC_ERR_INVALID, "null encoder");
    if (!e->finalized) return fail(HAC_ERR_INVALID, "encoder weights not finalized (hac_encoder_finalize)");
    if (B < 0 || L < 1 || L > 512 || L + 2 > e->cfg.max_pos || (B > 0 && (!ids_dev || !mask_dev || !out_dev)))
        return fail(HAC_ERR_INVALID, "forward: bad arguments (B=%d, L=%d; L must be in [1, min(512, max_pos-2)])", B, L);
    if (elem_bytes != 4 && elem_bytes != 8) return fail(HAC_ERR_INVALID, "forward: ids/mask must be int32 or int64");
    if (B == 0) return HAC_OK;
    DeviceGuard g(e->device);
    hipStream_t st = (hipStream_t)hip_stream;
    if (elem_bytes == 8) return forward_batched<long long>(e, (const long long *)ids_dev, (const long long *)mask_dev, B, L, out_dev, st);
    return forward_batched<int>(e, (const int *)ids_dev, (const int *)mask_dev, B, L, out_dev, st);
}

int hac_encoder_forward(hac_encoder *e, const int32_t *ids, const int32_t *mask, int B, int L, float *out) {
    if (!e) return fail(HAC_ERR_INVALID, "null encoder");
    if (B < 0 || (B > 0 && (!ids || !mask || !out))) return fail(HAC_ERR_INVALID, "forward: bad arguments");
    if (B == 0) return HAC_OK;
    DeviceGuard g(e->device);
    const size_t n = (size_t)B * L;
    HAC_TRY(e->ws_ids.reserve(n * 4));
    HAC_TRY(e->ws_mask.reserve(n * 4));
    HAC_TRY(e->ws_out.reserve((size_t)B * H * 4));
    const size_t need = std::max(n * 8, (size_t)B * H * 4);
    if (e->h_pin_bytes < need) {
        if (e->h_pin) (void)hipHostFree(e->h_pin);
        e->h_pin = nullptr;
        e->h_pin_bytes = 0;
        hipError_t err = hipHostMalloc(&e->h_pin, need + need / 4, hipHostMallocDefault);
        if (err != hipSuccess) return fail(HAC_ERR_OOM, "hipHostMalloc failed: %s", hipGetErrorString(err));
        e->h_pin_bytes = need + need / 4;
    }
    HAC_HIP(hipStreamSynchronize(e->stream));
    std::memcpy(e->h_pin, ids, n * 4);
    std::memcpy((char *)e->h_pin + n * 4, mask, n * 4);
    HAC_HIP(hipMemcpyAsync(e->ws_ids.p, e->h_pin, n * 4, hipMemcpyHostToDevice, e->stream));
    HAC_HIP(hipMemcpyAsync(e->ws_mask.p, (char *)e->h_pin + n * 4, n * 4, hipMemcpyHostToDevice, e->stream));
    HAC_TRY(hac_encoder_forward_device(e, e->ws_ids.p, e->ws_mask.p, 4, B, L, (float *)e->ws_out.p, e->stream));
    HAC_HIP(hipStreamSynchronize(e->stream));  // the pinned buffer is reused for the result
    HAC_HIP(hipMemcpyAsync(e->h_pin, e->ws_out.p, (size_t)B * H * 4, hipMemcpyDeviceToHost, e->stream));
    HAC_HIP(hipStreamSynchronize(e->stream));
    std::memcpy(out, e->h_pin, (size_t)B * H * 4);
    for (size_t i = 0; i < (size_t)B * H; ++i)
        if (out[i] != out[i])
            return fail(HAC_ERR_INVALID, "forward: sequence %zu: attention_mask must be a non-empty prefix mask (first len ones) and attended token ids "
                                         "must lie in [0, %d)", i / H, e->cfg.vocab);
    return HAC_OK;
}

int hac_encoder_set_option(hac_encoder *e, const char *name, const char *value) {
    if (!e || !name || !value) return fail(HAC_ERR_INVALID, "set_option: null argument");
    const std::string n(name), v(value);
    // a value outside the documented set is an error, never a silent default (a mistyped value in a cross-check test would
    // otherwise exercise the wrong kernels and still pass)
    if (n == "gemm") {
        if (v != "auto" && v != "classic" && v != "8phase") return fail(HAC_ERR_INVALID, "encoder option gemm = '%s': auto | classic | 8phase", value);
        e->gemm_mode = v == "classic" ? 0 : (v == "8phase" ? 1 : -1);
    } else if (n == "attn") {
        if (v != "stream" && v != "twopass") return fail(HAC_ERR_INVALID, "encoder option attn = '%s': stream | twopass", value);
        e->attn_mode = v == "twopass" ? 1 : 0;
    } else if (n == "attn_pipe") {
        if (v != "auto" && v != "off" && v != "all") return fail(HAC_ERR_INVALID, "encoder option attn_pipe = '%s': auto | off | all", value);
        e->attn_pipe = v == "off" ? 0 : (v == "all" ? 1 : -1);
    } else if (n == "g8_split") {
        char *end = nullptr;
        const long t = strtol(value, &end, 10);
        if (end == value || *end || t < 0 || t > 15) return fail(HAC_ERR_INVALID, "encoder option g8_split = '%s': a bit mask 0..15", value);
        e->g8_split = (int)t;
    } else if (n == "attn_qsplit") {
        if (v != "auto" && v != "off") return fail(HAC_ERR_INVALID, "encoder option attn_qsplit = '%s': auto | off", value);
        e->attn_qsplit = v == "off" ? 0 : -1;
    } else if (n == "g8_stagger") {
        if (v != "auto" && v != "off") return fail(HAC_ERR_INVALID, "encoder option g8_stagger = '%s': auto | off", value);
        e->g8_stagger = v == "off" ? 0 : -1;
    } else if (n == "attn_qs_pin") {
        if (v != "0" && v != "1" && v != "2" && v != "4" && v != "8" && v != "16") return fail(HAC_ERR_INVALID, "encoder option attn_qs_pin = '%s': 0 | 1 | 2 | 4 | 8 | 16", value);
        e->attn_qs_pin = atoi(v.c_str());
        drop_graphs(e);
    } else if (n == "ksplit_pin") {
        int a = 0, b = 0;
        if (sscanf(v.c_str(), "%d/%d", &a, &b) != 2 || a < 0 || b < 0 || a > 16 || b > 16) return fail(HAC_ERR_INVALID, "encoder option ksplit_pin = '%s': a/b with 0 <= a, b <= 16", value);
        e->ks_pin_out = a;
        e->ks_pin_down = b;
        drop_graphs(e);
    } else if (n == "ksplit") {
        if (v != "auto" && v != "off") return fail(HAC_ERR_INVALID, "encoder option ksplit = '%s': auto | off", value);
        e->ksplit_mode = v == "off" ? 0 : -1;
    } else if (n == "graph") {
        if (v != "auto" && v != "off") return fail(HAC_ERR_INVALID, "encoder option graph = '%s': auto | off", value);
        e->graph_mode = v == "off" ? 0 : -1;
    } else if (n == "max_tokens") {
        char *end = nullptr;
        const long t = strtol(value, &end, 10);
        if (end == value || *end || t < 4096) return fail(HAC_ERR_INVALID, "encoder option max_tokens = '%s': an integer >= 4096", value);
        e->max_tokens = t;
    } else {
        return fail(HAC_ERR_INVALID, "unknown encoder option '%s'", name);
    }
    return HAC_OK;
}

const char *hac_encoder_last_plan(hac_encoder *e) {
    if (!e) return "none";
    snprintf(e->last_plan, sizeof e->last_plan, "gemm=%s attn=%s sub_batches=%d rows=%ld graph=%s ksplit=%d/%d attn_form=%s", e->plan_gemm, e->attn_mode ? "twopass" : "stream",
             e->plan_sub_batches, e->plan_rows, e->plan_graph, e->plan_ks_out, e->plan_ks_down, e->attn_mode ? "twopass" : (e->plan_attn_pipe ? "woven" : "single"));
    return e->last_plan;
}

int hac_encoder_set_profiling(hac_encoder *e, int mask) {
    if (!e) return fail(HAC_ERR_INVALID, "null encoder");
    e->prof_mask = (unsigned)mask & ((2u << HAC_ENC_NCLASS) - 1u);
    return HAC_OK;
}

static int drain_pool(hac_encoder *e, int pool, float *ms_out, int cap, int *n_out) {
    auto &pl = e->pools[pool];
    int n = 0;
    for (size_t i = 0; i < pl.used && n < cap; ++i, ++n) {
        HAC_HIP(hipEventSynchronize(pl.ev[i].second));
        HAC_HIP(hipEventElapsedTime(&ms_out[n], pl.ev[i].first, pl.ev[i].second));
    }
    pl.used = 0;
    *n_out = n;
    return HAC_OK;
}

int hac_encoder_profile_drain(hac_encoder *e, float *ms_out, int cap, int *n_out) {
    if (!e || !n_out || (cap > 0 && !ms_out)) return fail(HAC_ERR_INVALID, "bad arguments");
    DeviceGuard g(e->device);
    return drain_pool(e, 0, ms_out, cap, n_out);
}

int hac_encoder_last_clock(hac_encoder *e, uint64_t out[2]) {
    if (!e || !out) return fail(HAC_ERR_INVALID, "hac_encoder_last_clock: null argument");
    out[0] = out[1] = 0;
    if (!e->clk_valid || !e->ws_clk.p) return HAC_OK;
    DeviceGuard g(e->device);
    unsigned long long h[4] = {0, 0, 0, 0};
    // (the event behind that launch, not the launch's stream -- the caller may have destroyed it since -- and not the whole device:
    // ADVICE r5)
    if (e->clk_ev) HAC_HIP(hipEventSynchronize(e->clk_ev));
    else HAC_HIP(hipDeviceSynchronize());
    HAC_HIP(hipMemcpy(h, e->ws_clk.p, sizeof h, hipMemcpyDeviceToHost));
    if (h[2] > h[0] && h[3] > h[1]) {
        out[0] = h[2] - h[0];
        out[1] = h[3] - h[1];
    }
    return HAC_OK;
}

int hac_encoder_attention_redo(hac_encoder *e, long long *items_out) {
    if (!e || !items_out) return fail(HAC_ERR_INVALID, "hac_encoder_attention_redo: null argument");
    *items_out = 0;
    if (!e->ws_redo.p) return HAC_OK;
    DeviceGuard g(e->device);
    int h[16] = {0};
    HAC_HIP(hipDeviceSynchronize());
    HAC_HIP(hipMemcpy(h, e->ws_redo.p, sizeof h, hipMemcpyDeviceToHost));
    // (a wave that finds a row outside the window counts its item once; several waves of one item may: an upper bound of the items, 0 iff none)
    for (int i = 0; i < 16; ++i) *items_out += h[i];
    return HAC_OK;
}

int hac_encoder_profile_drain_class(hac_encoder *e, int cls, float *ms_out, int cap, int *n_out) {
    if (!e || !n_out || (cap > 0 && !ms_out) || cls < 0 || cls >= HAC_ENC_NCLASS) return fail(HAC_ERR_INVALID, "bad arguments");
    DeviceGuard g(e->device);
    return drain_pool(e, 1 + cls, ms_out, cap, n_out);
}

}  // extern "C"
