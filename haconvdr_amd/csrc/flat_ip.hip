// Exact inner-product top-k search for gfx950 (MI355X): the faiss.IndexFlatIP
// replacement behind hac_index_* (include/haconvdr.h).
//
// Reference path replaced: src/test_HAConvDR_topiocqa.py:52 (IndexFlatIP(768)),
// :98 add, :102 search, :122 reset, :110 id remap, :126-149 block merge.
//
// Design:
//   * HBM layout "T64": the corpus is re-tiled at add() into groups of 64 rows;
//     inside a group chunk t (t = 0..d/4-1) is 1 KiB holding, for lane l = row l of
//     the group, the four floats x[row][4t..4t+3].  One dwordx4 load per lane is a
//     fully coalesced 1-KiB wave access AND already the A operand (4 k-steps) of
//     v_mfma_f32_16x16x1_4b_f32 — no LDS staging, no shuffles for the corpus.
//   * Scores are exact fp32: the MFMA's per-element arithmetic is the k-ordered
//     fmaf chain (verified bit-for-bit on MI355X), identical to the CPU oracle.
//   * One workgroup (4 waves) keeps <=16 queries in LDS ([d/4][QT] float4, the B
//     operand, conflict-free ds_read_b128) and streams groups; each wave owns one
//     64-row group per round.  Output tile: lane -> query (lane&15), 16 regs -> rows.
//   * Top-k: scores pass a per-query threshold (a proven lower bound of the final
//     k-th score: seeded from an exact top-k of a corpus sample, raised by local
//     compactions and a chip-wide atomicMax), survivors are appended to per-query
//     LDS candidate buffers as 64-bit keys (orderable score << 32 | ~position) and
//     compacted by an in-LDS wave-level bitonic sort.  Per-workgroup top-k lists
//     are merged by a second kernel.  The result is independent of timing: the
//     total order on keys is strict and the threshold never exceeds the true k-th.
#include "hac_common.h"

#include <algorithm>
#include <type_traits>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace hac {

std::string &last_error_slot() {
    static thread_local std::string s;
    return s;
}
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

namespace {

constexpr int GROUP_ROWS = 64;
constexpr int SCAN_WAVES = 4;          // waves per scan workgroup
constexpr int PF = 8;                  // 1-KiB chunks in flight per wave
constexpr int MAX_SEG = 64;
constexpr int MERGE_THREADS = 256;
constexpr size_t LDS_LIMIT = 160 * 1024;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SegDesc {
    const float4 *ptr;  // tiled rows of this segment (T64, fp32)
    const uint4 *himg;  // the segment's fp16 image (H64, scan_split.inc); null until a search takes the prefilter path
    u32 gstart;         // first global group index of this segment
    u32 pad_;
};

// ------------------------------------------------------------------ key packing
__device__ __forceinline__ u32 f2ord(float f) {
    u32 b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(u32 o) {
    u32 b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(b);
}
__device__ __forceinline__ u64 make_key(float s, u32 pos) {
    s = s + 0.0f;  // -0.0 -> +0.0 so that key order == (score desc, pos asc) under float compare
    return ((u64)f2ord(s) << 32) | (u64)(0xFFFFFFFFu - pos);
}

// Pointers that reach a kernel through memory (segment table, ScanArgs) are generic to the
// compiler, which then emits flat_load + "vmcnt(0) lgkmcnt(0)" drains.  Loading through an
// explicit global (address_space(1)) pointer gives global_load and counted vmcnt waits.  The
// cast must happen where the pointer is formed: cast at the load site after a select of two
// generic pointers, hipcc (ROCm 7.2) falls back to flat_load.
// f4: plain 4-float vector (HIP's float4 class cannot be copied out of address_space(1)).
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f4 *gf4ptr;
typedef const float4 *gf4ptr_t;
__device__ __forceinline__ gf4ptr as_global(const float4 *p) { return (gf4ptr)(p); }

__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

// In-LDS bitonic sort, descending, by ONE wave.  buf has capacity >= next_pow2(n).
__device__ void wave_sort_desc(u64 *buf, u32 n, int lane) {
    u32 np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (u32 i = n + lane; i < np2; i += 64) buf[i] = 0;
    lds_fence();
    for (u32 size = 2; size <= np2; size <<= 1) {
        for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
            for (u32 t = lane; t < (np2 >> 1); t += 64) {
                u32 lo = 2 * t - (t & (stride - 1));
                u32 hi = lo + stride;
                bool desc = (lo & size) == 0;
                u64 a = buf[lo], b = buf[hi];
                if ((a < b) == desc) {
                    buf[lo] = b;
                    buf[hi] = a;
                }
            }
            lds_fence();
        }
    }
}

// Same, by a whole workgroup (blockDim.x threads); all threads must call it.
__device__ void block_sort_desc(u64 *buf, u32 n, int tid, int nthreads) {
    u32 np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (u32 i = n + tid; i < np2; i += nthreads) buf[i] = 0;
    __syncthreads();
    for (u32 size = 2; size <= np2; size <<= 1) {
        for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
            for (u32 t = tid; t < (np2 >> 1); t += nthreads) {
                u32 lo = 2 * t - (t & (stride - 1));
                u32 hi = lo + stride;
                bool desc = (lo & size) == 0;
                u64 a = buf[lo], b = buf[hi];
                if ((a < b) == desc) {
                    buf[lo] = b;
                    buf[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ re-tiling
// Row-major rows -> T64 tiles.  One workgroup per destination group touched by
// segment rows [row0, row0 + m).  Reads are row-contiguous (256 B per 16 lanes),
// writes are whole 1-KiB chunks; the transpose goes through a padded LDS tile.
//
// The same pass writes the half-precision image the many-query prefilter streams ("H64", fp16 = round-to-nearest-even
// of every element): a group is K4/4 steps of 2 KiB; step t holds, for row 32*hf + r and k = 16t + 8hh .. +8, eight
// fp16 at byte t*2048 + hf*1024 + hh*512 + r*16 -- one KiB per (step, row half) is one LDS-DMA instruction of
// scanh_kernel and, lane for lane, the A operand of v_mfma_f32_32x32x16_f16.  +50 % of HBM for the corpus; the exact
// kernels and the rescoring read only the fp32 tiles.
__global__ __launch_bounds__(256) void tile_rows_kernel(const float4 *__restrict__ src, long m, int K4,
                                                        float4 *__restrict__ seg, uint4 *__restrict__ hseg, long row0,
                                                        u32 *__restrict__ norm2_max_bits) {
    __shared__ float4 tile[16][65];
    const int tid = threadIdx.x;
    const long gd = row0 / GROUP_ROWS + blockIdx.x;
    const long r_lo = max(row0, gd * GROUP_ROWS), r_hi = min(row0 + m, gd * GROUP_ROWS + GROUP_ROWS);
    float4 *dst = seg + gd * (long)K4 * GROUP_ROWS;
    uint4 *hdst = hseg ? hseg + gd * (long)K4 * (GROUP_ROWS / 2) : nullptr;   // 16-byte pieces: K4 * 512 bytes per group
    // |x|^2 of the rows passing through (thread = 4 rows x one of 16 column lanes): the largest row norm of
    // the index is the scale of the half-precision prefilter's error bound (scan_split.inc)
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k4b = 0; k4b < K4; k4b += 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            const int r = i >> 4, c = i & 15;
            const long dr = gd * GROUP_ROWS + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dr >= r_lo && dr < r_hi && k4b + c < K4) v = src[(dr - row0) * K4 + k4b + c];
            tile[c][r] = v;
            ss[j] = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, ss[j]))));
        }
        __syncthreads();
        for (int i = tid; i < 1024; i += 256) {
            const int c = i >> 6, r = i & 63;
            const long dr = gd * GROUP_ROWS + r;
            if (dr >= r_lo && dr < r_hi && k4b + c < K4) dst[(long)(k4b + c) * GROUP_ROWS + r] = tile[c][r];
        }
        // fp16 image (once the index has one): the 16 chunks are 4 steps; piece pid = (step, hf, hh, r) holds chunks
        // 4*step + 2*hh, +1 of row 32*hf + r
        for (int pid = tid; pid < 512 && hdst; pid += 256) {
            const int tl = pid >> 7, hf = (pid >> 6) & 1, hh = (pid >> 5) & 1, r = 32 * hf + (pid & 31), c0 = 4 * tl + 2 * hh;
            const long dr = gd * GROUP_ROWS + r;
            if (dr >= r_lo && dr < r_hi && k4b + c0 < K4) {
                const float4 a = tile[c0][r], b = tile[c0 + 1][r];
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                h8 h;
                h[0] = (_Float16)a.x; h[1] = (_Float16)a.y; h[2] = (_Float16)a.z; h[3] = (_Float16)a.w;
                h[4] = (_Float16)b.x; h[5] = (_Float16)b.y; h[6] = (_Float16)b.z; h[7] = (_Float16)b.w;
                hdst[(long)((k4b >> 2) + tl) * 128 + (pid & 127)] = __builtin_bit_cast(uint4, h);
            }
        }
        __syncthreads();
    }
    u32 best = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = ss[j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);   // the 16 column lanes of a row are adjacent lanes
        const u32 b = (s != s) ? 0x7FC00000u : __float_as_uint(s);   // non-negative float bits order like the value; NaN on top
        best = max(best, b);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, (u32)__shfl_xor((int)best, o));
    if ((tid & 63) == 0 && best != 0u) atomicMax(norm2_max_bits, best);
}

// The fp16 image of groups [g0, g0 + gridDim.x) of a segment from its fp32 tiles: what tile_rows_kernel writes beside the
// tiles once the image exists.  It is built lazily, by the first search that takes the prefilter path: an index that only
// ever sees a few queries per call (the HBM-bound regime) or has split = "0" never pays the +50 % of HBM.
// T64 tiles of groups g0 + blockIdx.x -> the same rows row-major ([row][K4] float4): the rescoring's copy (see ensure_row_image)
__global__ __launch_bounds__(256) void untile_rows_kernel(const float4 *__restrict__ seg, float4 *__restrict__ rows, long g0, int K4) {
    const long g = g0 + blockIdx.x;
    const int r = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const float4 *src = seg + (size_t)g * K4 * GROUP_ROWS + r;
    float4 *dst = rows + ((size_t)g * GROUP_ROWS + r) * K4;
    for (int t = tq; t < K4; t += 4) dst[t] = src[(size_t)t * GROUP_ROWS];
}

__global__ __launch_bounds__(256) void half_image_kernel(const float4 *__restrict__ seg, uint4 *__restrict__ hseg, long g0, int K4) {
    const long gd = g0 + blockIdx.x;
    const gf4ptr_t src = reinterpret_cast<gf4ptr_t>(seg) + gd * (long)K4 * GROUP_ROWS;
    uint4 *hdst = hseg + gd * (long)K4 * (GROUP_ROWS / 2);
    const int n_pieces = (K4 >> 2) * 128;
    for (int p = threadIdx.x; p < n_pieces; p += 256) {
        const int st = p >> 7, hf = (p >> 6) & 1, hh = (p >> 5) & 1, r = 32 * hf + (p & 31), c0 = 4 * st + 2 * hh;
        const float4 a = src[(long)c0 * GROUP_ROWS + r], b = src[(long)(c0 + 1) * GROUP_ROWS + r];
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 h;
        h[0] = (_Float16)a.x; h[1] = (_Float16)a.y; h[2] = (_Float16)a.z; h[3] = (_Float16)a.w;
        h[4] = (_Float16)b.x; h[5] = (_Float16)b.y; h[6] = (_Float16)b.z; h[7] = (_Float16)b.w;
        hdst[p] = __builtin_bit_cast(uint4, h);
    }
}

// ------------------------------------------------------------------ scan kernel
struct ScanArgs {
    const SegDesc *segs;
    int nseg;
    const float4 *q;        // [nq][K4] row-major queries
    const float4 *qt;       // scanq only: queries re-tiled to [tile][K4][NQ] float4, zero padded
    int nq, K4, k, C, QT;   // C: candidate slots per query (pow2 >= k + 64*SCAN_WAVES); QT queries per workgroup
    long n_rows;            // valid rows in the index
    u32 g_first, g_step, n_items;  // work items i -> group g_first + i*g_step
    const float *thr_init;  // [nq] lower bounds of the final k-th score, or null
    u32 *thr_glob;          // [nq] chip-wide threshold, orderable-uint encoding, 0 = none
    u64 *partial;           // [nq][gridDim.x * k] survivors of all workgroups, densely appended per query
    u32 *partial_cnt;       // [nq] entries appended so far (zeroed before the launch)
    u32 pos_base;
    // Searches whose query count is only known on the device (the prefilter's fallback, decided without a host read-back):
    // nq is then the capacity the grid was sized for and *nq_dev the number of queries actually present (0 in the common
    // case: every workgroup exits at once).  Workgroups of dead query tiles join the live ones (vgrid).
    const int *nq_dev;
};
// Hang-proofing.  The candidate loops of scanq_kernel / scanh_kernel count the passes of a round that ended in an overflow
// (in LDS, on the overflow route only) and give up at a bound no legal input reaches.  Nothing of this may cost the scan
// kernels a register (scanh_kernel<1> sits at 256 VGPRs and spills SGPRs into them: one more kernel argument, or gridDim,
// is a live SGPR pair from the first instruction on), so everything hangs off the chip-wide threshold array:
//   thr_glob[-4]     debug pass bound (tests), 0 = the kernel's own           (the words in front are zeroed with the array)
//   thr_glob[q]      = THR_POISON for the queries of a tile whose workgroup gave up.  As a threshold the value is inert (it
//                    decodes to a NaN, which fmaxf ignores, and no score encodes to it: NaN scores never pass a threshold);
//                    the survivors' flush keeps nothing for such a query and select_keys_kernel, which runs behind every
//                    scan, returns an EMPTY list for it and sets the index's sticky error word (hac_index_last_status).
constexpr u32 DEV_ERR_PASS_OVERRUN = 1u;
constexpr u32 THR_POISON = 0xFFFFFFFFu;
constexpr int THR_CTL_WORDS = 4;
__device__ __forceinline__ u32 scan_pass_bound(const ScanArgs &a, u32 own) {
    const u32 dbg = __hip_atomic_load(a.thr_glob - THR_CTL_WORDS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return dbg ? dbg : own;
}
// (workgroup-uniform call; the tile has at most as many queries as the workgroup has threads)
__device__ __forceinline__ void scan_overrun(const ScanArgs &a, int q0, int nq_tile, int tid) {
    asm volatile("" : "+v"(tid));   // the address is formed HERE: hoisted to the kernel's start it is a spilled register pair
    if (tid < nq_tile) atomicMax(&a.thr_glob[q0 + tid], THR_POISON);
}

struct VGrid {
    u32 bx, by, nx;   // row stream, query tile, row streams per query tile: what blockIdx.x, blockIdx.y, gridDim.x are without nq_dev
    int nq;
};
__device__ __forceinline__ bool vgrid(const ScanArgs &a, int QT, VGrid &v) {
    v.bx = blockIdx.x;
    v.by = blockIdx.y;
    v.nx = gridDim.x;
    v.nq = a.nq;
    if (!a.nq_dev) return true;
    const int nq = min(a.nq, *a.nq_dev);
    if (nq <= 0) return false;
    const u32 T = (u32)(nq + QT - 1) / (u32)QT, W = gridDim.x * gridDim.y, w = blockIdx.y * gridDim.x + blockIdx.x, S = W / T;
    if (w >= S * T) return false;
    v.bx = w / T;
    v.by = w % T;
    v.nx = S;
    v.nq = nq;
    return true;
}

__device__ __forceinline__ gf4ptr group_ptr(const ScanArgs &a, u32 g) {
    int s = 0;
    while (s + 1 < a.nseg && g >= a.segs[s + 1].gstart) ++s;
    return as_global(a.segs[s].ptr) + (size_t)(g - a.segs[s].gstart) * a.K4 * GROUP_ROWS;
}

// End of a workgroup's scan: hand its (at most k) survivors of one query to the merge.  They are
// appended densely to the query's global list (one atomicAdd reserves the range), unsorted unless the
// workgroup holds more than k: the merge kernel filters and sorts anyway, and reads only what was
// appended instead of gridDim.x fixed k-slot lists that are mostly padding.
__device__ __forceinline__ void flush_survivors(const ScanArgs &a, u32 n_streams, int q, u64 *b, u32 n, int lane) {
    if (n == 0) return;
    if (n > (u32)a.k) {
        wave_sort_desc(b, n, lane);
        n = (u32)a.k;
    }
    u32 base = 0;
    if (lane == 0) base = atomicAdd(&a.partial_cnt[q], n);
    base = __shfl(base, 0);
    u64 *out = a.partial + (size_t)q * n_streams * a.k + base;
    for (u32 i = lane; i < n; i += 64) out[i] = b[i];
}

#define HAC_MFMA4(av, bv)                                                        \
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32((av).x, (bv).x, acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32((av).y, (bv).y, acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32((av).z, (bv).z, acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32((av).w, (bv).w, acc, 0, 0, 0);

__global__ __launch_bounds__(SCAN_WAVES * 64) void scan16_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K4 = a.K4;
    VGrid vg;
    if (!vgrid(a, a.QT, vg)) return;
    const int q0 = vg.by * a.QT;
    const int QTr = min(a.QT, vg.nq - q0);
    const int C = a.C;

    f4 *ldsQ = reinterpret_cast<f4 *>(smem);                       // [K4][QTr]
    u64 *cand = reinterpret_cast<u64 *>(smem + (size_t)K4 * QTr * 16);     // [QTr][C]
    u32 *cnt = reinterpret_cast<u32 *>(cand + (size_t)QTr * C);            // [16]
    float *thr = reinterpret_cast<float *>(cnt + 16);                      // [16]

    for (int idx = tid; idx < K4 * QTr; idx += SCAN_WAVES * 64) {
        const int k4 = idx / QTr, j = idx - k4 * QTr;
        ldsQ[idx] = as_global(a.q)[(size_t)(q0 + j) * K4 + k4];
    }
    if (tid < 16) {
        cnt[tid] = 0;
        thr[tid] = (tid < QTr && a.thr_init) ? a.thr_init[q0 + tid] : -INFINITY;
    }
    __syncthreads();

    const int j = lane & 15;
    const int jc = min(j, QTr - 1);
    const f4 *qb = ldsQ + jc;
    const u32 stride = vg.nx * SCAN_WAVES;
    const u32 nrounds = (a.n_items + stride - 1) / stride;
    const u32 hw = (u32)C - 64u * SCAN_WAVES;  // compaction high-water mark (>= k)

    u32 item = vg.bx * SCAN_WAVES + w;
    bool have = item < a.n_items;
    gf4ptr gp = group_ptr(a, a.g_first + (have ? item : 0) * a.g_step) + lane;
    f4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = gp[i * 64];

    for (u32 r = 0; r < nrounds; ++r) {
        const u32 g = a.g_first + item * a.g_step;
        const u32 nitem = item + stride;
        const bool have_next = nitem < a.n_items;
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (have) {
            gf4ptr np = have_next ? group_ptr(a, a.g_first + nitem * a.g_step) + lane : gp;
            const int NB = K4 / PF;
            // clean entry state for the chunk loop (see scanq_kernel): counted vmcnt(PF-1) inside
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only, once per group
            // B (query) fragments run one chunk ahead of the MFMAs; the sched_barrier after every
            // chunk pins "consume ring[i] -> refill ring[i]" in program order, otherwise hipcc
            // clusters the eight refills at the end of the block and waits vmcnt(0) on them.
            f4 bcur = qb[0], bnxt;
            for (int tb = 0; tb < NB - 1; ++tb) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const int t = tb * PF + i;
                    bnxt = qb[(t + 1) * QTr];
                    const f4 av = ring[i];
                    HAC_MFMA4(av, bcur)
                    // refill AFTER the MFMAs that read ring[i]: the load can then reuse the register;
                    // issued before them it gets a fresh one and hipcc copies it back at the loop
                    // end behind vmcnt(7)...vmcnt(0) — a full drain every PF chunks
                    ring[i] = gp[(t + PF) * 64];
                    bcur = bnxt;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int t = (NB - 1) * PF + i;
                bnxt = qb[(i + 1 < PF ? t + 1 : 0) * QTr];
                const f4 av = ring[i];
                HAC_MFMA4(av, bcur)
                ring[i] = np[i * 64];  // next group's first chunks stay in flight across the epilogue
                bcur = bnxt;
                __builtin_amdgcn_sched_barrier(0);
            }
            gp = np;
        }

        __syncthreads();  // (A) last round's compactions are complete
        if (have) {
            const float th = thr[jc];
            bool anyp = false;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) anyp |= (acc[rr] >= th);
            if (__any(anyp)) {
                const long rem = a.n_rows - (long)g * GROUP_ROWS;
                const int rows_valid = rem < GROUP_ROWS ? (int)rem : GROUP_ROWS;
                const int rbase = 4 * (lane >> 4);
                if (j < QTr) {
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) {
                        const int row = 16 * (rr >> 2) + rbase + (rr & 3);
                        const float s = acc[rr];
                        if (s >= th && row < rows_valid) {
                            const u32 pos = atomicAdd(&cnt[j], 1u);
                            cand[(size_t)j * C + pos] = make_key(s, a.pos_base + g * GROUP_ROWS + row);
                        }
                    }
                }
            }
        }
        __syncthreads();  // (B) all appends of this round are visible
        for (int jj = w; jj < QTr; jj += SCAN_WAVES) {
            const u32 n = cnt[jj];
            float t_new = -INFINITY;
            if (n > hw) {  // wave-uniform
                wave_sort_desc(cand + (size_t)jj * C, n, lane);
                t_new = ord2f((u32)(cand[(size_t)jj * C + a.k - 1] >> 32));
            }
            if (lane == 0) {
                float t_cur = thr[jj];
                if (n > hw) {
                    cnt[jj] = a.k;
                    if (t_new > t_cur) {
                        t_cur = t_new;
                        atomicMax(&a.thr_glob[q0 + jj], f2ord(t_new));
                    }
                }
                const u32 go = __hip_atomic_load(&a.thr_glob[q0 + jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (go != 0u) t_cur = fmaxf(t_cur, ord2f(go));
                thr[jj] = t_cur;
            }
        }
        item = nitem;
        have = have_next;
    }

    __syncthreads();
    for (int jj = w; jj < QTr; jj += SCAN_WAVES) {
        const u32 n = cnt[jj];
        u64 *b = cand + (size_t)jj * C;
        flush_survivors(a, vg.nx, q0 + jj, b, n, lane);
    }
}

// ------------------------------------------------------------------ scan kernel, many queries
// GEMM-shaped variant for nq > 16: one workgroup of W waves scores NQ = 32*NT queries against
// W groups (64 rows each) per round.  A (corpus) still streams global -> VGPR in T64 chunks;
// B (queries) is re-tiled once into [tile][d/4][NQ] float4 and staged slice by slice (16 chunks =
// 64 k) through a double-buffered LDS image shared by all waves, so the corpus is re-read
// nq/NQ times instead of nq/16 times.  MFMA: v_mfma_f32_32x32x1_2b_f32 (lane l <-> row l of the
// group as A, query l&31 as B; bit-exact k-ordered fmaf chain).  Candidate buffers are only
// C = next_pow2(k+1) slots per query; a full buffer raises an overflow flag, the owners compact
// (sort, keep k, raise the threshold) and the waves re-offer what is still pending.
typedef float f32x32 __attribute__((ext_vector_type(32)));

__global__ void retile_queries_kernel(const float4 *__restrict__ q, int nq, int K4, int NQ, int n_tiles,
                                      float4 *__restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_tiles * K4 * NQ;
    if (idx >= total) return;
    const int j = (int)(idx % NQ);
    const long r = idx / NQ;
    const int k4 = (int)(r % K4);
    const int tile = (int)(r / K4);
    const long qi = (long)tile * NQ + j;
    out[idx] = qi < nq ? q[qi * K4 + k4] : make_float4(0.f, 0.f, 0.f, 0.f);
}

#define HAC_MFMA4_32(accv, av, bv)                                                   \
    accv = __builtin_amdgcn_mfma_f32_32x32x1f32((av).x, (bv).x, accv, 0, 0, 0);      \
    accv = __builtin_amdgcn_mfma_f32_32x32x1f32((av).y, (bv).y, accv, 0, 0, 0);      \
    accv = __builtin_amdgcn_mfma_f32_32x32x1f32((av).z, (bv).z, accv, 0, 0, 0);      \
    accv = __builtin_amdgcn_mfma_f32_32x32x1f32((av).w, (bv).w, accv, 0, 0, 0);

template <int NT, int W>
__global__ __launch_bounds__(W * 64) void scanq_kernel(ScanArgs a) {
    constexpr int NQ = 32 * NT;
    constexpr int KC = 16;                 // chunks (of 4 k) per staged query slice
    constexpr int NTHR = W * 64;
    constexpr int STG = KC * NQ / NTHR;    // float4 staged per thread per slice
    static_assert(KC * NQ % NTHR == 0, "slice must divide evenly over the workgroup");
    static_assert(KC % PF == 0, "prefetch ring must divide the slice");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K4 = a.K4;
    const int NS = K4 / KC;
    VGrid vg;
    if (!vgrid(a, NQ, vg)) return;
    const int q0 = vg.by * NQ;
    const int NQr = min(NQ, vg.nq - q0);
    const int C = a.C;

    f4 *qs = reinterpret_cast<f4 *>(smem);                         // [2][KC][NQ]
    u64 *cand = reinterpret_cast<u64 *>(smem + (size_t)2 * KC * NQ * 16);  // [NQ][C]
    u32 *cnt = reinterpret_cast<u32 *>(cand + (size_t)NQ * C);             // [NQ]
    float *thr = reinterpret_cast<float *>(cnt + NQ);                      // [NQ]
    u32 *ovf = reinterpret_cast<u32 *>(thr + NQ);                          // [1]
    gf4ptr qsrc = as_global(a.qt) + (size_t)vg.by * K4 * NQ;

    for (int i = tid; i < NQ; i += NTHR) {
        cnt[i] = 0;
        // padded queries admit nothing: NaN, not +inf -- a padded row is whatever the caller's buffer holds behind the count
        // (device-decided fallback), its scores can be +inf, +inf >= +inf passes, and the lists of padded queries are never
        // compacted: the overflow loop then never ends (found by a soak run; tests: stale fallback rows)
        thr[i] = i < NQr ? (a.thr_init ? a.thr_init[q0 + i] : -INFINITY) : __builtin_nanf("");
    }
    if (tid == 0) ovf[0] = ovf[1] = ovf[2] = 0;   // [0] overflow flag of the pass, [1] overflow passes of the round, [2] gave up
#pragma unroll
    for (int i = 0; i < STG; ++i) qs[tid + i * NTHR] = qsrc[tid + i * NTHR];
    __syncthreads();

    const int jl = lane & 31;
    const u32 stride = vg.nx * W;
    const u32 nrounds = (a.n_items + stride - 1) / stride;
    const u32 hw = (u32)(C + a.k) / 2;  // compact once a buffer is past the midpoint of its slack

    u32 item = vg.bx * W + w;
    bool have = item < a.n_items;
    gf4ptr gp = group_ptr(a, a.g_first + (have ? item : 0) * a.g_step) + lane;
    f4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = gp[i * 64];
    int par = 0;

    for (u32 r = 0; r < nrounds; ++r) {
        const u32 g = a.g_first + item * a.g_step;
        const u32 nitem = item + stride;
        const bool have_next = nitem < a.n_items;
        gf4ptr np = have_next ? group_ptr(a, a.g_first + nitem * a.g_step) + lane : gp;
        f32x32 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int e = 0; e < 32; ++e) acc[n][e] = 0.f;

        // One explicit drain per round gives the slice loop a clean entry state; without it hipcc's
        // waitcnt pass merges the (unordered) epilogue state into the loop header and emits
        // vmcnt(0) at the top of EVERY slice instead of the counted vmcnt(PF-1).
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
        for (int s = 0; s < NS; ++s) {
            const int sn = (s + 1 == NS) ? 0 : s + 1;
            gf4ptr qnext = qsrc + (size_t)sn * KC * NQ + tid;
            f4 st[STG];
            // No `if (have)` around the slice: a wave without work (last round only) scores group 0
            // again and discards it.  A branch here makes hipcc join the two paths before the LDS
            // write and take the idle path's vmcnt(0), draining the prefetch ring every slice.
            {
                const f4 *qcur = qs + par * (KC * NQ) + jl;
                const bool last = (s + 1 == NS);
                // B fragments are software-pipelined one chunk ahead; sched_barrier keeps hipcc from
                // hoisting a whole slice of LDS reads (64*NT VGPRs) above the MFMAs.
                f4 bcur[NT], bnxt[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) bcur[n] = qcur[n * 32];
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    if (c + 1 < KC) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) bnxt[n] = qcur[(c + 1) * NQ + n * 32];
                    }
                    const f4 av = ring[c % PF];
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        HAC_MFMA4_32(acc[n], av, bcur[n])
                    }
                    // refill after the MFMAs that read the slot, so the load reuses its registers
                    gf4ptr src = (last && c >= KC - PF) ? np + (c - (KC - PF)) * 64 : gp + (s * KC + c + PF) * 64;
                    ring[c % PF] = *src;
                    if (c == KC - 4) {  // next query slice: issued late so its registers live for 4 chunks only
#pragma unroll
                        for (int i = 0; i < STG; ++i) st[i] = qnext[i * NTHR];
                    }
#pragma unroll
                    for (int n = 0; n < NT; ++n) bcur[n] = bnxt[n];
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < STG; ++i) qs[(par ^ 1) * (KC * NQ) + tid + i * NTHR] = st[i];
            }
            __syncthreads();
            par ^= 1;
        }
        gp = np;

        // ---- epilogue: threshold filter, overflow-safe append, compaction by the owning waves
        u32 pend[NT];
        {
            const long rem = a.n_rows - (long)g * GROUP_ROWS;
            const int rows_valid = rem < GROUP_ROWS ? (int)rem : GROUP_ROWS;
            const int rbase = 4 * (lane >> 5);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                pend[n] = 0u;
                const float th = thr[n * 32 + jl];
                if (have) {
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        const int row = 32 * (e >> 4) + (e & 3) + 8 * ((e >> 2) & 3) + rbase;
                        if (acc[n][e] >= th && row < rows_valid) pend[n] |= (1u << e);
                    }
                }
            }
        }
        // Every pass that leaves something pending has compacted the lists that were full, and a compacted list has room
        // (C - max(hw, k) >= 1 slots): a query's W * 64 scores of a round are placed after at most W * 64 + 1 passes.  The
        // bound is what turns a broken invariant (e.g. a threshold that admits a score its list can never take) into
        // HAC_ERR_INTERNAL instead of a hang.
        // The pass count lives in LDS (ovf[1]), counted by thread 0 on the overflow route only: the common round pays nothing.
        for (;;) {
            bool anyp = false;
#pragma unroll
            for (int n = 0; n < NT; ++n) anyp |= (pend[n] != 0u);
            if (__any(anyp)) {
                // Rare path.  The tile's scores are copied to a per-lane scratch array so that the
                // survivors can be walked with a run-time loop over the set bits; an unrolled
                // 32*NT-way append would cost ~80 VGPRs in the hot loop's allocation.
                const int rbase = 4 * (lane >> 5);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (pend[n] != 0u) {
                        float sc[32];
#pragma unroll
                        for (int e = 0; e < 32; ++e) sc[e] = acc[n][e];
                        const int j = n * 32 + jl;
                        u32 m = pend[n];
                        while (m) {
                            const int e = __builtin_ctz(m);
                            m &= m - 1u;
                            const u32 pos = atomicAdd(&cnt[j], 1u);
                            if (pos < (u32)C) {
                                const int row = 32 * (e >> 4) + (e & 3) + 8 * ((e >> 2) & 3) + rbase;
                                cand[(size_t)j * C + pos] = make_key(sc[e], a.pos_base + g * GROUP_ROWS + row);
                                pend[n] &= ~(1u << e);
                            } else {
                                *ovf = 1u;
                            }
                        }
                    }
                }
            }
            __syncthreads();  // (B) appends of this pass visible, overflow flag final
            const bool over = (*ovf != 0u);
            for (int jj = w; jj < NQr; jj += W) {
                u32 n = cnt[jj];
                if (n > (u32)C) n = (u32)C;
                float t_new = -INFINITY;
                const bool compact = n > hw;  // wave-uniform; a full buffer (n == C) always compacts
                if (compact) {
                    wave_sort_desc(cand + (size_t)jj * C, n, lane);
                    t_new = ord2f((u32)(cand[(size_t)jj * C + a.k - 1] >> 32));
                }
                if (lane == 0) {
                    float t_cur = thr[jj];
                    if (compact) {
                        cnt[jj] = a.k;
                        if (t_new > t_cur) {
                            t_cur = t_new;
                            atomicMax(&a.thr_glob[q0 + jj], f2ord(t_new));
                        }
                    }
                    const u32 go = __hip_atomic_load(&a.thr_glob[q0 + jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (go != 0u) t_cur = fmaxf(t_cur, ord2f(go));
                    thr[jj] = t_cur;
                }
            }
            __syncthreads();  // (A) compactions done, thresholds final
            if (!over) break;  // workgroup-uniform
            if (tid == 0) {
                *ovf = 0u;
                const u32 np = ovf[1] + 1u;   // passes of this round that ended in an overflow
                ovf[1] = np;
                if (__builtin_expect(np >= scan_pass_bound(a, W * 64 + 4), 0)) ovf[2] = 1u;   // no legal input gets here
            }
            // re-filter what is still pending against the raised thresholds, then offer it again
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (pend[n] == 0u) continue;
                const float th = thr[n * 32 + jl];
#pragma unroll
                for (int e = 0; e < 32; ++e)
                    if ((pend[n] & (1u << e)) && !(acc[n][e] >= th)) pend[n] &= ~(1u << e);
            }
            __syncthreads();  // flag reset ordered before the next pass's overflow stores
            if (__builtin_expect(*(volatile u32 *)(ovf + 2) != 0u, 0)) {   // workgroup-uniform
                static_assert(NQ <= NTHR, "one thread per query of the tile");
                scan_overrun(a, q0, NQr, tid);   // poisoned tile, on with the next round
                __syncthreads();                 // every wave has read ovf[2] before tid 0 clears it below (as in scanh_kernel)
                break;
            }
        }
        if (tid == 0) ovf[1] = ovf[2] = 0u;   // (read again only behind the barriers of a later overflow pass)
        item = nitem;
        have = have_next;
    }

    for (int jj = w; jj < NQr; jj += W) {
        u32 n = cnt[jj];
        if (n > (u32)C) n = (u32)C;
        flush_survivors(a, vg.nx, q0 + jj, cand + (size_t)jj * C, n, lane);
    }
}

// ------------------------------------------------------------------ threshold seeding
// A lower bound of every query's final k-th score lets the scan kernels reject almost every row
// with one compare.  It is the exact k-th largest score over an evenly strided sample of groups:
//   sample_scores_kernel  best score of every 16-row quarter of the sample groups, [nq][4*n_groups]
//   kth_select_kernel     per query, exact k-th largest of those maxima by 4-pass 8-bit radix select
// The bound only filters; results never depend on it.
__global__ __launch_bounds__(SCAN_WAVES * 64) void sample_scores_kernel(ScanArgs a, float *__restrict__ scores, u32 S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K4 = a.K4;
    const int q0 = blockIdx.y * 16;
    const int QTr = min(16, a.nq - q0);
    f4 *ldsQ = reinterpret_cast<f4 *>(smem);  // [K4][QTr]
    for (int idx = tid; idx < K4 * QTr; idx += SCAN_WAVES * 64) {
        const int k4 = idx / QTr, j = idx - k4 * QTr;
        ldsQ[idx] = as_global(a.q)[(size_t)(q0 + j) * K4 + k4];
    }
    __syncthreads();
    const u32 item = blockIdx.x * SCAN_WAVES + w;
    if (item >= a.n_items) return;
    const u32 g = a.g_first + item * a.g_step;
    const int j = lane & 15;
    const f4 *qb = ldsQ + min(j, QTr - 1);
    gf4ptr gp = group_ptr(a, g) + lane;
    f4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = gp[i * 64];
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f4 bcur = qb[0], bnxt;
    const int NB = K4 / PF;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int tb = 0; tb < NB; ++tb) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int t = tb * PF + i;
            bnxt = qb[min(t + 1, K4 - 1) * QTr];
            const f4 av = ring[i];
            HAC_MFMA4(av, bcur)
            ring[i] = gp[min(t + PF, K4 - 1) * 64];
            bcur = bnxt;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // per (query, group): the best score of each 16-row quarter.  The k-th largest of these maxima is a
    // lower bound of the k-th largest score overall (k distinct rows reach it) and, maxima being tightly
    // distributed, a sharp one: well under 1 % of the rows pass it at k = 100 with ~250 sample groups.
    const long rem = a.n_rows - (long)g * GROUP_ROWS;
    float m = -INFINITY;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int row = 16 * (rr >> 2) + 4 * (lane >> 4) + (rr & 3);
        if (row < rem) m = fmaxf(m, acc[rr]);
    }
    // FOUR maxima per group (one per lane quarter = 16 of its rows): four times the sample points for
    // the same MFMA work, and maxima of 16 are still tight enough for the k-th of them to sit high.
    if (j < QTr) scores[(size_t)(q0 + j) * S + (size_t)item * 4 + (lane >> 4)] = m;
}

// thr[q] = k-th largest of scores[q][0..S) (NaN counts as -inf); -inf when fewer than k finite-or-inf entries
__global__ __launch_bounds__(256) void kth_select_kernel(const float *__restrict__ scores, u32 S, int k, float *__restrict__ thr) {
    __shared__ u32 hist[256];
    __shared__ u32 wtot[4];
    __shared__ u32 sel_prefix, sel_k;
    const int tid = threadIdx.x;
    const float *src = scores + (size_t)blockIdx.x * S;
    u32 prefix = 0, mask = 0, kk = (u32)k;  // keys matching (key & mask) == prefix are still candidates
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0;
        __syncthreads();
        const int shift = pass * 8;
        // (scores of one query share their sign and most exponent bits: a thread counts runs of one bin locally instead of
        // sending every element to the same LDS counter)
        u32 run_bin = 0, run = 0;
        for (u32 i = tid; i < S; i += 256) {
            const float v = src[i];
            const u32 key = (v != v) ? 0u : f2ord(v + 0.0f);
            if ((key & mask) == prefix) {
                const u32 b = (key >> shift) & 255u;
                if (b != run_bin && run) {
                    atomicAdd(&hist[run_bin], run);
                    run = 0;
                }
                run_bin = b;
                ++run;
            }
        }
        if (run) atomicAdd(&hist[run_bin], run);
        __syncthreads();
        {   // bin b holds the kk-th largest key iff  above(b) < kk <= above(b) + hist[b],  above(b) = sum of bins > b.
            // Parallel suffix sum over the 256 bins (one per thread): wave scan + cross-wave offsets.
            const u32 hcnt = hist[tid];
            u32 inc = hcnt;  // inclusive suffix sum within the wave (towards higher lanes)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 v = __shfl_down(inc, o);
                if ((tid & 63) + o < 64) inc += v;
            }
            if ((tid & 63) == 0) wtot[tid >> 6] = inc;   // total of this wave's 64 bins
            __syncthreads();
            u32 above = inc - hcnt;
            for (int ww = (tid >> 6) + 1; ww < 4; ++ww) above += wtot[ww];
            const bool hit = above < kk && kk <= above + hcnt;
            // fewer than kk candidates in total: fall to bin 0 (the caller reports -inf for prefix 0 chains)
            const bool none = (tid == 0) && (above + hcnt < kk);
            if (hit || none) {
                sel_prefix = prefix | ((u32)tid << shift);
                sel_k = hit ? kk - above : 1u;
            }
        }
        __syncthreads();
        prefix = sel_prefix;
        kk = sel_k;
        mask |= 255u << shift;
        __syncthreads();
    }
    // prefix is the key of the k-th largest element if at least k keys exist; with fewer, the walk
    // bottoms out in bin 0 chains (key 0 = NaN/-nothing): report -inf
    if (tid == 0) thr[blockIdx.x] = (S >= (u32)k && prefix != 0u) ? ord2f(prefix) : -INFINITY;
}

// ------------------------------------------------------------------ merge kernel
// One workgroup per query: streams L lists of k keys, keeps the k largest.
// Element (l, q, i) lives at lists[l*stride_l + q*stride_q + i].
__global__ __launch_bounds__(MERGE_THREADS) void merge_keys_kernel(const u64 *__restrict__ lists, int L,
                                                                   size_t stride_l, size_t stride_q, int k, int Cm,
                                                                   u64 *__restrict__ out, float *__restrict__ kth_out,
                                                                   const u32 *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *buf = reinterpret_cast<u64 *>(smem);          // [Cm]
    u32 *cnt = reinterpret_cast<u32 *>(buf + Cm);      // [1]
    u64 *thrk = reinterpret_cast<u64 *>(cnt + 2);      // [1]
    const int tid = threadIdx.x;
    const size_t q = blockIdx.x;
    if (tid == 0) {
        *cnt = 0;
        *thrk = 0;
    }
    __syncthreads();
    // slab mode: L lists of k slots.  dense mode (counts != null): counts[q] keys stored back to back.
    const long total = counts ? (long)min((unsigned long long)counts[q], (unsigned long long)L * k) : (long)L * k;
    for (long base = 0; base < total; base += MERGE_THREADS) {
        const long idx = base + tid;
        u64 key = 0;
        if (idx < total) {
            if (counts) {
                key = lists[q * stride_q + idx];
            } else {
                const long l = idx / k;
                const int i = (int)(idx - l * k);
                key = lists[(size_t)l * stride_l + q * stride_q + i];
            }
        }
        if (key > *thrk) {
            const u32 pos = atomicAdd(cnt, 1u);
            buf[pos] = key;
        }
        __syncthreads();
        const u32 n = *cnt;
        __syncthreads();  // nobody appends for the next chunk before everyone has read n
        if (n > (u32)(Cm - MERGE_THREADS)) {  // uniform
            block_sort_desc(buf, n, tid, MERGE_THREADS);
            if (tid == 0) {
                *cnt = k;
                *thrk = buf[k - 1];
            }
            __syncthreads();
        }
    }
    const u32 n = *cnt;
    if (n > 1) block_sort_desc(buf, n, tid, MERGE_THREADS);
    const u32 keep = n < (u32)k ? n : (u32)k;
    for (u32 i = tid; i < (u32)k; i += MERGE_THREADS) out[q * k + i] = i < keep ? buf[i] : 0ull;
    if (kth_out && tid == 0) kth_out[q] = (n >= (u32)k) ? ord2f((u32)(buf[k - 1] >> 32)) : -INFINITY;
}

// keys -> faiss-style (D, I); empty slots: -FLT_MAX / -1
__global__ void keys_to_results_kernel(const u64 *__restrict__ keys, long n, const long long *__restrict__ id_map,
                                       float *__restrict__ D, long long *__restrict__ I) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 key = keys[i];
    if (key == 0ull) {
        D[i] = -FLT_MAX;
        I[i] = -1;
    } else {
        const u32 pos = 0xFFFFFFFFu - (u32)key;
        D[i] = ord2f((u32)(key >> 32));
        I[i] = id_map ? id_map[pos] : (long long)pos;
    }
}

// In-process multi-device index: a shard numbers its rows densely in the order it received them, while the
// contract numbers rows by insertion order over the WHOLE index (every add() is split across the shards).
// spans[i] = (first shard-local row, rows, first global row) of the shard's share of add() number i; both
// numberings grow with i, so the translation keeps a sorted key list sorted.
struct SpanDesc {
    u32 local0, count, global0, pad;
};
__global__ void remap_positions_kernel(u64 *__restrict__ keys, long n, const SpanDesc *__restrict__ spans, int n_spans) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 key = keys[i];
    if (key == 0ull) return;
    const u32 pos = 0xFFFFFFFFu - (u32)key;
    int lo = 0, hi = n_spans - 1;   // last span with local0 <= pos
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (spans[mid].local0 <= pos) lo = mid;
        else hi = mid - 1;
    }
    const u32 g = spans[lo].global0 + (pos - spans[lo].local0);
    keys[i] = (key & 0xFFFFFFFF00000000ull) | (u64)(0xFFFFFFFFu - g);
}

u32 next_pow2(u32 v) {
    u32 p = 1;
    while (p < v) p <<= 1;
    return p;
}

#include "scan_split.inc"

// ------------------------------------------------------------------ one-device index
struct Segment {
    float4 *buf = nullptr;   // cap_rows * d fp32 (T64 tiles)
    uint4 *hbuf = nullptr;   // cap_rows * d fp16 (H64 image): allocated by the first search that takes the prefilter path
    int64_t cap_rows = 0;    // multiple of 64
    int64_t rows = 0;
    int64_t h_rows = 0;      // rows whose fp16 image is current (tile_rows_kernel keeps it current once hbuf exists)
    float4 *rbuf = nullptr;  // cap_rows * d fp32, row-major: the rescoring's copy (small indexes only, best effort: ensure_row_image)
    int64_t r_rows = 0;      // rows of it that are current
};

struct DeviceIndex {
    int d = 0, K4 = 0, device = 0, n_cu = 256;
    hipStream_t stream = nullptr;  // for the synchronous host API
    hipStream_t stream2 = nullptr; // rescoring of one query chunk runs here, under the next chunk's scan
    hipEvent_t ev_chunk[2] = {nullptr, nullptr}, ev_tail = nullptr;
    std::vector<Segment> segs;
    int64_t ntotal = 0;
    SegDesc *d_segs = nullptr;
    const float4 **d_rimg = nullptr, **h_rimg = nullptr;   // per live segment: its row-major copy for the rescoring, or null (ensure_row_image)
    bool segs_dirty = true;
    // Tuning / test switches.  The HAC_* environment variables are read ONCE, when the index is created, as the
    // defaults; hac_index_set_option() changes them on a live handle (tests, bench).
    struct Tuning {
        int split = -1;          // prefilter: -1 by size, 0 never, 1 whenever the shape allows
        int split_terms = 1;     // fp16 products per score at the first cascade level (1 or 3)
        bool force_scan16 = false;
        int scanq_nt = 0;        // 0: chosen by padding; 1..4 pins the query tiles per workgroup
        int scanq_waves = 8;
        bool no_p8 = false;
        int seed_groups_max = 0;         // cap of the seeding pass of the prefilter scan, in 64-row groups; 0 = 14 sqrt(groups)
        int split_decide = -1;           // who reads the certificates: -1 by entry point (host API: host, *_device: device), 0 host, 1 device
        bool no_halfq = false;           // development ("scan_halfq" = "0"): few-query searches run the full-tile instantiation
        bool image_eager = false;        // "fp16_image" = "eager": the fp16 image is written by add() (new segments), not by a later search
        int rescore_rows = -1;           // row-major copy for the rescoring: -1 by size (<= RESCORE_ROWS_MAX rows), 0 never, 1 whenever it can be allocated
        int debug_max_pass = 0;          // tests: pass bound of the candidate loops (0 = the kernels' own, which no legal input reaches)
        int debug_oom = 0;               // tests: the next N allocations the index needs behave as if their first attempt had found the device full (malloc_reclaiming)
        int scan_passes = 0;             // prefilter scan: 0 by size, 1..5 pins the number of passes (threshold refreshes between them)
        int pass_cut[2] = {0, 0};        // where the passes end, in thousandths of the row groups (option "scan_pass_cuts" = "a,b"); 0 = by size
    } tune;
    // Passes of the prefilter's main scan (search_keys_split) and where they end (row groups, whole rounds).  A refresh costs a
    // kernel tail + a selection (~0.1 ms): only scans of a few milliseconds are cut, and only with seeded thresholds (the refresh
    // raises them in place).  Measured optimum over 2M .. 25M rows (tools/ab_search.py sweeps, round 4): a SHORT seeding pass (1024
    // groups: its only job is to keep the first pass's lists from overflowing), the first cut after ~6k groups, the second at the
    // geometric mean of the first and the corpus (4M rows: 10 % / 31 %); a fourth pass is worth another 0.05 ms from ~8M rows
    // (25M: cuts at 1.6 % / 6.3 % / 25 %), a fifth nothing.  Below ~1.5M rows one pass is faster.
    static constexpr int MAX_PASSES = 5;
    int scan_passes(u32 G, u32 round_groups, bool seeded, u32 bounds[MAX_PASSES + 1]) const {
        bounds[0] = 0u;
        for (int i = 1; i <= MAX_PASSES; ++i) bounds[i] = G;
        if (!seeded) return 1;
        int n = tune.scan_passes > 0 ? tune.scan_passes : (G >= 131072u ? 4 : G >= 24576u ? 3 : 1);   // from 8.4M / 1.6M rows
        if (n >= 2) {
            // cuts c_1 .. c_{n-1} in geometric progression from c_1 towards G (n = 3 with the option's explicit second cut: that one)
            const double c1 = tune.pass_cut[0] ? tune.pass_cut[0] * 1e-3 * G : std::max(6144.0, 0.015 * G);
            const double ratio = std::pow((double)G / c1, 1.0 / (n - 1));
            double c = c1;
            for (int i = 1; i < n; ++i, c *= ratio) {
                const double ci = (i == 2 && n == 3 && tune.pass_cut[1]) ? tune.pass_cut[1] * 1e-3 * G : c;
                bounds[i] = (u32)(ci / round_groups + 0.5) * round_groups;
            }
            bool ok = true;
            for (int i = 1; i < n; ++i) ok = ok && bounds[i - 1] < bounds[i] && bounds[i] < G;
            if (!ok) {   // (small corpora with a pinned pass count)
                n = 1;
                for (int i = 1; i <= MAX_PASSES; ++i) bounds[i] = G;
            }
        }
        return n;
    }
    void read_env() {
        if (const char *e = getenv("HAC_SPLIT")) tune.split = e[0] == '0' ? 0 : (e[0] == '1' ? 1 : -1);
        if (const char *e = getenv("HAC_SPLIT_TERMS")) tune.split_terms = e[0] == '3' ? 3 : 1;
        if (const char *e = getenv("HAC_FORCE_SCAN16")) tune.force_scan16 = e[0] == '1';
        if (const char *e = getenv("HAC_SCANQ_NT")) tune.scanq_nt = atoi(e);
        if (const char *e = getenv("HAC_SCANQ_WAVES")) tune.scanq_waves = e[0] == '4' ? 4 : 8;
        if (getenv("HAC_SCAN_NO_P8")) tune.no_p8 = true;
    }
    // a value outside the documented set is an error, never a silent default (a mistyped value in a cross-check test
    // would otherwise exercise the wrong kernels and still pass)
    int set_option(const char *name, const char *value) {
        const std::string n(name), v(value ? value : "");
        auto one_of = [&](std::initializer_list<const char *> allowed) {
            for (const char *a : allowed)
                if (v == a) return true;
            std::string list;
            for (const char *a : allowed) list += std::string(list.empty() ? "" : " | ") + a;
            (void)fail(HAC_ERR_INVALID, "index option %s = '%s': %s", name, v.c_str(), list.c_str());
            return false;
        };
        if (n == "split") {
            if (!one_of({"0", "1", "auto"})) return HAC_ERR_INVALID;
            tune.split = v == "0" ? 0 : (v == "1" ? 1 : -1);
        } else if (n == "split_terms") {
            if (!one_of({"1", "3"})) return HAC_ERR_INVALID;
            tune.split_terms = v == "3" ? 3 : 1;
        } else if (n == "force_scan16") {
            if (!one_of({"0", "1"})) return HAC_ERR_INVALID;
            tune.force_scan16 = v == "1";
        } else if (n == "scanq_nt") {
            if (!one_of({"0", "1", "2", "3", "4"})) return HAC_ERR_INVALID;
            tune.scanq_nt = atoi(v.c_str());
        } else if (n == "scanq_waves") {
            if (!one_of({"4", "8"})) return HAC_ERR_INVALID;
            tune.scanq_waves = v == "4" ? 4 : 8;
        } else if (n == "scan_no_p8") {
            if (!one_of({"0", "1"})) return HAC_ERR_INVALID;
            tune.no_p8 = v == "1";
        } else if (n == "split_decide") {
            if (!one_of({"auto", "host", "device"})) return HAC_ERR_INVALID;
            tune.split_decide = v == "host" ? 0 : (v == "device" ? 1 : -1);
        } else if (n == "scan_pass_cuts") {
            int a = 0, b = 0;
            if (v == "auto") a = b = 0;
            else if (sscanf(v.c_str(), "%d,%d", &a, &b) != 2 || a < 1 || b <= a || b > 900) return HAC_ERR_INVALID;
            tune.pass_cut[0] = a;
            tune.pass_cut[1] = b;
        } else if (n == "scan_passes") {
            if (!one_of({"auto", "1", "2", "3", "4", "5"})) return HAC_ERR_INVALID;
            tune.scan_passes = v == "auto" ? 0 : atoi(v.c_str());
        } else if (n == "scan_halfq") {
            if (!one_of({"0", "1"})) return HAC_ERR_INVALID;
            tune.no_halfq = v == "0";
        } else if (n == "fp16_image") {
            if (!one_of({"lazy", "eager"})) return HAC_ERR_INVALID;
            tune.image_eager = v == "eager";
        } else if (n == "rescore_rows") {
            if (!one_of({"0", "1", "auto"})) return HAC_ERR_INVALID;
            tune.rescore_rows = v == "0" ? 0 : (v == "1" ? 1 : -1);
        } else if (n == "debug_oom") {
            char *end = nullptr;
            const long t = strtol(v.c_str(), &end, 10);
            if (v.empty() || *end || t < 0 || t > 1000) return fail(HAC_ERR_INVALID, "index option debug_oom = '%s': an integer 0..1000", v.c_str());
            tune.debug_oom = (int)t;
        } else if (n == "debug_max_pass") {
            char *end = nullptr;
            const long t = strtol(v.c_str(), &end, 10);
            if (v.empty() || *end || t < 0 || t > 1000000) return fail(HAC_ERR_INVALID, "index option debug_max_pass = '%s': an integer >= 0", v.c_str());
            tune.debug_max_pass = (int)t;
        } else if (n == "seed_groups_max") {
            char *end = nullptr;
            const long t = strtol(v.c_str(), &end, 10);
            if (v.empty() || *end || t < 0) return fail(HAC_ERR_INVALID, "index option seed_groups_max = '%s': an integer >= 0 (0 = 14 sqrt(groups))", v.c_str());
            tune.seed_groups_max = t <= 0 ? 0 : (int)std::max<long>(768, std::min<long>(t, 1 << 30));
        } else {
            return fail(HAC_ERR_INVALID, "unknown index option '%s'", name);
        }
        return HAC_OK;
    }
    GrowBuf ws_partial, ws_pcnt, ws_seedkeys, ws_thr, ws_thrglob, ws_q, ws_qt, ws_keys, ws_D, ws_I, ws_stage[2];
    // split-bf16 prefilter path (scan_split.inc)
    GrowBuf ws_norm, ws_qsplit, ws_delta, ws_cand, ws_akeys, ws_fail, ws_stat;
    GrowBuf ws_err;            // [0] device error bits (sticky until read), [1] workgroups that hit a pass bound (ScanArgs::err)
    u32 *h_err = nullptr;      // pinned copy

    GrowBuf ws_fbidx[2], ws_fbq[2], ws_fbkeys[2];   // per cascade level: failed queries, their matrix, their keys
    u32 *h_fb = nullptr;       // pinned: [0] failed queries, [1] max |s~ - s| / delta (float bits), [2..] flags / indices
    size_t h_fb_words = 0;
    long split_searches = 0, split_fallback_queries = 0;
    void *h_stage[2] = {nullptr, nullptr};
    size_t h_stage_bytes = 0;
    // Small host<->device traffic (queries, results, segment table) always goes through
    // pinned memory: pageable hipMemcpyAsync is not reliably stream-ordered on this stack.
    void *h_pin = nullptr;
    size_t h_pin_bytes = 0;
    SegDesc *h_segs = nullptr;
    int pin_reserve(size_t bytes) {
        if (bytes <= h_pin_bytes) return HAC_OK;
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr;
        h_pin_bytes = 0;
        hipError_t e = hipHostMalloc(&h_pin, bytes + bytes / 4, hipHostMallocDefault);
        if (e != hipSuccess) return fail(HAC_ERR_OOM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        h_pin_bytes = bytes + bytes / 4;
        return HAC_OK;
    }
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    // profiling: one hipEvent pair per search around the main scan kernel, on the launch stream
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;

    int init(int d_, int device_) {
        d = d_;
        K4 = d / 4;
        device = device_;
        read_env();
        DeviceGuard g(device);
        if (!g.ok) return fail(HAC_ERR_HIP, "cannot select HIP device %d (no MI355X visible?)", device);
        hipDeviceProp_t prop;
        HAC_HIP(hipGetDeviceProperties(&prop, device));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HAC_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HAC_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) HAC_HIP(hipEventCreateWithFlags(&ev_chunk[i], hipEventDisableTiming));
        HAC_HIP(hipEventCreateWithFlags(&ev_tail, hipEventDisableTiming));
        HAC_HIP(hipMalloc((void **)&d_segs, sizeof(SegDesc) * MAX_SEG));
        HAC_HIP(hipHostMalloc((void **)&h_segs, sizeof(SegDesc) * MAX_SEG, hipHostMallocDefault));
        HAC_HIP(hipMalloc((void **)&d_rimg, sizeof(float4 *) * MAX_SEG));
        HAC_HIP(hipMemset(d_rimg, 0, sizeof(float4 *) * MAX_SEG));
        HAC_HIP(hipHostMalloc((void **)&h_rimg, sizeof(float4 *) * MAX_SEG, hipHostMallocDefault));
        for (int i = 0; i < 2; ++i) HAC_HIP(hipEventCreateWithFlags(&stage_ev[i], hipEventDisableTiming));
        HAC_TRY(ws_norm.reserve(16));
        HAC_HIP(hipMemsetAsync(ws_norm.p, 0, 16, stream));
        HAC_TRY(ws_err.reserve(16));   // (a new GrowBuf is zero)
        HAC_HIP(hipHostMalloc((void **)&h_err, 16, hipHostMallocDefault));
        h_err[0] = h_err[1] = 0u;
        HAC_HIP(hipHostMalloc((void **)&h_plan, 8 * PLAN_RING, hipHostMallocDefault));
        for (int i = 0; i < PLAN_RING; ++i) {
            h_plan[2 * i] = h_plan[2 * i + 1] = 0u;
            HAC_HIP(hipEventCreateWithFlags(&ev_plan[i], hipEventDisableTiming));
        }
        static bool attr_done[64] = {false};
        if (device < 64 && !attr_done[device]) {
            HAC_HIP(hipFuncSetAttribute((const void *)scan16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)LDS_LIMIT));
            HAC_HIP(hipFuncSetAttribute((const void *)merge_keys_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(64 * 1024)));
            const void *fq[] = {(const void *)scanq_kernel<1, 4>, (const void *)scanq_kernel<2, 4>, (const void *)scanq_kernel<3, 4>,
                                (const void *)scanq_kernel<4, 4>, (const void *)scanq_kernel<1, 8>, (const void *)scanq_kernel<2, 8>,
                                (const void *)scanq_kernel<3, 8>, (const void *)scanq_kernel<4, 8>};
            for (const void *f : fq) HAC_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
            HAC_HIP(hipFuncSetAttribute((const void *)sample_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * 1024)));
            const void *fh[] = {(const void *)scanh_kernel<1, false>, (const void *)scanh_kernel<3, false>,
                                (const void *)scanh_kernel<1, true>, (const void *)scanh_kernel<3, true>,
                                (const void *)scanh_kernel<1, false, 8>, (const void *)scanh_kernel<1, true, 8>,
                                (const void *)scanh_kernel<1, false, 4>, (const void *)scanh_kernel<1, true, 4>};
            for (const void *f : fh) HAC_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
            attr_done[device] = true;
        }
        return HAC_OK;
    }

    void destroy() {
        DeviceGuard g(device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto &s : segs) {
            if (s.buf) (void)hipFree(s.buf);
            if (s.hbuf) (void)hipFree(s.hbuf);
            if (s.rbuf) (void)hipFree(s.rbuf);
        }
        segs.clear();
        if (d_segs) (void)hipFree(d_segs);
        if (h_segs) (void)hipHostFree(h_segs);
        if (d_rimg) (void)hipFree(d_rimg);
        if (h_rimg) (void)hipHostFree(h_rimg);
        if (h_pin) (void)hipHostFree(h_pin);
        for (GrowBuf *b : {&ws_partial, &ws_pcnt, &ws_seedkeys, &ws_thr, &ws_thrglob, &ws_q, &ws_qt, &ws_keys, &ws_D, &ws_I, &ws_stage[0],
                           &ws_stage[1], &ws_err, &ws_norm, &ws_qsplit, &ws_delta, &ws_cand, &ws_akeys, &ws_fail, &ws_stat, &ws_fbidx[0], &ws_fbidx[1],
                           &ws_fbq[0], &ws_fbq[1], &ws_fbkeys[0], &ws_fbkeys[1]})
            b->release();
        if (h_fb) (void)hipHostFree(h_fb);
        if (h_err) (void)hipHostFree(h_err);
        if (h_plan) (void)hipHostFree(h_plan);
        for (hipEvent_t ev : ev_plan)
            if (ev) (void)hipEventDestroy(ev);
        for (int i = 0; i < 2; ++i) {
            if (h_stage[i]) (void)hipHostFree(h_stage[i]);
            if (stage_ev[i]) (void)hipEventDestroy(stage_ev[i]);
        }
        for (auto &e : ev_pool) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        if (stream2) {
            (void)hipStreamSynchronize(stream2);
            (void)hipStreamDestroy(stream2);
        }
        for (int i = 0; i < 2; ++i)
            if (ev_chunk[i]) (void)hipEventDestroy(ev_chunk[i]);
        if (ev_tail) (void)hipEventDestroy(ev_tail);
        if (stream) (void)hipStreamDestroy(stream);
    }

    int reset() {
        DeviceGuard g(device);
        HAC_HIP(hipStreamSynchronize(stream));
        // keep the largest allocation for reuse (the reference resets after every block, :122)
        size_t best = 0;
        for (size_t i = 1; i < segs.size(); ++i)
            if (segs[i].cap_rows > segs[best].cap_rows) best = i;
        std::vector<Segment> keep;
        for (size_t i = 0; i < segs.size(); ++i) {
            if (i == best) {
                Segment s = segs[i];
                s.rows = 0;
                s.h_rows = 0;
                s.r_rows = 0;
                keep.push_back(s);
            } else if (segs[i].buf) {
                HAC_HIP(hipFree(segs[i].buf));
                if (segs[i].hbuf) HAC_HIP(hipFree(segs[i].hbuf));
                if (segs[i].rbuf) HAC_HIP(hipFree(segs[i].rbuf));
            }
        }
        segs.swap(keep);
        ntotal = 0;
        segs_dirty = true;
        half_image_unavailable = false;
        row_image_unavailable = false;
        row_images_reclaimed = false;
        split_searches_since_add = 0;
        light_searches_since_add = 0;
        HAC_HIP(hipMemsetAsync(ws_norm.p, 0, 16, stream));
        HAC_HIP(hipMemsetAsync(ws_err.p, 0, 16, stream));
        HAC_HIP(hipStreamSynchronize(stream));
        return HAC_OK;
    }

    // the chip-wide thresholds of the next scan launch and the control words in front of them (see scan_pass_bound)
    int clear_thrglob(size_t nq_pad, hipStream_t st) {
        HAC_HIP(hipMemsetAsync(ws_thrglob.p, 0, (nq_pad + THR_CTL_WORDS) * 4, st));
        if (tune.debug_max_pass > 0) HAC_HIP(hipMemsetD32Async((hipDeviceptr_t)ws_thrglob.p, tune.debug_max_pass, 1, st));
        return HAC_OK;
    }
    u32 *thrglob() const { return (u32 *)ws_thrglob.p + THR_CTL_WORDS; }

    // The device's error word, as of everything that has completed on `st` (enqueue + wait): HAC_ERR_INTERNAL if a scan
    // workgroup hit its pass bound since the word was last read; reading clears it.
    int fetch_err(hipStream_t st) {
        HAC_HIP(hipMemcpyAsync(h_err, ws_err.p, 8, hipMemcpyDeviceToHost, st));
        HAC_HIP(hipStreamSynchronize(st));
        return check_err(st);
    }
    // h_err already copied and the stream synchronized by the caller
    int check_err(hipStream_t st) {
        if (h_err[0] == 0u) return HAC_OK;
        const u32 bits = h_err[0], n = h_err[1];
        h_err[0] = h_err[1] = 0u;
        HAC_HIP(hipMemsetAsync(ws_err.p, 0, 16, st));
        return fail(HAC_ERR_INTERNAL, "device %d: %u scan workgroup(s) reached the pass bound of their candidate loop (error bits 0x%x): a broken "
                    "invariant of the library, not of the input; the queries of those tiles were returned with EMPTY lists", device, n, bits);
    }

    // The rescoring's row-major copies (ensure_row_image) are an optional cache of up to +100 % of the corpus: any allocation the
    // index NEEDS gives them back before it reports an out-of-memory error (an add / search loop that used to fit
    // could fail once the copies had been built), and they are not built again until the next reset.
    bool drop_row_images() {
        bool any = false;
        for (auto &s : segs) {
            if (s.rbuf) {
                (void)hipFree(s.rbuf);       // (waits for the kernels that read it)
                any = true;
            }
            s.rbuf = nullptr;
            s.r_rows = 0;
        }
        if (any) segs_dirty = true;
        row_images_reclaimed = true;        // (until the next reset: an add clears row_image_unavailable, this it leaves alone)
        return any;
    }
    hipError_t malloc_reclaiming(void **p, size_t bytes) {
        hipError_t e = tune.debug_oom > 0 ? hipErrorOutOfMemory : hipMalloc(p, bytes);
        if (tune.debug_oom > 0) {
            --tune.debug_oom;
            if (!drop_row_images()) return hipMalloc(p, bytes);   // (nothing to give back: the real allocator decides)
            return hipMalloc(p, bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();     // (the failed allocation's error is sticky)
            if (drop_row_images()) e = hipMalloc(p, bytes);
            if (e != hipSuccess) (void)hipGetLastError();
        }
        return e;
    }

    // make room for m more rows: returns (segment index) whose rows..cap_rows can take them in pieces
    int new_segment(int64_t rows_needed, hipStream_t st) {
        if ((int)segs.size() >= MAX_SEG) HAC_TRY(consolidate(st));
        Segment s;
        s.cap_rows = (rows_needed + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
        const size_t bytes = (size_t)s.cap_rows * d * sizeof(float);
        if ((bytes >> 6) > 0xFFFFFFFFull) return fail(HAC_ERR_UNSUPPORTED, "a single add() of %lld rows exceeds the segment size limit (256 GiB of fp32 rows)", (long long)rows_needed);
        hipError_t e = malloc_reclaiming((void **)&s.buf, bytes);   // the fp32 tiles; the fp16 image comes with the first prefilter search
        if (e != hipSuccess) return fail(HAC_ERR_OOM, "hipMalloc(%zu) for %lld rows failed: %s", bytes, (long long)rows_needed, hipGetErrorString(e));
        // zero the last group so that padding rows are finite
        const size_t gbytes = (size_t)GROUP_ROWS * d * sizeof(float);
        HAC_HIP(hipMemsetAsync((char *)s.buf + bytes - gbytes, 0, gbytes, st));
        segs.push_back(s);
        return HAC_OK;
    }

    // The fp16 image of every segment, current up to its last row: allocated and filled the first time a search takes the
    // prefilter path (+50 % of the corpus bytes, which an index that only ever scans with a few queries, or runs with
    // split = "0", never spends), extended here for rows added since to a segment that had none.  Segments that own an image
    // get their new rows' pieces from tile_rows_kernel directly.
    // (auto mode: when the image does not fit -- +50 % of the corpus bytes on a nearly full HBM -- the images that were
    // allocated are given back, the sticky allocation error is cleared, no search tries again until the next add / reset,
    // and the exact fp32 kernels answer: search_keys.  split = "1" keeps the hard error.)
    bool half_image_unavailable = false;
    void drop_half_images() {
        for (auto &s : segs) {
            if (s.hbuf) (void)hipFree(s.hbuf);
            s.hbuf = nullptr;
            s.h_rows = 0;
        }
        segs_dirty = true;
    }
    int ensure_half_image(hipStream_t st) {
        for (auto &s : segs) {
            if (s.rows == 0 || (s.hbuf && s.h_rows == s.rows)) continue;
            if (!s.hbuf) {
                const size_t hbytes = (size_t)s.cap_rows * d * 2;
                hipError_t e = malloc_reclaiming((void **)&s.hbuf, hbytes);
                if (e != hipSuccess) {
                    s.hbuf = nullptr;
                    return fail(HAC_ERR_OOM, "hipMalloc(%zu) for the fp16 image of %lld rows failed: %s (split = \"0\" searches without it)", hbytes,
                                (long long)s.cap_rows, hipGetErrorString(e));
                }
                s.h_rows = 0;
                segs_dirty = true;
            }
            const long g_lo = (long)(s.h_rows / GROUP_ROWS), g_hi = (long)((s.rows + GROUP_ROWS - 1) / GROUP_ROWS);
            if (g_hi > g_lo) {
                half_image_kernel<<<dim3((unsigned)(g_hi - g_lo)), dim3(256), 0, st>>>(s.buf, s.hbuf, g_lo, K4);
                HAC_HIP(hipGetLastError());
            }
            s.h_rows = s.rows;
        }
        return HAC_OK;
    }

    // The rescoring reads ~130 candidate rows per query, each a different row: out of the T64 tiles that is one useful 16-byte
    // piece per 64-byte sector (0.43-0.49 ms per 1000 queries whatever the corpus size: 1.6 GB fetched for 0.4 GB used) -- the
    // price of the layout that lets scan16_kernel stream at 0.77 of the HBM rate.  On a small index that is a fifth of a
    // search (1M rows: 2.15 ms), on 25M rows 1.5 %.  So small indexes keep the rows once more, ROW-MAJOR, for the rescoring
    // alone (whole 128-byte lines per candidate): built lazily from the tiles by the first prefilter search, like the fp16
    // image, +100 % of a corpus that is small by definition, best effort (an allocation that fails just leaves the tiles'
    // route), kept current by later searches, dropped by consolidate.  Same fmaf chain over the same values: same bits.
    static constexpr int64_t RESCORE_ROWS_MAX = 12000000;
    // auto mode builds the copy with the THIRD prefilter search after the last add / reset: making it moves twice the corpus
    // bytes (1M rows: ~2 ms, the saving of about seven searches), so an index that is searched once per block -- the reference's
    // add / search / reset loop (:98-122) -- never pays for it (measured: 5 -> 12-15 ms per 2.5M-row block when it did)
    static constexpr int RESCORE_ROWS_AFTER = 2;
    int split_searches_since_add = 0;
    bool row_image_unavailable = false;
    bool row_images_reclaimed = false;
    const char *rescore_from() const {   // what the plan text says: "rows" when every live segment has a current row-major copy
        bool all = !segs.empty();
        for (auto &sg : segs)
            if (sg.rows > 0 && !(sg.rbuf && sg.r_rows == sg.rows)) all = false;
        return all ? "rows" : "tiles";
    }
    // (a search that is being captured into a HIP graph must not allocate, free or synchronize: the lazy images are neither
    // built nor dropped by it -- the capture records what the call before it ran)
    static bool stream_is_capturing(hipStream_t st) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return cs != hipStreamCaptureStatusNone;
    }
    int ensure_row_image(hipStream_t st) {
        const bool allowed = tune.rescore_rows == 1 || (tune.rescore_rows < 0 && ntotal <= RESCORE_ROWS_MAX);
        const bool capturing = stream_is_capturing(st);
        if (!allowed) {   // switched off on a live handle, or grown past the size that gets one: the copies go (hipFree waits for their readers)
            if (capturing) return HAC_OK;     // (a captured search frees nothing and leaves the segment table alone: the copies go with the next live search)
            for (auto &s : segs) {
                if (!s.rbuf) continue;
                HAC_HIP(hipFree(s.rbuf));
                s.rbuf = nullptr;
                s.r_rows = 0;
                segs_dirty = true;
            }
            return HAC_OK;
        }
        if (row_image_unavailable || (row_images_reclaimed && tune.rescore_rows < 0)) return HAC_OK;
        if (capturing) {   // use what is current, build nothing
            bool need = false;
            for (auto &s : segs) need |= s.rows > 0 && !(s.rbuf && s.r_rows == s.rows);
            if (need) return HAC_OK;
        }
        if (tune.rescore_rows < 0) {
            // an index that HAS a copy keeps it current (an add then costs the new rows once more, on the device); one that has
            // none waits for its third search
            bool have = false;
            for (auto &s : segs) have |= s.rbuf != nullptr && s.r_rows > 0;
            if (!have && split_searches_since_add++ < RESCORE_ROWS_AFTER) return HAC_OK;
        }
        for (auto &s : segs) {
            if (s.rows == 0 || (s.rbuf && s.r_rows == s.rows)) continue;
            if (!s.rbuf) {
                const size_t rbytes = (size_t)s.cap_rows * d * 4;
                if (tune.rescore_rows < 0) {      // auto: only while it leaves twice its own size free for whoever shares the device (the fp16 image, the encoder, torch)
                    size_t free_b = 0, total_b = 0;
                    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 3 * rbytes) {
                        (void)hipGetLastError();
                        row_image_unavailable = true;
                        return HAC_OK;
                    }
                }
                if (hipMalloc((void **)&s.rbuf, rbytes) != hipSuccess) {
                    s.rbuf = nullptr;
                    (void)hipGetLastError();
                    row_image_unavailable = true;      // until the next add / reset: nothing is retried per search
                    return HAC_OK;
                }
                s.r_rows = 0;
            }
            const long g_lo = (long)(s.r_rows / GROUP_ROWS), g_hi = (long)((s.rows + GROUP_ROWS - 1) / GROUP_ROWS);
            if (g_hi > g_lo) {
                untile_rows_kernel<<<dim3((unsigned)(g_hi - g_lo)), dim3(256), 0, st>>>(s.buf, s.rbuf, g_lo, K4);
                HAC_HIP(hipGetLastError());
            }
            s.r_rows = s.rows;
            segs_dirty = true;
        }
        return HAC_OK;
    }

    // fuse all segments into one allocation (groups are self-contained, so this is plain copies)
    int consolidate(hipStream_t st) {
        int64_t total_cap = 0;
        for (auto &s : segs) total_cap += (s.rows + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
        Segment big;
        big.cap_rows = total_cap;
        {
            const hipError_t e = malloc_reclaiming((void **)&big.buf, (size_t)total_cap * d * 4);
            if (e != hipSuccess) return fail(HAC_ERR_OOM, "hipMalloc(%zu) to fuse %zu segments failed: %s", (size_t)total_cap * d * 4, segs.size(), hipGetErrorString(e));
        }
        int64_t off_rows = 0;
        for (auto &s : segs) {
            const int64_t gr = (s.rows + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
            if (gr) HAC_HIP(hipMemcpyAsync((char *)big.buf + (size_t)off_rows * d * 4, s.buf, (size_t)gr * d * 4, hipMemcpyDeviceToDevice, st));
            off_rows += gr;
            big.rows += s.rows;
        }
        HAC_HIP(hipStreamSynchronize(st));
        // the fused segment's fp16 image is rebuilt from its tiles by the next prefilter search (ensure_half_image)
        for (auto &s : segs) {
            HAC_HIP(hipFree(s.buf));
            if (s.hbuf) HAC_HIP(hipFree(s.hbuf));
            if (s.rbuf) HAC_HIP(hipFree(s.rbuf));
        }
        segs.clear();
        segs.push_back(big);
        segs_dirty = true;
        return HAC_OK;
    }

    // src_dev: row-major [n][d] on this device.  Appends rows in dense numbering: every
    // segment but the last holds a multiple of 64 rows.
    int add_device_rows(const float *src_dev, int64_t n, hipStream_t st, int64_t reserve_hint = 0) {
        if (((uintptr_t)src_dev & 15) != 0) return fail(HAC_ERR_INVALID, "add: device pointer must be 16-byte aligned");
        int64_t done = 0;
        while (done < n) {
            if (segs.empty() || segs.back().rows == segs.back().cap_rows) {
                // a fresh segment must start on a group boundary of the global numbering
                if (!segs.empty() && segs.back().rows % GROUP_ROWS != 0) return fail(HAC_ERR_INVALID, "internal: partial segment");
                HAC_TRY(new_segment(std::max(n, reserve_hint) - done, st));
            }
            Segment &s = segs.back();
            if (tune.image_eager && !s.hbuf && s.rows == 0 && tune.split != 0) {
                // option "fp16_image" = "eager": a resident index that will be searched with few queries per call gets its image
                // while the rows are tiled (the add is bound by the rows' arrival, the +50 % of writes ride along) instead of by
                // its third search.  Best effort: without room the lazy route decides later.
                if (hipMalloc((void **)&s.hbuf, (size_t)s.cap_rows * d * 2) != hipSuccess) {
                    s.hbuf = nullptr;
                    (void)hipGetLastError();
                }
                s.h_rows = 0;
            }
            const int64_t m = std::min(n - done, s.cap_rows - s.rows);
            const long row0 = (long)s.rows;
            const long g_lo = row0 / GROUP_ROWS, g_hi = (row0 + m + GROUP_ROWS - 1) / GROUP_ROWS;
            tile_rows_kernel<<<dim3((unsigned)(g_hi - g_lo)), dim3(256), 0, st>>>(
                reinterpret_cast<const float4 *>(src_dev + (size_t)done * d), (long)m, K4, s.buf, s.hbuf && s.h_rows == s.rows ? s.hbuf : nullptr, row0,
                (u32 *)ws_norm.p);
            HAC_HIP(hipGetLastError());
            if (s.hbuf && s.h_rows == s.rows) s.h_rows += m;
            s.rows += m;
            done += m;
        }
        ntotal += n;
        segs_dirty = true;
        half_image_unavailable = false;   // (memory may have been freed since: the next eligible search tries again)
        row_image_unavailable = false;
        split_searches_since_add = 0;
        light_searches_since_add = 0;
        return HAC_OK;
    }

    int add_host_rows(const float *x, int64_t n) {
        DeviceGuard g(device);
        const int64_t chunk_rows = std::max<int64_t>(GROUP_ROWS, (int64_t)(64u << 20) / (d * 4) / GROUP_ROWS * GROUP_ROWS);  // ~64 MiB
        const size_t chunk_bytes = (size_t)chunk_rows * d * 4;
        if (h_stage_bytes < chunk_bytes) {
            for (int i = 0; i < 2; ++i) {
                if (h_stage[i]) (void)hipHostFree(h_stage[i]);
                h_stage[i] = nullptr;
                hipError_t e = hipHostMalloc(&h_stage[i], chunk_bytes, hipHostMallocDefault);
                if (e != hipSuccess) return fail(HAC_ERR_OOM, "hipHostMalloc(%zu) failed: %s", chunk_bytes, hipGetErrorString(e));
            }
            h_stage_bytes = chunk_bytes;
        }
        for (int i = 0; i < 2; ++i) HAC_TRY(ws_stage[i].reserve(chunk_bytes));
        int64_t done = 0;
        int slot = 0;
        bool used[2] = {false, false};
        while (done < n) {
            const int64_t m = std::min(chunk_rows, n - done);
            if (used[slot]) HAC_HIP(hipEventSynchronize(stage_ev[slot]));
            {   // pageable -> pinned staging: one core copies ~12 GB/s, the link takes ~50: split the chunk over a few threads
                const size_t bytes = (size_t)m * d * 4;
                const char *src = reinterpret_cast<const char *>(x + (size_t)done * d);
                char *dst = static_cast<char *>(h_stage[slot]);
                const int nt = bytes >= (8u << 20) ? 4 : 1;
                const size_t part = (bytes / nt + 4095) & ~(size_t)4095;
                std::vector<std::thread> pool;
                for (int t = 1; t < nt; ++t) {
                    const size_t o = std::min(bytes, (size_t)t * part), e = std::min(bytes, (size_t)(t + 1) * part);
                    if (e > o) pool.emplace_back([=] { std::memcpy(dst + o, src + o, e - o); });
                }
                std::memcpy(dst, src, std::min(bytes, part));
                for (auto &th : pool) th.join();
            }
            HAC_HIP(hipMemcpyAsync(ws_stage[slot].p, h_stage[slot], (size_t)m * d * 4, hipMemcpyHostToDevice, stream));
            // one new segment per add() at most: size it for everything that is left
            HAC_TRY(add_device_rows((const float *)ws_stage[slot].p, m, stream, n - done));
            HAC_HIP(hipEventRecord(stage_ev[slot], stream));
            used[slot] = true;
            slot ^= 1;
            done += m;
        }
        HAC_HIP(hipStreamSynchronize(stream));
        return HAC_OK;
    }

    int upload_segs(hipStream_t st) {
        if (!segs_dirty) return HAC_OK;
        if (stream_is_capturing(st))
            return fail(HAC_ERR_INVALID, "a search that is being captured into a graph found this index's segment table changed since its last search "
                        "(add / reset / a dropped image): run one search outside the capture first (include/haconvdr.h: warm up before capturing)");
        // h_segs may still be the source of an earlier in-flight copy on another stream
        HAC_HIP(hipStreamSynchronize(st));
        SegDesc *h = h_segs;
        u32 g = 0;
        int n = 0;
        for (auto &s : segs) {
            if (s.rows == 0) continue;
            h[n].ptr = s.buf;
            h[n].himg = s.hbuf;
            h_rimg[n] = (s.rbuf && s.r_rows == s.rows) ? s.rbuf : nullptr;   // (a table of its own: SegDesc is cached in the scan kernels' LDS, which is full)
            h[n].gstart = g;
            h[n].pad_ = 0;
            g += (u32)((s.rows + GROUP_ROWS - 1) / GROUP_ROWS);
            ++n;
        }
        if (n) HAC_HIP(hipMemcpyAsync(d_segs, h, sizeof(SegDesc) * n, hipMemcpyHostToDevice, st));
        if (n) HAC_HIP(hipMemcpyAsync(d_rimg, h_rimg, sizeof(float4 *) * n, hipMemcpyHostToDevice, st));
        nseg_live = n;
        segs_dirty = false;
        return HAC_OK;
    }
    int nseg_live = 0;
    char last_plan[320] = "none";
    // a device-decided prefilter search leaves its fallback count and err / bound on the device: plan() completes the text
    char plan_head[200] = "";
    // The status words of a device-decided search reach the host through a pinned copy enqueued on the CALLER's stream and
    // a library-owned event behind it: plan() waits for that event only -- never for the caller's stream, which may be gone
    // or capturing by the time someone asks.  Back-to-back device-decided searches each take their own (words, event) slot of
    // a small ring (with one slot, a search issued before the previous one had completed overwrote its words and
    // that search's fallback count never reached split_fallback_queries); when every slot is pending the new search's words are not collected.
    static constexpr int PLAN_RING = 32;
    hipEvent_t ev_plan[PLAN_RING] = {};
    u32 *h_plan = nullptr;     // pinned [PLAN_RING][2]: (queries that fell back, max err / bound as float bits)
    bool slot_pending[PLAN_RING] = {};
    int64_t slot_nq[PLAN_RING] = {};
    int plan_head_slot = 0;    // next slot to hand out; pending slots are the ones before it, oldest first
    int plan_text_slot = -1;   // the slot whose search last_plan describes (-1: last_plan is complete as it stands)
    // true when slot i has been retired (its count added); wait = block until it has
    bool plan_retire(int i, bool wait) {
        if (!slot_pending[i]) return true;
        hipError_t e = wait ? hipEventSynchronize(ev_plan[i]) : hipEventQuery(ev_plan[i]);
        if (e == hipErrorNotReady) {
            (void)hipGetLastError();
            return false;
        }
        // (an error here = the search was captured into a graph: the event was never recorded for real and cannot be waited
        // for; the pinned words then are those of the last replay that has completed -- synchronize the stream you replay on)
        if (e != hipSuccess) (void)hipGetLastError();
        if (i == plan_text_slot) {
            float maxratio;
            std::memcpy(&maxratio, &h_plan[2 * i + 1], 4);
            snprintf(last_plan, sizeof last_plan, "%s fallback=%u/%lld err/bound=%.3g decided=device%s", plan_head, h_plan[2 * i], (long long)slot_nq[i],
                     (double)maxratio, e == hipSuccess ? "" : " (captured: as of the last completed replay)");
            plan_text_slot = -1;
        }
        split_fallback_queries += h_plan[2 * i];
        slot_pending[i] = false;
        return true;
    }
    // (called at the start of the next device-decided search too, so that the fallback count of a search nobody polled is
    // collected as soon as it has completed)
    void plan_collect(bool wait) {
        for (int n = 0; n < PLAN_RING; ++n) {
            const int i = (plan_head_slot + n) % PLAN_RING;   // oldest first
            if (!plan_retire(i, wait)) continue;              // (callers may use several streams: a younger search can be done before an older one)
        }
    }
    // a slot for the search being enqueued, or -1: every slot still belongs to a search in flight (the *_device entry points never
    // wait on the host: that search's status words are then simply not collected -- its results do not depend on them)
    long long plan_uncollected = 0;
    int plan_slot() {
        const int i = plan_head_slot;
        if (slot_pending[i] && !plan_retire(i, false)) {
            ++plan_uncollected;
            return -1;
        }
        plan_head_slot = (i + 1) % PLAN_RING;
        return i;
    }
    bool plan_pending() const {
        for (bool b : slot_pending)
            if (b) return true;
        return false;
    }
    const char *plan() {
        if (plan_pending()) {
            DeviceGuard g(device);
            plan_collect(true);
        }
        return last_plan;
    }

    struct Plan {
        int kind;  // 0: scan16 (<=16 queries per workgroup, Q resident in LDS)   1: scanq<NT,W>
        int NT, W;
        int QT, C, n_qtiles, P;
        size_t lds_scan;
    };

    static size_t scanq_lds(int NQ, int C) { return (size_t)NQ * (2 * 16 * 16 + (size_t)C * 8 + 8) + 16; }

    int make_plan(int64_t nq, int k, u32 n_items, Plan &pl, bool want16 = false) const {
        pl.kind = 0;
        pl.NT = 0;
        pl.W = SCAN_WAVES;
        // many queries: GEMM-shaped kernel, NQ = 32*NT queries per workgroup
        if (nq > 16 && K4 % 16 == 0 && !want16 && !tune.force_scan16) {
            const int C2 = (int)std::max<u32>(32u, next_pow2((u32)k + 1u));
            int best_nt = 0;
            int64_t best_pad = 0;
            for (int nt = 1; nt <= 4; ++nt) {
                if (scanq_lds(32 * nt, C2) > LDS_LIMIT) break;
                const int64_t pad = (nq + 32 * nt - 1) / (32 * nt) * (32 * nt);
                if (best_nt == 0 || pad <= best_pad) {
                    best_nt = nt;
                    best_pad = pad;
                }
            }
            if (tune.scanq_nt >= 1 && tune.scanq_nt <= 4 && scanq_lds(32 * tune.scanq_nt, C2) <= LDS_LIMIT)
                best_nt = tune.scanq_nt;   // tuning experiments only
            if (best_nt) {
                pl.kind = 1;
                pl.NT = best_nt;
                pl.W = tune.scanq_waves;
                pl.QT = 32 * best_nt;
                pl.C = C2;
                pl.lds_scan = scanq_lds(pl.QT, C2);
            }
        }
        if (pl.kind == 0) {
            pl.C = (int)next_pow2((u32)k + 64u * SCAN_WAVES);
            const size_t per_q = (size_t)K4 * 16 + (size_t)pl.C * 8;
            const size_t fixed = 16 * 4 + 16 * 4;
            int qt = (int)std::min<size_t>(16, (LDS_LIMIT - fixed) / per_q);
            if (qt < 1) return fail(HAC_ERR_UNSUPPORTED, "k=%d with d=%d does not fit the LDS candidate buffers", k, d);
            qt = (int)std::min<int64_t>(qt, nq);
            pl.QT = qt;
            pl.lds_scan = per_q * qt + fixed;
        }
        pl.n_qtiles = (int)((nq + pl.QT - 1) / pl.QT);
        // resident workgroups: LDS- and wave-limited
        int per_cu = (int)std::min<size_t>(LDS_LIMIT / pl.lds_scan, 32 / pl.W);
        per_cu = std::max(1, std::min(per_cu, pl.kind == 1 ? 2 : 4));
        const long resident = (long)n_cu * per_cu;
        long P = std::max<long>(1, resident / pl.n_qtiles);
        if (P >= 8 && !tune.no_p8) P = P / 8 * 8;  // same-row workgroups of different query tiles share an XCD (L2)
        const long maxP = (n_items + pl.W - 1) / pl.W;
        P = std::max<long>(1, std::min(P, maxP));
        pl.P = (int)P;
        return HAC_OK;
    }

    template <int NT, int W>
    static void launch_scanq(const ScanArgs &a, int P, int n_qtiles, size_t lds, hipStream_t st) {
        scanq_kernel<NT, W><<<dim3((unsigned)P, (unsigned)n_qtiles), dim3(W * 64), lds, st>>>(a);
    }

    int run_scan(const Plan &pl, const float *q_dev, int64_t nq, int k, u32 g_first, u32 g_step, u32 n_items,
                 const float *thr_init, u32 pos_base, int P, hipStream_t st, bool timed, const int *nq_dev = nullptr, bool cleared = false) {
        ScanArgs a;
        a.nq_dev = nq_dev;
        a.segs = d_segs;
        a.nseg = nseg_live;
        a.q = reinterpret_cast<const float4 *>(q_dev);
        a.qt = reinterpret_cast<const float4 *>(ws_qt.p);
        a.nq = (int)nq;
        a.K4 = K4;
        a.k = k;
        a.C = pl.C;
        a.QT = pl.QT;
        a.n_rows = (long)ntotal;
        a.g_first = g_first;
        a.g_step = g_step;
        a.n_items = n_items;
        a.thr_init = thr_init;
        a.thr_glob = thrglob();
        a.partial = (u64 *)ws_partial.p;
        a.partial_cnt = (u32 *)ws_pcnt.p;
        a.pos_base = pos_base;
        if (!cleared) {     // (the device-decided fallback's first chunk: gather_rows_kernel has cleared both)
            HAC_HIP(hipMemsetAsync(ws_pcnt.p, 0, (size_t)nq * 4, st));
            HAC_TRY(clear_thrglob((size_t)pl.n_qtiles * pl.QT, st));
        }
        if (timed) {
            if (ev_used == ev_pool.size()) {
                hipEvent_t a0, a1;
                HAC_HIP(hipEventCreate(&a0));
                HAC_HIP(hipEventCreate(&a1));
                ev_pool.emplace_back(a0, a1);
            }
            HAC_HIP(hipEventRecord(ev_pool[ev_used].first, st));
        }
        if (pl.kind == 0) {
            scan16_kernel<<<dim3((unsigned)P, (unsigned)pl.n_qtiles), dim3(SCAN_WAVES * 64), pl.lds_scan, st>>>(a);
        } else {
            const int key = pl.NT * 10 + pl.W;
            switch (key) {
                case 14: launch_scanq<1, 4>(a, P, pl.n_qtiles, pl.lds_scan, st); break;
                case 24: launch_scanq<2, 4>(a, P, pl.n_qtiles, pl.lds_scan, st); break;
                case 34: launch_scanq<3, 4>(a, P, pl.n_qtiles, pl.lds_scan, st); break;
                case 44: launch_scanq<4, 4>(a, P, pl.n_qtiles, pl.lds_scan, st); break;
                case 18: launch_scanq<1, 8>(a, P, pl.n_qtiles, pl.lds_scan, st); break;
                case 28: launch_scanq<2, 8>(a, P, pl.n_qtiles, pl.lds_scan, st); break;
                case 38: launch_scanq<3, 8>(a, P, pl.n_qtiles, pl.lds_scan, st); break;
                case 48: launch_scanq<4, 8>(a, P, pl.n_qtiles, pl.lds_scan, st); break;
                default: return fail(HAC_ERR_UNSUPPORTED, "internal: no scanq<%d,%d>", pl.NT, pl.W);
            }
        }
        HAC_HIP(hipGetLastError());
        if (timed) {
            HAC_HIP(hipEventRecord(ev_pool[ev_used].second, st));
            ++ev_used;
        }
        return HAC_OK;
    }

    // keys_out: device u64 [nq][k]; the exact fp32 kernels only.  nq_dev (optional): the number of queries present lives on
    // the device and nq is the capacity everything is sized for (the prefilter's device-decided fallback): no threshold
    // seeding (fewer launches on a path that is empty in the common case), the scan's workgroups are shared among the live
    // query tiles (vgrid), and rows of keys_out beyond *nq_dev are left alone.
    int search_keys_exact(const float *q_dev, int64_t nq, int k, u64 *keys_out, u32 pos_base, hipStream_t st, const int *nq_dev = nullptr, bool cleared = false) {
        if (nq == 0) return HAC_OK;
        if (((uintptr_t)q_dev & 15) != 0) return fail(HAC_ERR_INVALID, "search: query pointer must be 16-byte aligned");
        if ((uint64_t)pos_base + (uint64_t)ntotal > 0xFFFFFFFFull) return fail(HAC_ERR_UNSUPPORTED, "row positions exceed 32 bits");
        if (ntotal == 0) {
            HAC_HIP(hipMemsetAsync(keys_out, 0, (size_t)nq * k * 8, st));
            return HAC_OK;
        }
        HAC_TRY(upload_segs(st));
        const u32 G = (u32)((ntotal + GROUP_ROWS - 1) / GROUP_ROWS);
        Plan pl;
        HAC_TRY(make_plan(nq, k, G, pl));
        // device-side count: a live tile may own every workgroup of the grid; QT * (P * n_qtiles) * k keys bound any split
        HAC_TRY(ws_partial.reserve(nq_dev ? (size_t)pl.QT * pl.P * pl.n_qtiles * k * 8 : (size_t)nq * pl.P * k * 8));
        HAC_TRY(ws_pcnt.reserve((size_t)nq * 4));
        HAC_TRY(ws_thrglob.reserve(((size_t)pl.n_qtiles * pl.QT + THR_CTL_WORDS) * 4));
        if (pl.kind == 1) {
            const long total = (long)pl.n_qtiles * K4 * pl.QT;
            HAC_TRY(ws_qt.reserve((size_t)total * 16));
            retile_queries_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(
                reinterpret_cast<const float4 *>(q_dev), (int)nq, K4, pl.QT, pl.n_qtiles, (float4 *)ws_qt.p);
            HAC_HIP(hipGetLastError());
        }
        const float *thr_init = nullptr;
        // Threshold seeding: exact top-k of an evenly strided sample of groups gives a
        // lower bound of every query's final k-th score; it only filters, never decides.
        // sample ~1.6 % of the groups, at least 64 and enough for 2k maxima (4 per group)
        const u32 n_sample = std::max<u32>(std::max<u32>(64u, G / 64u), ((u32)k + 1u) / 2u);
        if (G >= 4u * n_sample && !nq_dev) {
            const u32 S = 4u * n_sample;                             // four maxima per sample group
            HAC_TRY(ws_seedkeys.reserve((size_t)nq * S * 4));        // group maxima [nq][S]
            HAC_TRY(ws_thr.reserve((size_t)nq * 4));
            ScanArgs a{};
            a.segs = d_segs;
            a.nseg = nseg_live;
            a.q = reinterpret_cast<const float4 *>(q_dev);
            a.nq = (int)nq;
            a.K4 = K4;
            a.n_rows = (long)ntotal;
            a.g_first = 0;
            a.g_step = G / n_sample;
            a.n_items = n_sample;
            const int qt16 = (int)((nq + 15) / 16);
            sample_scores_kernel<<<dim3((n_sample + SCAN_WAVES - 1) / SCAN_WAVES, (unsigned)qt16), dim3(SCAN_WAVES * 64),
                                   (size_t)K4 * 16 * 16, st>>>(a, (float *)ws_seedkeys.p, S);
            HAC_HIP(hipGetLastError());
            kth_select_kernel<<<dim3((unsigned)nq), dim3(256), 0, st>>>((const float *)ws_seedkeys.p, S, k, (float *)ws_thr.p);
            HAC_HIP(hipGetLastError());
            thr_init = (const float *)ws_thr.p;
        }
        if (nq_dev) {
            // (the caller's plan stays: this is its fallback)
        } else if (pl.kind == 1)
            snprintf(last_plan, sizeof last_plan, "scanq_kernel<NT=%d,W=%d> grid=(%d,%d) NQ=%d C=%d lds=%zu seed=%d", pl.NT, pl.W, pl.P,
                     pl.n_qtiles, pl.QT, pl.C, pl.lds_scan, thr_init ? 1 : 0);
        else
            snprintf(last_plan, sizeof last_plan, "scan16_kernel<W=%d> grid=(%d,%d) QT=%d C=%d lds=%zu seed=%d", SCAN_WAVES, pl.P,
                     pl.n_qtiles, pl.QT, pl.C, pl.lds_scan, thr_init ? 1 : 0);
        if (!nq_dev) plan_text_slot = -1;   // (last_plan is this search's text; older device-decided searches still deliver their counts)
        HAC_TRY(run_scan(pl, q_dev, nq, k, 0, 1, G, thr_init, pos_base, pl.P, st, profiling, nq_dev, cleared));
        // the workgroups' survivors sit densely per query: radix select of the k best, one sort of k keys
        const int np2 = (int)next_pow2((u32)k);
        select_keys_kernel<<<dim3((unsigned)nq), dim3(256), (size_t)np2 * 8, st>>>((const u64 *)ws_partial.p, (size_t)pl.P * k,
                                                                                   (const u32 *)ws_pcnt.p, (u32)((size_t)pl.P * k), k, np2,
                                                                                   keys_out, nullptr, nq_dev, (int)nq, (u32)(pl.P * pl.n_qtiles),
                                                                                   (u32)pl.QT, thrglob(), (u32 *)ws_err.p);
        HAC_HIP(hipGetLastError());
        return HAC_OK;
    }

    // ---- split-bf16 prefilter + exact rescoring (scan_split.inc): same results, ~3x the query rate when the
    // exact kernels are bound by the fp32 matrix rate.  Worth its fixed cost only for many (query, row) pairs.
    static constexpr int SPLIT_K2 = 256, SPLIT_C2 = 512;
    // Who takes the prefilter.  It streams the fp16 image -- half the bytes of the fp32 tiles -- so once that image exists it beats
    // the exact kernels at EVERY query count on a corpus large enough to hide its ~0.18 ms of launches (tools/nq_sweep.py,
    // profiles/r05_nq_sweep.txt: 6.75M rows 2.1-2.4 ms against 3.4-6.5 ms for 1 ... 64 queries; 1M rows 0.49-0.61 against 0.62-1.06;
    // break-even near 500k rows for <= 16 queries, 200k for 64, 100k for 256, 50k for 1000; never at 20k).  Three cases:
    //   * nq >= 48 and nq x rows >= 1e8: the image pays for itself within one or two searches -> built at once (rounds 2-4);
    //   * otherwise faster by the table above and an image is already there -> used (and kept current across adds);
    //   * otherwise faster but no image yet (+50 % of the corpus in HBM, one pass over it): an index that keeps being searched
    //     builds it with its THIRD such search after the last add / reset; an index searched once or twice per block never does.
    bool split_supported(int k) const {
        // d % 64 == 0 (whole query slices) and at least 12 k-steps per row: the corpus ring runs up to 9 steps ahead and
        // may reach into the NEXT group only
        return !(K4 % 16 != 0 || K4 < 48 || d > HAC_MAX_D || k > SPLIT_K2 - 64 || ntotal < SPLIT_K2);
    }
    bool half_image_present() const {   // some live segment has one: the rest is brought up to date by ensure_half_image (the new rows only)
        for (auto &s : segs)
            if (s.rows > 0 && s.hbuf && s.h_rows > 0) return true;
        return false;
    }
    static constexpr int LIGHT_SEARCHES_BEFORE_IMAGE = 2;
    int light_searches_since_add = 0;
    // (not a pure predicate: a light search of an index without an image counts towards building one)
    bool half_image_current() const {   // every live segment's image covers all of its rows: a search can use it as it is
        bool any = false;
        for (auto &s : segs) {
            if (s.rows == 0) continue;
            if (!(s.hbuf && s.h_rows == s.rows)) return false;
            any = true;
        }
        return any;
    }
    bool decide_prefilter(int64_t nq, int k, bool capturing = false) {
        if (tune.split == 0) return false;   // 0: never, 1: whenever supported (tests), -1: by size
        if (!split_supported(k)) return false;
        // a search that is being captured into a graph neither allocates nor uploads nor synchronizes (stream_is_capturing): it takes
        // the prefilter only over an image that is complete and a segment table that is on the device -- whatever split says and
        // however many pairs there are -- and the exact kernels (same bits) otherwise
        if (capturing && !(half_image_current() && !segs_dirty)) return false;
        if (tune.split == 1) return true;
        const double pairs = (double)nq * (double)ntotal;
        if (nq >= 48 && pairs >= 1.0e8) return true;
        const bool faster = ntotal >= 750000 || (nq >= 17 && ntotal >= 500000) || (nq >= 48 && pairs >= 2.5e7 && ntotal >= 40000);
        if (!faster) return false;
        if (half_image_present()) return true;
        if (capturing) return false;          // (a captured search builds nothing: see stream_is_capturing)
        return light_searches_since_add++ >= LIGHT_SEARCHES_BEFORE_IMAGE;
    }

    int fb_reserve(size_t words) {
        if (words <= h_fb_words) return HAC_OK;
        if (h_fb) (void)hipHostFree(h_fb);
        h_fb = nullptr;
        h_fb_words = 0;
        hipError_t e = hipHostMalloc((void **)&h_fb, words * 4 * 2, hipHostMallocDefault);
        if (e != hipSuccess) return fail(HAC_ERR_OOM, "hipHostMalloc(%zu) failed: %s", words * 8, hipGetErrorString(e));
        h_fb_words = words * 2;
        return HAC_OK;
    }

    // Cascade: one fp16 product per score (|s~ - s| ~ 1.2e-3 |q||x| proven) decides every query whose candidate
    // list it can certify; if many fail, three products (hi/lo split, ~2.8e-4 |q||x|, three times the MFMA work)
    // retry those; whatever is left goes to the exact fp32 kernels.  level 0 -> terms 1, level 1 -> terms 3.
    //
    // Queries go through in chunks of at most 1024 (four 256-query tiles x 64 row streams fill the chip): the
    // rescoring of chunk c (HBM gathers, no matrix work) runs on a second stream under the scan of chunk c+1.
    static constexpr int64_t SPLIT_CHUNK = 1024;
    int search_keys_split(const float *q_dev, int64_t nq, int k, u64 *keys_out, u32 pos_base, hipStream_t st, int level = 0,
                          bool device_decides = false) {
        const int K2 = SPLIT_K2, C2 = SPLIT_C2;
        const int terms = level == 0 ? tune.split_terms : 3;   // split_terms = 3: tests pin the first level
        plan_collect(false);                                   // an earlier device-decided search nobody asked about
        HAC_TRY(ensure_half_image(st));
        HAC_TRY(ensure_row_image(st));
        HAC_TRY(upload_segs(st));
        const u32 G = (u32)((ntotal + GROUP_ROWS - 1) / GROUP_ROWS);
        const int64_t chunk = std::min<int64_t>(nq, SPLIT_CHUNK);
        const int n_qtiles_max = (int)((chunk + SH_NQ - 1) / SH_NQ);
        const size_t lds = terms == 3 ? ShCfg<3>::LDS : ShCfg<1>::LDS;
        long Pmax = std::max<long>(1, n_cu);
        HAC_TRY(ws_qsplit.reserve((size_t)n_qtiles_max * SH_NQ * d * 2 * (terms == 3 ? 2 : 1)));
        HAC_TRY(ws_delta.reserve((size_t)(nq + SH_NQ) * 4));
        HAC_TRY(ws_cand.reserve((size_t)Pmax * SH_NQ * C2 * 8));          // P * n_qtiles <= n_cu workgroups
        // (n queries of a chunk in ceil(n / SH_NQ) tiles with P <= n_cu / tiles row streams each: n * P <= SH_NQ * n_cu)
        HAC_TRY(ws_partial.reserve((size_t)std::min<int64_t>(chunk, SH_NQ) * Pmax * K2 * 8 * MAX_PASSES));
        HAC_TRY(ws_pcnt.reserve((size_t)chunk * 4));
        HAC_TRY(ws_thrglob.reserve(((size_t)n_qtiles_max * SH_NQ + THR_CTL_WORDS) * 4));
        HAC_TRY(ws_akeys.reserve((size_t)nq * K2 * 8));
        HAC_TRY(ws_fail.reserve((size_t)nq * 4));
        HAC_TRY(ws_stat.reserve(16));
        HAC_TRY(ws_thr.reserve((size_t)chunk * 4));
        HAC_TRY(fb_reserve((size_t)nq + 8));
        // (ws_stat: cleared by the first chunk's split_queries_kernel)

        ScanArgs a{};
        a.segs = d_segs;
        a.nseg = nseg_live;
        a.K4 = K4;
        a.n_rows = (long)ntotal;
        a.pos_base = pos_base;
        a.k = K2;
        a.g_first = 0;
        a.g_step = 1;
        a.thr_glob = thrglob();
        a.partial = (u64 *)ws_partial.p;
        a.partial_cnt = (u32 *)ws_pcnt.p;
        SplitArgs sp{};
        sp.qsplit = (const u32x4 *)ws_qsplit.p;
        sp.cand = (u64 *)ws_cand.p;
        sp.C2 = C2;
        sp.K2 = K2;
        sp.thr_is_approx = 1;
        long P_last = 0;
        int n_qtiles_last = 0, seeded = 0, n_chunks = 0, passes_last = 1;
        int qt_act_last = 16;
        for (int64_t off = 0; off < nq; off += chunk, ++n_chunks) {
            const int64_t n = std::min<int64_t>(chunk, nq - off);
            const float *qc = q_dev + (size_t)off * d;
            const int n_qtiles = (int)((n + SH_NQ - 1) / SH_NQ);
            const int64_t nq_pad = (int64_t)n_qtiles * SH_NQ;
            long P = std::max<long>(1, n_cu / n_qtiles);
            if (P >= 8) P = P / 8 * 8;  // same-row workgroups of different query tiles share an XCD (L2)
            P = std::max<long>(1, std::min<long>(P, (G + SH_GPR - 1) / SH_GPR));
            const long pstride = (long)P * K2 * MAX_PASSES;   // a workgroup flushes at most K2 survivors per query and pass
            float *delta_c = (float *)ws_delta.p + off;
            u64 *akeys_c = (u64 *)ws_akeys.p + (size_t)off * K2;
            split_queries_kernel<<<dim3((unsigned)nq_pad), dim3(192), 0, st>>>(reinterpret_cast<const float4 *>(qc), (int)n, K4, terms,
                                                                              (const u32 *)ws_norm.p, (h16 *)ws_qsplit.p, delta_c, (u32 *)ws_pcnt.p,
                                                                              (u32 *)ws_thrglob.p, (u32)std::max(0, tune.debug_max_pass),
                                                                              off == 0 ? (u32 *)ws_stat.p : nullptr);
            HAC_HIP(hipGetLastError());
            a.q = reinterpret_cast<const float4 *>(qc);
            a.nq = (int)n;
            sp.delta = delta_c;
            sp.pstride = pstride;
            // (ws_pcnt, the thresholds and their control words: cleared by split_queries_kernel)
            if (profiling) {
                if (ev_used == ev_pool.size()) {
                    hipEvent_t a0, a1;
                    HAC_HIP(hipEventCreate(&a0));
                    HAC_HIP(hipEventCreate(&a1));
                    ev_pool.emplace_back(a0, a1);
                }
                HAC_HIP(hipEventRecord(ev_pool[ev_used].first, st));
            }
            // Seeding pass: the head of the corpus is scored once just for its per-quarter maxima; their K2-th largest
            // opens the real pass over ALL rows with thresholds that only ~K2 * G / GA rows per query pass.  Without sharp
            // opening thresholds list compactions (sorts) cost as much as half the MFMA work; the seeding pass itself (and
            // the selection over its 4 GA maxima per query) costs in proportion to GA: a sixteenth of the corpus up to
            // 1M rows, then ~14 sqrt(G) groups (measured optimum at 6.75M / 10M / 25M rows: 4.1k / 5.1k / 10k groups;
            // a sixteenth of 25M rows cost 1.5 ms more per 1000-query search, 2k groups 11 ms more).
            const dim3 grid((unsigned)P, (unsigned)n_qtiles), blk(SH_W * 64);
            // one tile of at most 128 / 64 real queries: the instantiations without the empty 16-query tiles' matrix work (scanh_kernel, QT_ACT)
            const bool few = terms == 1 && SH_M16 && n_qtiles == 1 && !tune.no_halfq;
            const int qt_act = few && n <= SH_NQ / 4 ? 4 : few && n <= SH_NQ / 2 ? 8 : 16;
            const u32 round_groups = (u32)P * SH_GPR;
            // (a scan that will be cut into passes refreshes its thresholds after ~6k groups anyway: 1024 groups of seeding do)
            u32 probe_bounds[MAX_PASSES + 1];
            const bool multipass = scan_passes(G, round_groups, true, probe_bounds) > 1;
            const u32 seed_cap = tune.seed_groups_max > 0 ? (u32)tune.seed_groups_max : multipass ? 1024u : (u32)(14.0 * std::sqrt((double)G));
            u32 GA = std::min<u32>(G, (std::max<u32>(std::min<u32>(G / 16u, seed_cap), 768u) + round_groups - 1u) / round_groups * round_groups);
            const float *thr_init = nullptr;
            if ((size_t)4 * GA >= (size_t)K2) {
                const u32 S = 4u * GA;
                HAC_TRY(ws_seedkeys.reserve((size_t)chunk * S * 4));
                a.n_items = GA;
                a.thr_init = nullptr;
                sp.maxima = (float *)ws_seedkeys.p;
                if (terms == 3) scanh_kernel<3, true><<<grid, blk, lds, st>>>(a, sp);
                else if (qt_act == 4) scanh_kernel<1, true, 4><<<grid, blk, lds, st>>>(a, sp);
                else if (qt_act == 8) scanh_kernel<1, true, 8><<<grid, blk, lds, st>>>(a, sp);
                else scanh_kernel<1, true><<<grid, blk, lds, st>>>(a, sp);
                HAC_HIP(hipGetLastError());
                kth_select_kernel<<<dim3((unsigned)n), dim3(256), 0, st>>>((const float *)ws_seedkeys.p, S, K2, (float *)ws_thr.p);
                HAC_HIP(hipGetLastError());
                thr_init = (const float *)ws_thr.p;
            }
            // The scan itself, in up to three passes over consecutive row ranges (big corpora, seeded thresholds).  A workgroup's own
            // lists never reach their compaction mark (~100 candidates per query and row stream over a whole 25M-row scan), so
            // within one launch the seeded threshold is all a query ever has: 0.07 % of the pairs pass it where K2 / rows would
            // do, and parking + appending those candidates is ~12 % of the kernel.  Between passes the K2-th best s~ of everything
            // found SO FAR (all row streams together: select_keys_kernel over the survivors flushed by the passes before) is a
            // valid, much sharper bound: after 8 % of the rows ~3x fewer pairs pass, after 30 % ~8x fewer.
            u32 bounds[MAX_PASSES + 1];
            const int n_pass = scan_passes(G, round_groups, thr_init != nullptr, bounds);
            a.thr_init = thr_init;
            for (int ps = 0; ps < n_pass; ++ps) {
                a.g_first = bounds[ps];
                a.n_items = bounds[ps + 1] - bounds[ps];
                if (terms == 3) scanh_kernel<3, false><<<grid, blk, lds, st>>>(a, sp);
                else if (qt_act == 4) scanh_kernel<1, false, 4><<<grid, blk, lds, st>>>(a, sp);
                else if (qt_act == 8) scanh_kernel<1, false, 8><<<grid, blk, lds, st>>>(a, sp);
                else scanh_kernel<1, false><<<grid, blk, lds, st>>>(a, sp);
                HAC_HIP(hipGetLastError());
                if (ps + 1 < n_pass) {   // refresh: ws_thr[q] = max(ws_thr[q], K2-th best s~ so far)
                    select_keys_kernel<<<dim3((unsigned)n), dim3(256), (size_t)K2 * 8, st>>>((const u64 *)ws_partial.p, (size_t)pstride,
                                                                                             (const u32 *)ws_pcnt.p, (u32)pstride, K2, K2, nullptr, (float *)ws_thr.p,
                                                                                             nullptr, 0, 0, 1, nullptr, nullptr, true);
                    HAC_HIP(hipGetLastError());
                }
            }
            a.g_first = 0;
            passes_last = n_pass;
            qt_act_last = qt_act;
            if (profiling) {
                HAC_HIP(hipEventRecord(ev_pool[ev_used].second, st));
                ++ev_used;
            }
            // exact top-K2 by approximate score over all workgroups' survivors
            select_keys_kernel<<<dim3((unsigned)n), dim3(256), (size_t)K2 * 8, st>>>((const u64 *)ws_partial.p, (size_t)pstride,
                                                                                     (const u32 *)ws_pcnt.p, (u32)pstride, K2, K2, akeys_c, nullptr,
                                                                                     nullptr, 0, 0, 1, thrglob(), (u32 *)ws_err.p);
            HAC_HIP(hipGetLastError());
            // rescoring + certificate of this chunk on the second stream (reads only akeys, delta, the queries and
            // the corpus; everything the next chunk's scan reuses is already consumed)
            HAC_HIP(hipEventRecord(ev_chunk[n_chunks & 1], st));
            HAC_HIP(hipStreamWaitEvent(stream2, ev_chunk[n_chunks & 1], 0));
            rescore_kernel<<<dim3((unsigned)n), dim3(256), 0, stream2>>>(a, d_rimg, akeys_c, delta_c, K2, k, keys_out + (size_t)off * k,
                                                                        (u32 *)ws_fail.p + off, (u32 *)ws_stat.p);
            HAC_HIP(hipGetLastError());
            P_last = P;
            n_qtiles_last = n_qtiles;
            seeded = thr_init ? 1 : 0;
        }
        HAC_HIP(hipEventRecord(ev_tail, stream2));
        HAC_HIP(hipStreamWaitEvent(st, ev_tail, 0));
        if (device_decides) {
            // No read-back: the failed queries (none, in the common case) are compacted into a list, searched again by the
            // exact fp32 kernels and scattered back into place, every launch sized for all nq and cut down on the device by
            // the count the certificates left in ws_stat[0].  The stream is never synchronized; the plan text (fallback
            // count, err / bound) is completed when hac_index_last_plan asks for it.
            GrowBuf &fbidx = ws_fbidx[0], &fbq = ws_fbq[0], &fbkeys = ws_fbkeys[0];
            const int n_fchunks = (int)((nq + QUERY_CHUNK - 1) / QUERY_CHUNK);
            HAC_TRY(fbidx.reserve(((size_t)nq + n_fchunks) * 4));
            HAC_TRY(fbq.reserve((size_t)nq * d * 4));
            HAC_TRY(fbkeys.reserve((size_t)nq * k * 8));
            int *chunk_cnt = (int *)fbidx.p + nq;
            compact_failed_kernel<<<dim3(1), dim3(256), 0, st>>>((const u32 *)ws_fail.p, (int)nq, (int *)fbidx.p, chunk_cnt, n_fchunks, (int)QUERY_CHUNK);
            HAC_HIP(hipGetLastError());
            const int *nf_dev = (const int *)ws_stat.p;
            // (the first fallback chunk's survivor counts and thresholds are cleared by gather_rows_kernel: sized from that chunk's plan)
            Plan pl0;
            const int64_t n0 = std::min<int64_t>(QUERY_CHUNK, nq);
            HAC_TRY(make_plan(n0, k, G, pl0));
            const int n_thr0 = pl0.n_qtiles * pl0.QT + THR_CTL_WORDS;
            HAC_TRY(ws_pcnt.reserve((size_t)n0 * 4));
            HAC_TRY(ws_thrglob.reserve((size_t)n_thr0 * 4));
            gather_rows_kernel<<<dim3((unsigned)((std::max<long>((long)nq * K4, n_thr0) + 255) / 256)), dim3(256), 0, st>>>(
                reinterpret_cast<const float4 *>(q_dev), (const int *)fbidx.p, (int)nq, K4, (float4 *)fbq.p, nf_dev, (u32 *)ws_pcnt.p, (int)n0,
                (u32 *)ws_thrglob.p, n_thr0, (u32)std::max(0, tune.debug_max_pass));
            HAC_HIP(hipGetLastError());
            const bool prof = profiling;
            profiling = false;   // timed kernels of a search: the prefilter's scans
            int rc = HAC_OK;
            for (int c = 0; c < n_fchunks && rc == HAC_OK; ++c) {
                const int64_t off = (int64_t)c * QUERY_CHUNK, n = std::min<int64_t>(QUERY_CHUNK, nq - off);
                rc = search_keys_exact((const float *)fbq.p + (size_t)off * d, n, k, (u64 *)fbkeys.p + (size_t)off * k, pos_base, st, chunk_cnt + c, c == 0);
            }
            profiling = prof;
            HAC_TRY(rc);
            scatter_keys_kernel<<<dim3((unsigned)(((long)nq * k + 255) / 256)), dim3(256), 0, st>>>((const u64 *)fbkeys.p, (const int *)fbidx.p,
                                                                                                  (int)nq, k, keys_out, nf_dev);
            HAC_HIP(hipGetLastError());
            ++split_searches;
            snprintf(plan_head, sizeof plan_head, "split: scanh_kernel<%d> grid=(%ld,%d) NQ=%d K2=%d chunks=%d lds=%zu seed=%d passes=%d rescore=%s%s", terms, P_last,
                     n_qtiles_last, SH_NQ, K2, n_chunks, lds, seeded, passes_last, rescore_from(), qt_act_last == 8 ? " tiles=half" : qt_act_last == 4 ? " tiles=quarter" : "");
            snprintf(last_plan, sizeof last_plan, "%s fallback=device-side/%lld", plan_head, (long long)nq);
            // (a search that is being captured takes no slot: every replay would write the slot's pinned words again, under whichever
            // live search owns the slot by then)
            const int slot = stream_is_capturing(st) ? -1 : plan_slot();
            if (slot < 0) {
                snprintf(last_plan, sizeof last_plan, "%s fallback=device-side/%lld (status not collected)", plan_head, (long long)nq);
                plan_text_slot = -1;
                return HAC_OK;
            }
            HAC_HIP(hipMemcpyAsync(h_plan + 2 * slot, ws_stat.p, 8, hipMemcpyDeviceToHost, st));
            HAC_HIP(hipEventRecord(ev_plan[slot], st));
            slot_pending[slot] = true;
            slot_nq[slot] = nq;
            plan_text_slot = slot;
            return HAC_OK;
        }
        HAC_HIP(hipMemcpyAsync(h_fb, ws_stat.p, 8, hipMemcpyDeviceToHost, st));
        HAC_HIP(hipStreamSynchronize(st));
        const u32 nfail = h_fb[0];
        float maxratio;
        std::memcpy(&maxratio, &h_fb[1], 4);
        if (level == 0) ++split_searches;
        char plan_here[sizeof last_plan];
        snprintf(plan_here, sizeof plan_here, "split: scanh_kernel<%d> grid=(%ld,%d) NQ=%d K2=%d chunks=%d lds=%zu seed=%d passes=%d rescore=%s%s fallback=%u/%lld err/bound=%.3g",
                 terms, P_last, n_qtiles_last, SH_NQ, K2, n_chunks, lds, seeded, passes_last, rescore_from(), qt_act_last == 8 ? " tiles=half" : qt_act_last == 4 ? " tiles=quarter" : "", nfail, (long long)nq, (double)maxratio);
        std::memcpy(last_plan, plan_here, sizeof last_plan);
        plan_text_slot = -1;
        if (nfail == 0) return HAC_OK;

        // certificate failed for some queries: the next level decides those
        HAC_HIP(hipMemcpyAsync(h_fb + 8, ws_fail.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        HAC_HIP(hipStreamSynchronize(st));
        std::vector<int> idx;
        idx.reserve(nfail);
        for (int64_t i = 0; i < nq; ++i)
            if (h_fb[8 + i]) idx.push_back((int)i);
        const int nf = (int)idx.size();
        std::memcpy(h_fb + 8, idx.data(), (size_t)nf * 4);
        GrowBuf &fbidx = ws_fbidx[level & 1], &fbq = ws_fbq[level & 1], &fbkeys = ws_fbkeys[level & 1];
        HAC_TRY(fbidx.reserve((size_t)nf * 4));
        HAC_TRY(fbq.reserve((size_t)nf * d * 4));
        HAC_TRY(fbkeys.reserve((size_t)nf * k * 8));
        HAC_HIP(hipMemcpyAsync(fbidx.p, h_fb + 8, (size_t)nf * 4, hipMemcpyHostToDevice, st));
        gather_rows_kernel<<<dim3((unsigned)(((long)nf * K4 + 255) / 256)), dim3(256), 0, st>>>(
            reinterpret_cast<const float4 *>(q_dev), (const int *)fbidx.p, nf, K4, (float4 *)fbq.p);
        HAC_HIP(hipGetLastError());
        HAC_HIP(hipStreamSynchronize(st));   // h_fb is reused by the next level
        const bool prof = profiling;
        profiling = false;   // timed kernels of a search: the first level's scans
        int rc;
        if (terms == 1 && nf >= 64) {
            rc = search_keys_split((const float *)fbq.p, nf, k, (u64 *)fbkeys.p, pos_base, st, 1);
            char both[sizeof last_plan];
            snprintf(both, sizeof both, "%.160s ; then %.130s", plan_here, last_plan + 7);
            std::memcpy(plan_here, both, sizeof plan_here);
        } else {
            split_fallback_queries += nf;
            rc = search_keys_exact((const float *)fbq.p, nf, k, (u64 *)fbkeys.p, pos_base, st);
        }
        profiling = prof;
        std::memcpy(last_plan, plan_here, sizeof last_plan);
        HAC_TRY(rc);
        scatter_keys_kernel<<<dim3((unsigned)(((long)nf * k + 255) / 256)), dim3(256), 0, st>>>((const u64 *)fbkeys.p,
                                                                                              (const int *)fbidx.p, nf, k, keys_out);
        HAC_HIP(hipGetLastError());
        HAC_HIP(hipStreamSynchronize(st));
        return HAC_OK;
    }

    // keys_out: device u64 [nq][k], canonical (score desc, row asc) keys of the k best rows per query.
    // Large query sets (the reference searches a whole test set per block: 2.5k - 16k queries) go through
    // in chunks: 512 inside the prefilter (search_keys_split), 1024 for the exact kernels (16 query tiles x 16
    // row streams fill the chip with same-row workgroups sharing an XCD; one launch over 33 query tiles does neither).
    static constexpr int64_t QUERY_CHUNK = 1024;
    // device_entry: the call came through a *_device entry point, which must not synchronize the caller's stream: the
    // prefilter's certificates are then read by the device (option split_decide pins either way for tests)
    int search_keys(const float *q_dev, int64_t nq, int k, u64 *keys_out, u32 pos_base, hipStream_t st, bool device_entry = false) {
        if (nq > 0 && ntotal > 0) {
            if (((uintptr_t)q_dev & 15) != 0) return fail(HAC_ERR_INVALID, "search: query pointer must be 16-byte aligned");
            if ((uint64_t)pos_base + (uint64_t)ntotal > 0xFFFFFFFFull) return fail(HAC_ERR_UNSUPPORTED, "row positions exceed 32 bits");
        }
        if (nq > 0 && ntotal > 0 && decide_prefilter(nq, k, stream_is_capturing(st)) && !(tune.split < 0 && half_image_unavailable)) {
            const int rc_img = ensure_half_image(st);
            if (rc_img == HAC_OK) return search_keys_split(q_dev, nq, k, keys_out, pos_base, st, 0, tune.split_decide < 0 ? device_entry : tune.split_decide == 1);
            if (rc_img != HAC_ERR_OOM || tune.split == 1) return rc_img;
            // auto: no room for the fp16 image -> the exact fp32 kernels answer, with the same bits
            HAC_HIP(hipStreamSynchronize(st));   // (a half_image_kernel of an earlier segment may still run)
            drop_half_images();
            half_image_unavailable = true;
        }
        for (int64_t off = 0; off < nq || off == 0; off += QUERY_CHUNK) {
            const int64_t n = std::min<int64_t>(QUERY_CHUNK, nq - off);
            HAC_TRY(search_keys_exact(q_dev + (size_t)off * d, n, k, keys_out + (size_t)off * k, pos_base, st));
            if (nq == 0) break;
        }
        return HAC_OK;
    }
};

int merge_plan_lds(int k, int &Cm, size_t &lds) {
    Cm = (int)next_pow2((u32)k + MERGE_THREADS);
    lds = (size_t)Cm * 8 + 32;
    return HAC_OK;
}

int check_k(int k) {
    if (k < 1 || k > HAC_MAX_K) return fail(HAC_ERR_INVALID, "k=%d out of range [1, %d]", k, HAC_MAX_K);
    return HAC_OK;
}

}  // namespace
}  // namespace hac

// =============================================================================
// C ABI
// =============================================================================
using namespace hac;

struct hac_index {
    int d = 0;
    std::vector<DeviceIndex *> shards;  // one per device (n_dev == 1 in the one-process-per-GPU deployment)
    int64_t ntotal = 0;
    // multi-device merge workspace lives on shard 0
    GrowBuf ws_lists, ws_out;
    // n_dev > 1: per shard, its share of every add() (see remap_positions_kernel)
    std::vector<std::vector<SpanDesc>> spans;
    std::vector<GrowBuf> ws_spans;
    std::vector<char> spans_dirty;
};

extern "C" {

const char *hac_last_error(void) { return last_error_slot().c_str(); }
const char *hac_version(void) { return "haconvdr-amd 0.6.0 (gfx950)"; }

int hac_index_create(int d, const int *device_ids, int n_dev, hac_index **out) {
    if (!out) return fail(HAC_ERR_INVALID, "hac_index_create: out is null");
    *out = nullptr;
    if (d <= 0 || d % 32 != 0 || d > HAC_MAX_D) return fail(HAC_ERR_INVALID, "d=%d must be a positive multiple of 32, <= %d", d, HAC_MAX_D);
    if (n_dev < 1 || !device_ids) return fail(HAC_ERR_INVALID, "need at least one device id");
    int n_visible = 0;
    if (hipGetDeviceCount(&n_visible) != hipSuccess || n_visible <= 0)
        return fail(HAC_ERR_HIP, "no HIP device visible: the haconvdr search path has no CPU fallback");
    for (int i = 0; i < n_dev; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= n_visible) return fail(HAC_ERR_INVALID, "device id %d out of range (visible: %d)", device_ids[i], n_visible);
    hac_index *idx = new hac_index();
    idx->d = d;
    for (int i = 0; i < n_dev; ++i) {
        DeviceIndex *s = new DeviceIndex();
        int rc = s->init(d, device_ids[i]);
        if (rc != HAC_OK) {
            s->destroy();
            delete s;
            hac_index_destroy(idx);
            return rc;
        }
        idx->shards.push_back(s);
    }
    idx->spans.resize(n_dev);
    idx->ws_spans.resize(n_dev);
    idx->spans_dirty.assign(n_dev, 1);
    *out = idx;
    return HAC_OK;
}

void hac_index_destroy(hac_index *idx) {
    if (!idx) return;
    if (!idx->shards.empty()) {
        DeviceGuard g(idx->shards[0]->device);
        idx->ws_lists.release();
        idx->ws_out.release();
    }
    for (size_t i = 0; i < idx->shards.size(); ++i) {
        if (i < idx->ws_spans.size()) {
            DeviceGuard g(idx->shards[i]->device);
            idx->ws_spans[i].release();
        }
        idx->shards[i]->destroy();
        delete idx->shards[i];
    }
    delete idx;
}

int hac_index_add(hac_index *idx, const float *x, int64_t n) {
    if (!idx) return fail(HAC_ERR_INVALID, "null index");
    if (n < 0 || (n > 0 && !x)) return fail(HAC_ERR_INVALID, "add: bad arguments");
    if (n == 0) return HAC_OK;
    const int S = (int)idx->shards.size();
    if ((uint64_t)idx->ntotal + (uint64_t)n > 0xFFFFFFFFull) return fail(HAC_ERR_UNSUPPORTED, "row positions exceed 32 bits");
    // faiss index_cpu_to_gpu_multiple(shard=True): the rows of one add() are split contiguously across the devices;
    // rows stay numbered by insertion order over the whole index (the per-shard spans translate back)
    int64_t off = 0;
    for (int s = 0; s < S; ++s) {
        const int64_t m = n / S + (s < n % S ? 1 : 0);
        if (m) {
            const int64_t local0 = idx->shards[s]->ntotal;
            HAC_TRY(idx->shards[s]->add_host_rows(x + (size_t)off * idx->d, m));
            if (S > 1) {
                idx->spans[s].push_back(SpanDesc{(u32)local0, (u32)m, (u32)(idx->ntotal + off), 0u});
                idx->spans_dirty[s] = 1;
            }
        }
        off += m;
    }
    idx->ntotal += n;
    return HAC_OK;
}

int hac_index_add_device(hac_index *idx, const float *x_dev, int64_t n, void *hip_stream) {
    if (!idx) return fail(HAC_ERR_INVALID, "null index");
    if (idx->shards.size() != 1) return fail(HAC_ERR_UNSUPPORTED, "add_device needs a single-device index");
    if (n < 0 || (n > 0 && !x_dev)) return fail(HAC_ERR_INVALID, "add_device: bad arguments");
    if (n == 0) return HAC_OK;
    DeviceIndex *s = idx->shards[0];
    DeviceGuard g(s->device);
    HAC_TRY(s->add_device_rows(x_dev, n, (hipStream_t)hip_stream));
    idx->ntotal += n;
    return HAC_OK;
}

int hac_index_reset(hac_index *idx) {
    if (!idx) return fail(HAC_ERR_INVALID, "null index");
    for (auto *s : idx->shards) HAC_TRY(s->reset());
    for (size_t i = 0; i < idx->spans.size(); ++i) {
        idx->spans[i].clear();
        idx->spans_dirty[i] = 1;
    }
    idx->ntotal = 0;
    return HAC_OK;
}

int64_t hac_index_ntotal(const hac_index *idx) { return idx ? idx->ntotal : -1; }

int hac_index_search_keys_device(hac_index *idx, const float *q_dev, int64_t nq, int k, uint64_t *keys_dev,
                                 uint32_t pos_base, void *hip_stream) {
    if (!idx) return fail(HAC_ERR_INVALID, "null index");
    if (idx->shards.size() != 1) return fail(HAC_ERR_UNSUPPORTED, "search_keys_device needs a single-device index");
    HAC_TRY(check_k(k));
    if (nq < 0 || (nq > 0 && (!q_dev || !keys_dev))) return fail(HAC_ERR_INVALID, "search: bad arguments");
    DeviceIndex *s = idx->shards[0];
    DeviceGuard g(s->device);
    return s->search_keys(q_dev, nq, k, (u64 *)keys_dev, pos_base, (hipStream_t)hip_stream, true);
}

int hac_keys_to_results_device(int device, const uint64_t *keys_dev, int64_t n_keys, const int64_t *id_map_dev,
                               float *D_dev, int64_t *I_dev, void *hip_stream) {
    if (n_keys < 0 || (n_keys > 0 && (!keys_dev || !D_dev || !I_dev))) return fail(HAC_ERR_INVALID, "keys_to_results: bad arguments");
    if (n_keys == 0) return HAC_OK;
    DeviceGuard g(device);
    if (!g.ok) return fail(HAC_ERR_HIP, "cannot select HIP device %d", device);
    keys_to_results_kernel<<<dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream>>>(
        (const u64 *)keys_dev, (long)n_keys, (const long long *)id_map_dev, D_dev, (long long *)I_dev);
    HAC_HIP(hipGetLastError());
    return HAC_OK;
}

int hac_merge_keys_device(int device, const uint64_t *lists_dev, int n_lists, int64_t nq, int k, uint64_t *out_dev,
                          void *hip_stream) {
    HAC_TRY(check_k(k));
    if (n_lists < 1 || nq < 0 || (nq > 0 && (!lists_dev || !out_dev))) return fail(HAC_ERR_INVALID, "merge: bad arguments");
    if (nq == 0) return HAC_OK;
    DeviceGuard g(device);
    if (!g.ok) return fail(HAC_ERR_HIP, "cannot select HIP device %d", device);
    int Cm;
    size_t lds;
    merge_plan_lds(k, Cm, lds);
    static bool attr_done[64] = {false};
    if (device >= 0 && device < 64 && !attr_done[device]) {
        HAC_HIP(hipFuncSetAttribute((const void *)merge_keys_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * 1024)));
        attr_done[device] = true;
    }
    merge_keys_kernel<<<dim3((unsigned)nq), dim3(MERGE_THREADS), lds, (hipStream_t)hip_stream>>>(
        (const u64 *)lists_dev, n_lists, (size_t)nq * k, (size_t)k, k, Cm, (u64 *)out_dev, nullptr, nullptr);
    HAC_HIP(hipGetLastError());
    return HAC_OK;
}

int hac_index_search_device(hac_index *idx, const float *q_dev, int64_t nq, int k, float *D_dev, int64_t *I_dev,
                            const int64_t *id_map_dev, void *hip_stream) {
    if (!idx) return fail(HAC_ERR_INVALID, "null index");
    if (idx->shards.size() != 1) return fail(HAC_ERR_UNSUPPORTED, "search_device needs a single-device index");
    HAC_TRY(check_k(k));
    if (nq < 0 || (nq > 0 && (!q_dev || !D_dev || !I_dev))) return fail(HAC_ERR_INVALID, "search: bad arguments");
    if (nq == 0) return HAC_OK;
    DeviceIndex *s = idx->shards[0];
    DeviceGuard g(s->device);
    HAC_TRY(s->ws_keys.reserve((size_t)nq * k * 8));
    HAC_TRY(s->search_keys(q_dev, nq, k, (u64 *)s->ws_keys.p, 0u, (hipStream_t)hip_stream, true));
    return hac_keys_to_results_device(s->device, (const uint64_t *)s->ws_keys.p, nq * k, id_map_dev, D_dev, I_dev, hip_stream);
}

int hac_index_search(hac_index *idx, const float *q, int64_t nq, int k, float *D, int64_t *I) {
    if (!idx) return fail(HAC_ERR_INVALID, "null index");
    HAC_TRY(check_k(k));
    if (nq < 0 || (nq > 0 && (!q || !D || !I))) return fail(HAC_ERR_INVALID, "search: bad arguments");
    if (nq == 0) return HAC_OK;
    const int S = (int)idx->shards.size();
    const size_t qbytes = (size_t)nq * idx->d * 4;
    // each shard searches its rows; positions are global (shard base + local row)
    DeviceIndex *s0 = idx->shards[0];
    {
        DeviceGuard g0(s0->device);
        HAC_TRY(idx->ws_lists.reserve((size_t)S * nq * k * 8));
        HAC_TRY(idx->ws_out.reserve((size_t)nq * k * 8));
    }
    // Every shard searches its own rows on its own stream.  With several devices one host thread per shard drives
    // its device (the prefilter path reads a status word back per search: in sequence the devices would take turns);
    // a thread's failure message is carried back to the caller's error slot.
    auto search_shard = [&](int si) -> int {
        DeviceIndex *s = idx->shards[si];
        DeviceGuard g(s->device);
        if (!g.ok) return fail(HAC_ERR_HIP, "cannot select HIP device %d", s->device);
        HAC_TRY(s->ws_q.reserve(qbytes));
        HAC_TRY(s->ws_keys.reserve((size_t)nq * k * 8));
        HAC_TRY(s->pin_reserve(std::max(qbytes, (size_t)nq * k * 12)));
        HAC_HIP(hipStreamSynchronize(s->stream));
        std::memcpy(s->h_pin, q, qbytes);
        HAC_HIP(hipMemcpyAsync(s->ws_q.p, s->h_pin, qbytes, hipMemcpyHostToDevice, s->stream));
        HAC_TRY(s->search_keys((const float *)s->ws_q.p, nq, k, (u64 *)s->ws_keys.p, 0u, s->stream));
        HAC_HIP(hipMemcpyAsync(s->h_err, s->ws_err.p, 8, hipMemcpyDeviceToHost, s->stream));   // checked after the final wait
        if (S > 1) {   // shard-local rows -> insertion-order positions of the whole index
            const std::vector<SpanDesc> &sp = idx->spans[si];
            if (!sp.empty()) {
                if (idx->spans_dirty[si]) {
                    HAC_TRY(idx->ws_spans[si].reserve(sp.size() * sizeof(SpanDesc)));
                    HAC_HIP(hipStreamSynchronize(s->stream));
                    HAC_HIP(hipMemcpy(idx->ws_spans[si].p, sp.data(), sp.size() * sizeof(SpanDesc), hipMemcpyHostToDevice));
                    idx->spans_dirty[si] = 0;
                }
                const long nk = (long)nq * k;
                remap_positions_kernel<<<dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s->stream>>>(
                    (u64 *)s->ws_keys.p, nk, (const SpanDesc *)idx->ws_spans[si].p, (int)sp.size());
                HAC_HIP(hipGetLastError());
            }
            HAC_HIP(hipStreamSynchronize(s->stream));
        }
        return HAC_OK;
    };
    if (S == 1) {
        HAC_TRY(search_shard(0));
    } else {
        std::vector<int> rcs(S, HAC_OK);
        std::vector<std::string> msgs(S);
        std::vector<std::thread> pool;
        for (int si = 1; si < S; ++si)
            pool.emplace_back([&, si] {
                rcs[si] = search_shard(si);
                if (rcs[si] != HAC_OK) msgs[si] = last_error_slot();
            });
        rcs[0] = search_shard(0);
        if (rcs[0] != HAC_OK) msgs[0] = last_error_slot();
        for (auto &t : pool) t.join();
        for (int si = 0; si < S; ++si)
            if (rcs[si] != HAC_OK) return fail(rcs[si], "shard %d (device %d): %s", si, idx->shards[si]->device, msgs[si].c_str());
    }
    DeviceGuard g0(s0->device);
    const uint64_t *final_keys = (const uint64_t *)s0->ws_keys.p;
    if (S > 1) {
        // gather the per-shard key slabs on shard 0's stream (never the null stream: the index
        // streams are non-blocking, a legacy-stream D2D copy would not be ordered with them)
        for (int si = 0; si < S; ++si) {
            DeviceIndex *s = idx->shards[si];
            HAC_HIP(hipMemcpyAsync((char *)idx->ws_lists.p + (size_t)si * nq * k * 8, s->ws_keys.p, (size_t)nq * k * 8,
                                   hipMemcpyDeviceToDevice, s0->stream));
        }
        HAC_TRY(hac_merge_keys_device(s0->device, (const uint64_t *)idx->ws_lists.p, S, nq, k, (uint64_t *)idx->ws_out.p, s0->stream));
        final_keys = (const uint64_t *)idx->ws_out.p;
    }
    HAC_TRY(s0->ws_D.reserve((size_t)nq * k * 4));
    HAC_TRY(s0->ws_I.reserve((size_t)nq * k * 8));
    HAC_TRY(hac_keys_to_results_device(s0->device, final_keys, nq * k, nullptr, (float *)s0->ws_D.p, (int64_t *)s0->ws_I.p, s0->stream));
    char *hp = (char *)s0->h_pin;
    HAC_HIP(hipMemcpyAsync(hp, s0->ws_I.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s0->stream));
    HAC_HIP(hipMemcpyAsync(hp + (size_t)nq * k * 8, s0->ws_D.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s0->stream));
    HAC_HIP(hipStreamSynchronize(s0->stream));
    std::memcpy(I, hp, (size_t)nq * k * 8);
    std::memcpy(D, hp + (size_t)nq * k * 8, (size_t)nq * k * 4);
    // a scan workgroup that ran into its pass bound: the results are delivered (the affected lists EMPTY), the call says so
    // (every shard's words are read and cleared before returning: a word left set would be blamed on a later search)
    int rc_all = HAC_OK;
    std::string msg_all;
    for (auto *s : idx->shards) {
        DeviceGuard gs(s->device);
        const int rc = s->check_err(s->stream);
        if (rc != HAC_OK && rc_all == HAC_OK) {
            rc_all = rc;
            msg_all = last_error_slot();
        }
    }
    if (rc_all != HAC_OK) return fail(rc_all, "%s", msg_all.c_str());
    return HAC_OK;
}

int hac_index_last_status(hac_index *idx) {
    if (!idx) return fail(HAC_ERR_INVALID, "null index");
    for (auto *s : idx->shards) {
        DeviceGuard g(s->device);
        if (!g.ok) return fail(HAC_ERR_HIP, "cannot select HIP device %d", s->device);
    }
    int rc_all = HAC_OK;
    std::string msg_all;
    for (auto *s : idx->shards) {   // every shard's word is read and cleared, the first failure is the one reported
        DeviceGuard g(s->device);
        const int rc = s->fetch_err(s->stream);
        if (rc != HAC_OK && rc_all == HAC_OK) {
            rc_all = rc;
            msg_all = last_error_slot();
        }
    }
    if (rc_all != HAC_OK) return fail(rc_all, "%s", msg_all.c_str());
    return HAC_OK;
}

int hac_index_set_option(hac_index *idx, const char *name, const char *value) {
    if (!idx || !name) return fail(HAC_ERR_INVALID, "set_option: null argument");
    for (auto *s : idx->shards) HAC_TRY(s->set_option(name, value));
    return HAC_OK;
}

int hac_index_set_profiling(hac_index *idx, int enable) {
    if (!idx) return fail(HAC_ERR_INVALID, "null index");
    for (auto *s : idx->shards) s->profiling = enable != 0;
    return HAC_OK;
}

const char *hac_index_last_plan(const hac_index *idx) {
    return (idx && !idx->shards.empty()) ? idx->shards[0]->plan() : "none";
}

int hac_index_profile_drain(hac_index *idx, float *ms_out, int cap, int *n_out) {
    if (!idx || !n_out || (cap > 0 && !ms_out)) return fail(HAC_ERR_INVALID, "bad arguments");
    DeviceIndex *s = idx->shards[0];
    DeviceGuard g(s->device);
    int n = 0;
    for (size_t i = 0; i < s->ev_used && n < cap; ++i, ++n) {
        HAC_HIP(hipEventSynchronize(s->ev_pool[i].second));
        HAC_HIP(hipEventElapsedTime(&ms_out[n], s->ev_pool[i].first, s->ev_pool[i].second));
    }
    s->ev_used = 0;
    *n_out = n;
    return HAC_OK;
}

}  // extern "C"

#ifdef SH_STAMP
extern "C" int hac_debug_scan_stamps(unsigned long long *out) {   // development builds only (tools/ab_search.py)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hac::g_sh_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
