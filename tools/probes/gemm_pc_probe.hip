// Probe (round 4): producer / consumer wave specialisation for the encoder's large-batch GEMM.
//
// What the hardware allows.  Every wave of a dispatch gets the SAME register allocation (the kernel descriptor carries one
// granulated VGPR count), so "one 448-register MFMA wave + one 64-register loader wave per SIMD" cannot exist: with two
// waves per SIMD each has at most 256 registers (VGPR + AGPR together).  The form that CAN be built: 8 waves, waves 0..3
// (one per SIMD) only issue MFMAs and their own fragment reads, waves 4..7 (their SIMD partners) issue every LDS-DMA;
// a consumer wave holds a 128 x 96 tile (192 accumulator registers, 64 left for fragments), the workgroup 256 x 192.
//
// Bare k-loop (no epilogue), persistent workgroups, k-steps of 32 through a ring of R = 5 slots (28 KiB each: 16 + 12
// fragment-shaped 1-KiB pieces = one MFMA operand each, lane-linear in LDS, read back conflict-free by ds_read_b128).
// One raw barrier per k-step, placed in the consumer's stream where it no longer needs the current slot: loaders wait
// (counted vmcnt) until step g+1 has landed, barrier, then refill the slot of step g with step g+R.
//   -DPC_MODE=0  both (default)    1  loaders only (consumers just keep the barriers): the CU's L2 -> LDS intake alone
//                2  consumers only (operands resident, no DMA): the MFMA side alone
//   -DPC_FULL_LINES  the loaders fetch 8 rows x 128 B per piece (whole lines, gemm8's source shape) instead of 16 rows x 64 B
//                (timing only: the consumers then multiply the wrong bytes)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DPC_MODE=..] [-DPC_FULL_LINES] tools/probes/gemm_pc_probe.hip -o gemm_pc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
#ifndef PC_MODE
#define PC_MODE 0
#endif
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int R = 5, TM = 256, TN = 192, PA = TM / 16, PW = TN / 16, PIECES = PA + PW, SLOT = PIECES * 1024, PPW = PIECES / 4;   // 7 pieces per loader wave and step
__device__ __forceinline__ void glds16(const bf16 *src, unsigned char *dst) {
    typedef const __attribute__((address_space(1))) void *gvp;
    typedef __attribute__((address_space(3))) void *lvp;
    __builtin_amdgcn_global_load_lds((gvp)src, (lvp)dst, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// C[M,N] = A[M,K] . W[N,K]^T; tiles of 256 x 192; K a multiple of 32
__global__ __launch_bounds__(512, 2) void pc_kernel(const bf16 *__restrict__ A, const bf16 *__restrict__ W, float *__restrict__ out, int K, int ntiles_n, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // R slots x (A 16 pieces | W 12 pieces)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = K >> 5;
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const long G = (long)my_tiles * KS;      // k-steps of this workgroup, over all its tiles
    if (G == 0) return;
#if PC_MODE == 2
    // no DMA: the slots keep what is written here (random signs and mantissas in [1, 2): zeros would flatter the clock)
    for (int i = tid; i < R * SLOT / 4; i += 512) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        reinterpret_cast<unsigned *>(smem)[i] = (h & 0x807F807Fu) | 0x3F803F80u;
    }
    __syncthreads();
#endif
    if (w >= 4) {
        // ------------------------------------------------------------------ loaders: waves 4..7, 7 pieces per step each
        const int lw = w - 4;
        long g_issue = 0;
        int tile = blockIdx.x, ks = 0, slot = 0;
        const bf16 *Ab = A + (size_t)(tile / ntiles_n) * TM * K, *Wb = W + (size_t)(tile % ntiles_n) * TN * K;
#ifdef PC_FULL_LINES
        const unsigned loff = (unsigned)((lane >> 3) * K + (lane & 7) * 8);
#else
        const unsigned loff = (unsigned)((lane & 15) * K + (lane >> 4) * 8);
#endif
        auto issue_step = [&]() {
#if PC_MODE != 2
            unsigned char *dst = smem + slot * SLOT + lw * PPW * 1024;
#pragma unroll
            for (int p = 0; p < PPW; ++p) {
                const int piece = lw * PPW + p;          // 0..15: A row block, 16..27: W row block
#ifdef PC_FULL_LINES
                const int p8 = (ks & 1) * PIECES + piece;   // 8-row block 0..55 of the 64-deep k-tile ks >> 1: A 0..31, W 32..55
                const bf16 *src = (p8 < 2 * PA ? Ab + (size_t)(p8 * 8) * K : Wb + (size_t)((p8 - 2 * PA) * 8) * K) + (ks >> 1) * 64;
#else
                const bf16 *src = (piece < PA ? Ab + (size_t)(piece * 16) * K : Wb + (size_t)((piece - PA) * 16) * K) + ks * 32;
#endif
                glds16(src + loff, dst + p * 1024);
            }
#endif
            ++g_issue;
            slot = slot + 1 == R ? 0 : slot + 1;
            if (++ks == KS) {
                ks = 0;
                tile += gridDim.x;
                Ab = A + (size_t)(tile / ntiles_n) * TM * K;
                Wb = W + (size_t)(tile % ntiles_n) * TN * K;
            }
        };
        for (int i = 0; i < R && g_issue < G; ++i) issue_step();
        if (G >= R) wait_vmcnt<PPW * (R - 1)>(); else wait_vmcnt<0>();     // step 0 has landed
        __builtin_amdgcn_s_barrier();                                        // B_{-1}
        for (long g = 0; g < G; ++g) {
            if (g + R <= G) wait_vmcnt<PPW * (R - 2)>(); else wait_vmcnt<0>();   // step g + 1 has landed
            __builtin_amdgcn_s_barrier();                                    // B_g: the consumers are done with the slot of step g
            if (g_issue < G) issue_step();
        }
        wait_vmcnt<0>();
        return;
    }
    // ---------------------------------------------------------------------- consumers: waves 0..3, 128 x 96 each
    const int cr = w >> 1, cc = w & 1;
    f32x4 acc[8][6];
    bf16x8 af[8], wcur, wnext;
    const unsigned char *fa = smem + (cr * 8) * 1024 + lane * 16, *fw = smem + (PA + cc * 6) * 1024 + lane * 16;
    __builtin_amdgcn_s_barrier();                                            // B_{-1}: step 0 has landed
    int slot = 0;
#if PC_MODE != 1
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(fa + i * 1024);
    wcur = *reinterpret_cast<const bf16x8 *>(fw);
#endif
    long g = 0;
    for (int seq = 0; seq < my_tiles; ++seq) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks, ++g) {
            const int nslot = slot + 1 == R ? 0 : slot + 1;
#if PC_MODE == 1
            __builtin_amdgcn_s_barrier();
#else
            const unsigned char *sw_ = fw + slot * SLOT, *na = fa + nslot * SLOT, *nw = fw + nslot * SLOT;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                wnext = *reinterpret_cast<const bf16x8 *>(sw_ + (j + 1) * 1024);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur, af[i], acc[i][j], 0, 0, 0);
                wcur = wnext;
            }
            // every read of this step's slot is in registers: the loaders may refill it; step g + 1 has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                                    // B_g
            __builtin_amdgcn_sched_barrier(0);
            // (no branch in the matrix stream: behind the last step these reads fetch a stale slot and are never used)
            wnext = *reinterpret_cast<const bf16x8 *>(nw);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i][5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur, af[i], acc[i][5], 0, 0, 0);
                af[i] = *reinterpret_cast<const bf16x8 *>(na + i * 1024);   // the next step's fragment, behind its last use
            }
            wcur = wnext;
#endif
            slot = nslot;
        }
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) t += acc[i][j];
        out[(size_t)(blockIdx.x + seq * gridDim.x) * 256 + tid] = t.x + t.y + t.z + t.w;
    }
}

int main() {
    const int M = 131072;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    auto mk = [&](size_t n, float sc) { std::vector<bf16> h(n); for (auto &v : h) v = (bf16)(nd(rng) * sc); bf16 *d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d; };
    const size_t pool = (size_t)8192 * 3072;
    bf16 *Apool = mk(pool, 1.0f);
    bf16 *A; CK(hipMalloc(&A, (size_t)M * 3072 * 2));
    for (size_t off = 0; off < (size_t)M * 3072; off += pool) CK(hipMemcpy(A + off, Apool, std::min(pool, (size_t)M * 3072 - off) * 2, hipMemcpyDeviceToDevice));
    bf16 *W = mk((size_t)3072 * 3072, 0.02f);
    float *out; CK(hipMalloc(&out, (size_t)(M / 256) * 16 * 256 * 4));
    const int lds = R * SLOT;
    CK(hipFuncSetAttribute((const void *)pc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    struct Cfg { const char *name; int N, K; } cfgs[] = {{"QKV   N=2304 K=768 ", 2304, 768}, {"OUT   N=768  K=768 ", 768, 768}, {"FFN1  N=3072 K=768 ", 3072, 768}, {"FFN2  N=768  K=3072", 768, 3072}};
    printf("producer/consumer probe, mode %d%s, tile 256x192, ring %d x %d KiB\n", PC_MODE,
#ifdef PC_FULL_LINES
           " full-line pieces",
#else
           " fragment-shaped pieces",
#endif
           R, SLOT / 1024);
    for (auto &c : cfgs) {
        const int ntn = c.N / TN, nt = (M / TM) * ntn;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            pc_kernel<<<256, 512, lds>>>(A, W, out, c.K, ntn, nt);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) pc_kernel<<<256, 512, lds>>>(A, W, out, c.K, ntn, nt);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 5);
        }
        const double steps_per_cu = (double)nt * (c.K / 32) / 256.0;
        printf("%s : %.3f ms %.0f TF-equivalent | %.0f ns per 256x192x32 k-step per CU | %.1f GB/s of DMA per CU\n", c.name, best, 2.0 * M * c.N * c.K / best / 1e9,
               best * 1e6 / steps_per_cu, SLOT * steps_per_cu / (best * 1e-3) / 1e9);
    }
#if PC_MODE == 0 && !defined(PC_FULL_LINES)
    {   // correctness of the structure: one tile against a host dot product of a few elements' SUM is not practical here; the
        // per-lane checksum of tile 0 is compared between this build and the consumers-only arithmetic on the host
        const int K = 768, ntn = 768 / TN;
        pc_kernel<<<256, 512, lds>>>(A, W, out, K, ntn, (M / TM) * ntn);
        CK(hipDeviceSynchronize());
        std::vector<float> h(256); CK(hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost));
        std::vector<bf16> ha((size_t)256 * K), hw((size_t)192 * K);
        CK(hipMemcpy(ha.data(), A, ha.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hw.data(), W, hw.size() * 2, hipMemcpyDeviceToHost));
        // consumer wave 0 (tid 0..63): rows 0..127, cols 0..95; lane (m16 = tid & 15, kg = tid >> 4) sums C[i*16 + m16][j*16 + 4 kg + e] (transposed accumulators)
        double worst = 0;
        for (int tid = 0; tid < 64; tid += 7) {
            double ref = 0;
            for (int i = 0; i < 8; ++i) for (int j = 0; j < 6; ++j) for (int e = 0; e < 4; ++e) {
                const int row = i * 16 + (tid & 15), col = j * 16 + 4 * (tid >> 4) + e;
                double s = 0; for (int k = 0; k < K; ++k) s += (double)(float)ha[(size_t)row * K + k] * (double)(float)hw[(size_t)col * K + k];
                ref += s;
            }
            worst = std::max(worst, std::abs(ref - h[tid]) / (std::abs(ref) + 1.0));
        }
        printf("tile 0 checksum vs host: worst relative difference %.2e %s\n", worst, worst < 2e-2 ? "(ok)" : "(MISMATCH)");
    }
#endif
    return 0;
}
