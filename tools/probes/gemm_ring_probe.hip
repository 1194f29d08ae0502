// Probe (round 5): would several k-tiles in flight per workgroup shorten the small-batch GEMM's k-loop?
// gemm_small_probe.hip measured one exposed load latency per k-tile (0.4 us alone, 0.8 us with 288 tiles pulling) in the
// shipped two-stage 128^2 kernel (stage k+1 is issued, stage k computed, vmcnt(0) + barrier).  This is the same tile, the same
// LDS image, the same MFMA loop with a ring of STAGES buffers and counted waits: STAGES - 1 k-tiles in flight while one is
// computed, one raw s_barrier per k-tile.  Bare epilogue (a few stores per lane, so that nothing is optimized away) -- what is
// compared is the k-loop, on the same number of tiles with one workgroup per CU for every variant.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/gemm_ring_probe.hip -o scratch/p/gemm_ring
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// C[M, N] = A[M, K] . W[N, K]^T, 128 x 128 tiles, 4 waves (2 x 2), 64-deep k-tiles, STAGES ring buffers of 32 KiB.
// One tile per workgroup (grid = tiles): the cross-tile prefetch of the shipped kernel is not what is being measured.
template <int STAGES>
__global__ __launch_bounds__(256, 1) void ring_gemm_kernel(const bf16 *__restrict__ A, const bf16 *__restrict__ W, float *__restrict__ C, int N, int K) {
    constexpr int BM = 128, BN = 128, BK = 64, STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT = K / BK, nx = N / BN;
    const int tile = blockIdx.x, m0 = (tile / nx) * BM, n0 = (tile % nx) * BN;
    const int wm = w >> 1, wn = w & 1, r = lane & 31, hh = lane >> 5;
    typedef const __attribute__((address_space(1))) void *gvp;
    typedef __attribute__((address_space(3))) void *lvp;
    const int srow = lane >> 3;
    const int sch_even = (lane & 7) ^ (srow >> 1), sch_odd = sch_even ^ 4;
    const size_t lane_src_e = (size_t)(w * 32 + srow) * K + sch_even * 8, lane_src_o = (size_t)(w * 32 + srow) * K + sch_odd * 8;
    auto stage = [&](int buf, int kt) {   // 8 DMA instructions per wave
        const bf16 *gA = A + (size_t)m0 * K + kt * BK, *gW = W + (size_t)n0 * K + kt * BK;
        unsigned char *sb = smem + buf * STAGE + w * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t ls = ((i & 1) ? lane_src_o : lane_src_e) + (size_t)i * 8 * K;
            __builtin_amdgcn_global_load_lds((gvp)(gA + ls), (lvp)(sb + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gvp)(gW + ls), (lvp)(sb + BM * 128 + i * 1024), 16, 0, 0);
        }
    };
    int aoff[2], woff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) aoff[t] = (wm * 64 + t * 32 + r) * 128, woff[t] = BM * 128 + (wn * 64 + t * 32 + r) * 128;
    const int sw = (r >> 1) & 7;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    // prologue: STAGES - 1 k-tiles on their way
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < KT) stage(s, s);
    int cur = 0, nxt = STAGES - 1;   // buffer of k-tile kt, buffer that takes k-tile kt + STAGES - 1
    for (int kt = 0; kt < KT; ++kt) {
        // k-tile kt has landed (this wave's part): the DMAs issued after it are those of k-tiles kt+1 .. kt+STAGES-2 that exist
        const int younger = min(STAGES - 2, KT - 1 - kt);
        if (younger >= 3) wait_vm<24>();
        else if (younger == 2) wait_vm<16>();
        else if (younger == 1) wait_vm<8>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();   // every wave's part has landed, and every wave has finished reading buffer nxt (k-tile kt - 1)
        if (kt + STAGES - 1 < KT) stage(nxt, kt + STAGES - 1);
        const unsigned char *sc = smem + cur * STAGE;
        bf16x8 af[2][2], wf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            af[0][t] = *reinterpret_cast<const bf16x8 *>(sc + aoff[t] + ((hh ^ sw) << 4));
            wf[0][t] = *reinterpret_cast<const bf16x8 *>(sc + woff[t] + ((hh ^ sw) << 4));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
                const int c = ((ks + 1) * 2 + hh) ^ sw;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    af[(ks + 1) & 1][t] = *reinterpret_cast<const bf16x8 *>(sc + aoff[t] + (c << 4));
                    wf[(ks + 1) & 1][t] = *reinterpret_cast<const bf16x8 *>(sc + woff[t] + (c << 4));
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][a], wf[ks & 1][b], acc[a][b], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the fragment reads of this k-tile are done before the next barrier lets its buffer go)
        cur = cur == STAGES - 1 ? 0 : cur + 1;
        nxt = nxt == STAGES - 1 ? 0 : nxt + 1;
    }
    // bare epilogue: every accumulator reaches memory (column-strided 4-byte stores: not a product epilogue, the same for every variant)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh, n = n0 + wn * 64 + b * 32 + r;
                C[(size_t)m * N + n] = acc[a][b][e];
            }
}

template <int STAGES> static float run(const bf16 *A, const bf16 *W, float *C, int M, int N, int K, int iters) {
    const size_t lds = (size_t)STAGES * 32768;
    CK(hipFuncSetAttribute((const void *)ring_gemm_kernel<STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles = (M / 128) * (N / 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        for (int i = 0; i < 3; ++i) ring_gemm_kernel<STAGES><<<tiles, 256, lds>>>(A, W, C, N, K);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) ring_gemm_kernel<STAGES><<<tiles, 256, lds>>>(A, W, C, N, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best * 1e3f / iters;
}

int main() {
    const int M = 2048, NMAX = 3072, KMAX = 3072;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    auto mk = [&](size_t n, float sc) { std::vector<bf16> h(n); for (auto &v : h) v = (bf16)(nd(rng) * sc); bf16 *d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d; };
    bf16 *A = mk((size_t)M * KMAX, 1.f), *W = mk((size_t)NMAX * KMAX, 0.02f);
    float *C[4]; for (auto &c : C) CK(hipMalloc(&c, (size_t)M * NMAX * 4));
    // correctness of the ring forms against the two-stage form (same arithmetic, same order: bit-equal)
    {
        const int N = 1920, K = 768;
        run<2>(A, W, C[0], M, N, K, 1); run<3>(A, W, C[1], M, N, K, 1); run<4>(A, W, C[2], M, N, K, 1); run<5>(A, W, C[3], M, N, K, 1);
        std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
        CK(hipMemcpy(h0.data(), C[0], h0.size() * 4, hipMemcpyDeviceToHost));
        for (int v = 1; v < 4; ++v) {
            CK(hipMemcpy(h1.data(), C[v], h1.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
            printf("ring of %d stages against 2: %zu of %zu outputs differ\n", v + 2, bad, h0.size());
        }
        double s = 0; for (float v : h0) s += v; printf("(checksum %.6g)\n", s);
    }
    printf("\nus per launch (chains of 200, best of 5); tiles of 128^2, one workgroup per tile\n");
    struct Sh { const char *name; int N, K; };
    for (Sh sh : {Sh{"N = 1920 (240 tiles), K = 768 ", 1920, 768}, Sh{"N = 768  ( 96 tiles), K = 768 ", 768, 768}, Sh{"N = 768  ( 96 tiles), K = 3072", 768, 3072},
                  Sh{"N = 128  ( 16 tiles), K = 768 ", 128, 768}, Sh{"N = 1920 (240 tiles), K = 3072", 1920, 3072}, Sh{"N = 1920 (240 tiles), K = 64  ", 1920, 64}})
        printf("  %s  2 stages %6.2f   3 stages %6.2f   4 stages %6.2f   5 stages %6.2f\n", sh.name, run<2>(A, W, C[0], M, sh.N, sh.K, 200), run<3>(A, W, C[1], M, sh.N, sh.K, 200),
               run<4>(A, W, C[2], M, sh.N, sh.K, 200), run<5>(A, W, C[3], M, sh.N, sh.K, 200));
    return 0;
}
