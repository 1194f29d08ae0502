// Feasibility probe (round 2): 256x256x64 tiles with FOUR waves -- one per SIMD, 128x128 per wave, the accumulators in the 256
// AGPRs -- which needs a third fewer LDS fragment bytes per flop than gemm8.inc's 128x64 wave tiles.  Bare k-loop, compiler
// scheduled, two-stage LDS-DMA ring, one barrier per k-tile.  Measured: 0.73-0.80 PF with the DMA stream, 1.07-1.24 PF with
// the operands resident (-DG4_NODMA), against 1.02-1.19 PF for gemm8's bare k-loop WITH its stream: with one wave per SIMD a
// DMA instruction's issue stall is the SIMD's stall.  Not pursued.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DG4_NODMA] tools/probes/gemm_4wave_probe.hip -o gemm_4wave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const bf16 *src, unsigned char *dst) {
    typedef const __attribute__((address_space(1))) void *gvp;
    typedef __attribute__((address_space(3))) void *lvp;
    __builtin_amdgcn_global_load_lds((gvp)src, (lvp)dst, 16, 0, 0);
}
// C[M,N] = A[M,K] . W[N,K]^T; one workgroup per (m tile, n tile); K multiple of 64
__global__ __launch_bounds__(256, 1) void g4_kernel(const bf16 *__restrict__ A, const bf16 *__restrict__ W, float *__restrict__ out, int N, int K, int ntiles_n, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 stages x (A 32 KiB | W 32 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int KT = K >> 6;
    const int m16 = lane & 15, kg = lane >> 4;
    const int srow = lane >> 3;
    // DMA lane offset: piece = 8 rows; rows of piece p are 8p + srow; swizzle term ((p&1)*4 + (srow>>1)) & 7
    f32x4 acc[8][8];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int mt = tile / ntiles_n, nt = tile - mt * ntiles_n;
        const bf16 *Ab = A + (size_t)mt * 256 * K, *Wb = W + (size_t)nt * 256 * K;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto stage = [&](int kt, int slot) {   // 64 pieces of 1 KiB (A 32 | W 32), 16 per wave
#ifndef G4_NODMA
            int l = lane;
            asm volatile("" : "+v"(l));
            const unsigned lo0 = (unsigned)((l >> 3) * K + (((l & 7) ^ ((l >> 4) & 7)) * 8));
            const unsigned lo1 = (unsigned)((l >> 3) * K + (((l & 7) ^ ((4 + (l >> 4)) & 7)) * 8));
            const bf16 *base = (w < 2 ? Ab : Wb) + (size_t)((w & 1) * 128) * K + kt * 64;
            unsigned char *dst = smem + slot * 65536 + w * 16384;
#pragma unroll
            for (int j = 0; j < 16; ++j) glds16(base + (size_t)(j * 8) * K + ((j & 1) ? lo1 : lo0), dst + j * 1024);
#endif
        };
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < KT; ++kt) {
            const int slot = kt & 1;
            if (kt + 1 < KT) stage(kt + 1, slot ^ 1);
            const unsigned char *sa = smem + slot * 65536 + (wr * 128) * 128, *sw_ = smem + slot * 65536 + 32768 + (wc * 128) * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[8], wf[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = i * 16 + m16;
                    af[i] = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + (((ks * 4 + kg) ^ ((row >> 1) & 7)) << 4));
                    wf[i] = *reinterpret_cast<const bf16x8 *>(sw_ + row * 128 + (((ks * 4 + kg) ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // keep the result alive: one value per lane
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) t += acc[i][j];
        out[(size_t)tile * 256 + tid] = t.x + t.y + t.z + t.w;
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
    float *out; CK(hipMalloc(&out, (size_t)(M / 256) * 12 * 256 * 4));
    CK(hipFuncSetAttribute((const void *)g4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    struct Cfg { const char *name; int N, K; } cfgs[] = {{"QKV   N=2304 K=768 ", 2304, 768}, {"OUT   N=768  K=768 ", 768, 768}, {"FFN1  N=3072 K=768 ", 3072, 768}, {"FFN2  N=768  K=3072", 768, 3072}};
    for (auto &c : cfgs) {
        const int ntn = c.N / 256, nt = (M / 256) * ntn;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) g4_kernel<<<256, 256, 131072>>>(A, W, out, c.N, c.K, ntn, nt);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) g4_kernel<<<256, 256, 131072>>>(A, W, out, c.N, c.K, ntn, nt);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%s : %.3f ms %.0f TF\n", c.name, ms, 2.0 * M * c.N * c.K / ms / 1e9);
    }
    return 0;
}
