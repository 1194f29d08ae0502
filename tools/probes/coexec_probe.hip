// Probe (round 5): do VALU work and MFMA work of DIFFERENT waves on one SIMD overlap on this part?
// One workgroup per CU, 16 waves (4 per SIMD).  Mode bits: 1 = waves with ((w >> 2) & 1) == 0 run an MFMA loop (independent
// chains), 2 = the other waves run a VALU loop (v_exp_f32 + v_fma_f32, the attention softmax's mix); both = 3.  If the two kinds
// of work co-execute, time(3) ~ max(time(1), time(2)); if the SIMD serialises them, time(3) ~ time(1) + time(2).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/coexec_probe.hip -o scratch/p/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(1024) void coexec_kernel(float *out, int iters, int mode, int mfma_waves_mask) {
    const int w = threadIdx.x >> 6;
    const bool is_mfma = ((mfma_waves_mask >> (w & 15)) & 1) != 0;
    float acc_out = 0.f;
    if (is_mfma) {
        if ((mode & 1) && (mode & 8)) {     // v_mfma_f32_16x16x32_bf16 (4 passes), the same flops per iteration: 16 of them, 8 independent chains
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
            f32x4 c[8];
            for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) c[j][e] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
            }
            for (int j = 0; j < 8; ++j) acc_out += c[j][0] + c[j][3];
        } else if (mode & 1) {
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
            f32x16 c[CHAINS];
            for (int j = 0; j < CHAINS; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8 / CHAINS; ++r)
#pragma unroll
                    for (int j = 0; j < CHAINS; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
            }
            for (int j = 0; j < CHAINS; ++j) acc_out += c[j][0] + c[j][7];
        }
    } else {
        if (mode & 2) {
            float v[16];
            for (int e = 0; e < 16; ++e) v[e] = 0.01f * (threadIdx.x + e);
            float l = 0.f;
            for (int it = 0; it < iters; ++it) {
                if (mode & 4) {                     // plain full-rate VALU only: three fma per score
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float p = fmaf(v[e], 0.999f, 0.001f);
                        l = fmaf(p, 0.5f, l);
                        v[e] = fmaf(p, 0.5f, -0.25f);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {      // the softmax step's mix per 16 scores: exp2, add, fma
                        const float p = __builtin_amdgcn_exp2f(v[e]);
                        l += p;
                        v[e] = fmaf(p, 0.5f, -0.25f);
                    }
                }
            }
            acc_out = l + v[3];
        }
    }
    if (acc_out == 12345.678f) out[threadIdx.x] = acc_out;
}

template <int CHAINS> float run(float *out, int iters, int mode, int mask) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    coexec_kernel<CHAINS><<<256, 1024>>>(out, iters, mode, mask);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        coexec_kernel<CHAINS><<<256, 1024>>>(out, iters, mode, mask);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    float *out; CK(hipMalloc(&out, 4096));
    const int iters = 20000;
    // waves w, w+4, w+8, w+12 share a SIMD: mask 0x0F0F = two MFMA waves + two VALU waves per SIMD; 0x00FF idem (other pairing);
    // 0x5555: if consecutive waves alternate SIMD halves this puts the kinds on different SIMDs (control)
    for (int mask : {0x0F0F, 0x00FF, 0x5555}) {
        const float m1 = run<1>(out, iters, 1, mask), v = run<1>(out, iters, 2, mask), b1 = run<1>(out, iters, 3, mask);
        const float m4 = run<4>(out, iters, 1, mask), b4 = run<4>(out, iters, 3, mask);
        printf("mask %04x: MFMA-only (1 dependent chain) %.3f ms, (4 chains) %.3f ms | VALU-only %.3f ms | both: %.3f ms (1 chain), %.3f ms (4 chains)\n", mask, m1, m4, v, b1, b4);
        const float vf = run<4>(out, iters, 2 | 4, mask), bf = run<4>(out, iters, 3 | 4, mask);
        printf("           plain fma instead of exp2 + add + fma: VALU-only %.3f ms | both %.3f ms (4 chains)\n", vf, bf);
        const float m16 = run<4>(out, iters, 1 | 8, mask), b16 = run<4>(out, iters, 3 | 8, mask), b16f = run<4>(out, iters, 3 | 4 | 8, mask);
        printf("           v_mfma_f32_16x16x32_bf16 (same flops): MFMA-only %.3f ms | both %.3f ms (exp2 mix), %.3f ms (plain fma)\n", m16, b16, b16f);
    }
    return 0;
}
