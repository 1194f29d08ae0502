// Probe (round 6): does the softmax's vector work hide in the MFMA gaps of the SAME wave's stream?
// (round 5's coexec_probe.hip asked the question for DIFFERENT waves of a SIMD: there the times add.)
// One workgroup per CU, WPS waves per SIMD (1: __launch_bounds__(256), 512 VGPRs; 2: 512 threads, 256 VGPRs; 4: 1024, 128).
// Every wave runs the attention step's arithmetic on registers only (no LDS, no memory):
//   S  = 4 x v_mfma_f32_32x32x16_bf16 (one dependent chain, C operand of the first = the reference tuple)
//   softmax mix on the 16 scores of a lane: 8 v_max3, 16 v_exp, 16 v_add, 8 v_cvt_pk   (48 vector instructions)
//   O += 4 x v_mfma_f32_32x32x16_bf16 (two chains of two)
// Forms:  0 = MFMAs only, 1 = softmax only, 2 = serial (S, softmax, PV inside one iteration, today's step),
//         3 = software pipeline: S of block i+1 and PV of block i-1 are independent of the softmax of block i, the three are
//             woven by sched_group_barrier: one MFMA, then FILL vector instructions, eight times,
//         4 = the same work left to hipcc's own scheduler (no group barriers).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/inwave_probe.hip -o scratch/p/inwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void softmax_mix(const f32x16 &s, float &lsum, float &mx, bf16x8 (&pf)[2]) {
    float m = mx;
#pragma unroll
    for (int e = 0; e < 16; e += 2) m = __builtin_fmaxf(__builtin_fmaxf(s[e], s[e + 1]), m);   // v_max3_f32
    mx = m;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(s[e]);
        lsum += p;
        pf[e >> 3][e & 7] = (__bf16)p;
    }
}

template <int FORM, int FILL, int THREADS>
__global__ __launch_bounds__(THREADS) void inwave_kernel(float *out, unsigned long long *cyc, int iters) {
    bf16x8 kf[4], qf[4], vf[2][2];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            kf[j][i] = (__bf16)(0.01f * ((threadIdx.x + i + j) & 15) - 0.07f);
            qf[j][i] = (__bf16)(0.02f * ((threadIdx.x - i + 3 * j) & 7) - 0.06f);
            vf[j >> 1][j & 1][i] = (__bf16)(0.03f * ((threadIdx.x + 5 * i + j) & 3));
        }
    f32x16 negm, o[2], s_cur, s_nxt;
    for (int e = 0; e < 16; ++e) { negm[e] = -0.5f - 1e-3f * (threadIdx.x & 3); o[0][e] = 0.f; o[1][e] = 0.f; s_cur[e] = 0.01f * e; s_nxt[e] = 0.02f * e; }
    bf16x8 pf[2], pf_prev[2];
    for (int i = 0; i < 8; ++i) { pf[0][i] = pf[1][i] = pf_prev[0][i] = pf_prev[1][i] = (__bf16)0.5f; }
    float lsum = 0.f, mx = -1e30f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (FORM == 0) {
            f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], negm, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][tt], pf[s2], o[tt], 0, 0, 0);
            asm volatile("" ::"v"(s));    // keep S alive
        } else if (FORM == 1) {
            asm volatile("" : "+v"(s_cur));   // opaque: nothing of the mix is loop-invariant
            softmax_mix(s_cur, lsum, mx, pf);
            asm volatile("" ::"v"(pf[0]), "v"(pf[1]));
        } else if (FORM == 2) {
            f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], negm, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            softmax_mix(s, lsum, mx, pf);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][tt], pf[s2], o[tt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // block i: softmax of s_a -> pf_a; block i+1: S MFMAs -> s_b; block i-1: PV MFMAs with pf_b.  Two blocks per iteration with the
            // roles of the register sets swapped, so no copies are needed.
            auto body = [&](const f32x16 &s_in, f32x16 &s_out, const bf16x8 (&p_in)[2], bf16x8 (&p_out)[2]) {
                s_out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], negm, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) s_out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s_out, 0, 0, 0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][tt], p_in[s2], o[tt], 0, 0, 0);
                softmax_mix(s_in, lsum, mx, p_out);
                if (FORM == 3) {
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x402, FILL, 0);    // VALU | TRANS
                    }
                }
            };
            body(s_cur, s_nxt, pf_prev, pf);
            body(s_nxt, s_cur, pf, pf_prev);
            ++it;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = lsum + mx + o[0][0] + o[1][5] + negm[0] + s_cur[2] + s_nxt[3] + (float)pf_prev[1][1] + (float)pf[0][1];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) { cyc[threadIdx.x >> 6] = t1 - t0; cyc[16 + (threadIdx.x >> 6)] = r1 - r0; }
}

template <int FORM, int FILL, int THREADS> void run(const char *name, float *out, unsigned long long *cyc, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    inwave_kernel<FORM, FILL, THREADS><<<256, THREADS>>>(out, cyc, iters);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        inwave_kernel<FORM, FILL, THREADS><<<256, THREADS>>>(out, cyc, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    unsigned long long h[32]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    // s_memtime counts shader cycles, s_memrealtime 100 MHz: wave 0 (the oldest of its SIMD) and the workgroup's last wave
    const int nw = THREADS / 64, wps = THREADS / 256;
    const double mhz = (double)h[0] / (double)h[16] * 100.0;
    printf("%-40s waves/SIMD %d: %7.3f ms | per SIMD %6.1f ns per block | wave 0: %6.1f cycles per block at %4.0f MHz, last wave %6.1f | SIMD cycles per block (wall x clock / blocks) %6.1f\n", name, wps, best,
           best * 1e6 / iters / wps, (double)h[0] / iters, mhz, (double)h[nw - 1] / iters, best * 1e-3 * mhz * 1e6 / iters / wps);
}

int main() {
    float *out; CK(hipMalloc(&out, 4096));
    unsigned long long *cyc; CK(hipMalloc(&cyc, 32 * 8));
    const int iters = 40000;
    run<0, 0, 256>("MFMA only (8 per block)", out, cyc, iters);
    run<1, 0, 256>("softmax mix only (48 vector instr.)", out, cyc, iters);
    run<2, 0, 256>("serial S | softmax | PV (today's step)", out, cyc, iters);
    run<3, 4, 256>("pipelined, 1 MFMA : 4 vector", out, cyc, iters);
    run<3, 5, 256>("pipelined, 1 MFMA : 5 vector", out, cyc, iters);
    run<3, 6, 256>("pipelined, 1 MFMA : 6 vector", out, cyc, iters);
    run<3, 7, 256>("pipelined, 1 MFMA : 7 vector", out, cyc, iters);
    run<4, 0, 256>("pipelined, hipcc's own schedule", out, cyc, iters);
    run<0, 0, 512>("MFMA only", out, cyc, iters);
    run<1, 0, 512>("softmax mix only", out, cyc, iters);
    run<2, 0, 512>("serial", out, cyc, iters);
    run<3, 6, 512>("pipelined, 1 MFMA : 6 vector", out, cyc, iters);
    run<4, 0, 512>("pipelined, hipcc's own schedule", out, cyc, iters);
    run<0, 0, 1024>("MFMA only", out, cyc, iters);
    run<1, 0, 1024>("softmax mix only", out, cyc, iters);
    run<2, 0, 1024>("serial", out, cyc, iters);
    run<3, 6, 1024>("pipelined, 1 MFMA : 6 vector", out, cyc, iters);
    return 0;
}
