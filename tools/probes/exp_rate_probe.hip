// Probe for the attention kernel's softmax (LABNOTES rounds 1-4, 2.3 / VERDICT r3 item 5a): what does one 2^x cost on the VALU of a SIMD
//   A  v_exp_f32 (the shipped path: one transcendental instruction per element)
//   B  packed fp16: clamp, round-to-integer by the 1.5 * 2^10 magic add, f = x - n, cubic 2^f, scale 2^n built by a 16-bit shift/add, multiply
//      (two elements per instruction; P is rounded to bf16 behind it anyway)
//   C  as B with a quadratic 2^f (max relative error 1.8e-3, about bf16's own rounding)
//   D / E / F  the plain rates: v_pk_fma_f32, v_pk_fma_f16 (pairs), v_fma_f32 -- is there a cheaper arithmetic for the epilogues?
// Each variant runs 16 independent chains per lane inside a long loop, 4 waves per SIMD, every CU: elements per cycle and SIMD from
// the wall clock and s_memtime-free arithmetic (clock from the A run's known issue rate is not assumed: the table prints ns per element-wave).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/exp_rate_probe.hip -o scratch/exp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <int DEG> __device__ __forceinline__ h2 exp2_pk(h2 x) {
    const h2 lo = {(_Float16)-15.0f, (_Float16)-15.0f}, magic = {(_Float16)1536.0f, (_Float16)1536.0f};
    x = __builtin_elementwise_max(x, lo);
    const h2 t = x + magic;                 // low mantissa bits = round(x) + 512
    const h2 n = t - magic;
    const h2 f = x - n;                     // [-0.5, 0.5]
    h2 p;
    if (DEG == 3) {
        const h2 c3 = {(_Float16)0.0555f, (_Float16)0.0555f}, c2 = {(_Float16)0.2402f, (_Float16)0.2402f}, c1 = {(_Float16)0.6931f, (_Float16)0.6931f}, c0 = {(_Float16)1.0f, (_Float16)1.0f};
        p = (((c3 * f + c2) * f + c1) * f + c0);
    } else {
        const h2 c2 = {(_Float16)0.2436f, (_Float16)0.2436f}, c1 = {(_Float16)0.6951f, (_Float16)0.6951f}, c0 = {(_Float16)0.9998f, (_Float16)0.9998f};
        p = ((c2 * f + c1) * f + c0);
    }
    us2 tb = __builtin_bit_cast(us2, t);
    tb = (us2)(tb << 10) + (us2){(unsigned short)(15u << 10), (unsigned short)(15u << 10)};   // 2^n as fp16 bits; n = -15 -> +0
    return p * __builtin_bit_cast(h2, tb);
}

template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float seed) {
    float acc = 0.f;
    if (MODE == 0) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = seed * (float)(i + 1) - (float)(threadIdx.x & 7);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.0f;   // exp + one full-rate op keeping the chain bounded
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
    } else {
        h2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = h2{(_Float16)(seed * (float)(i + 1)), (_Float16)(-(float)(threadIdx.x & 7))};
        const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = exp2_pk<MODE == 1 ? 3 : 2>(v[i]) - one;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += (float)v[i].x + (float)v[i].y;
    }
    if (acc == 12345.678f) out[0] = acc;   // never true: keeps the chains alive
}

// D / E: what a packed fma costs -- v_pk_fma_f32 (two fp32 per lane and instruction) against v_pk_fma_f16, 8 independent chains of pairs
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void fma_kernel(float *out, int iters, float seed) {
    float acc = 0.f;
    if (MODE == 0) {
        f2 v[8];
        const f2 a = {0.999f, 0.998f}, b = {seed, -seed};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f2{seed * (float)(i + 1), (float)(threadIdx.x & 7)};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], a, b);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i].x + v[i].y;
    } else if (MODE == 1) {
        h2 v[8];
        const h2 a = {(_Float16)0.999f, (_Float16)0.998f}, b = {(_Float16)seed, (_Float16)-seed};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = h2{(_Float16)(seed * (float)(i + 1)), (_Float16)(float)(threadIdx.x & 7)};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], a, b);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += (float)v[i].x + (float)v[i].y;
    } else {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = seed * (float)(i + 1) + (float)(threadIdx.x & 7);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], 0.999f, seed);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int MODE> float run_fma(float *out, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fma_kernel<MODE><<<256 * 4, 256>>>(out, iters, 0.001f); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); fma_kernel<MODE><<<256 * 4, 256>>>(out, iters, 0.001f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    return best;
}

// accuracy of the packed forms against exp2f over [-16, 0]
__global__ void acc_kernel(float *err) {
    float e3 = 0.f, e2 = 0.f, a3 = 0.f, a2 = 0.f;
    for (int i = threadIdx.x; i < 65536; i += blockDim.x) {
        const float x = -16.0f * (float)i / 65536.0f;
        const h2 xx = {(_Float16)x, (_Float16)x};
        const float ref = exp2f((float)xx.x);
        const float r3 = (float)exp2_pk<3>(xx).x, r2 = (float)exp2_pk<2>(xx).x;
        if (ref >= 6.2e-5f) { e3 = fmaxf(e3, fabsf(r3 - ref) / ref); e2 = fmaxf(e2, fabsf(r2 - ref) / ref); }
        a3 = fmaxf(a3, fabsf(r3 - ref)); a2 = fmaxf(a2, fabsf(r2 - ref));
    }
    atomicMax((unsigned *)&err[0], __float_as_uint(e3)); atomicMax((unsigned *)&err[1], __float_as_uint(e2));
    atomicMax((unsigned *)&err[2], __float_as_uint(a3)); atomicMax((unsigned *)&err[3], __float_as_uint(a2));
}

template <int MODE> float run(float *out, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<MODE><<<256 * 4, 256>>>(out, iters, 0.001f); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); rate_kernel<MODE><<<256 * 4, 256>>>(out, iters, 0.001f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    float *out; CK(hipMalloc(&out, 64)); CK(hipMemset(out, 0, 64));
    const int iters = 20000;
    const char *names[3] = {"A v_exp_f32 (+1 full-rate op)", "B packed fp16, cubic (+1 packed op)", "C packed fp16, quadratic (+1 packed op)"};
    float ms[3] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters)};
    // 1024 workgroups of 4 waves on 256 CUs x 4 SIMDs: 4 waves per SIMD; per wave iters * 16 elements per lane
    for (int m = 0; m < 3; ++m) {
        const double wave_elems = (double)iters * 16, ns_per_wave_elem = ms[m] * 1e6 / (wave_elems * 4);   // per SIMD: 4 waves in sequence on one VALU
        printf("%-42s %.3f ms: %.3f ns per wave-wide element on a SIMD (= %.1f cycles at 2.4 GHz)\n", names[m], ms[m], ns_per_wave_elem, ns_per_wave_elem * 2.4);
    }
    {
        const char *fn[3] = {"D v_pk_fma_f32 (8 pairs per lane and iteration)", "E v_pk_fma_f16 (8 pairs per lane and iteration)", "F v_fma_f32 (16 per lane and iteration)"};
        float fm[3] = {run_fma<0>(out, iters), run_fma<1>(out, iters), run_fma<2>(out, iters)};
        const int instr[3] = {8, 8, 16};
        for (int m = 0; m < 3; ++m) {
            const double ns_per_instr = fm[m] * 1e6 / ((double)iters * instr[m] * 4);
            printf("%-50s %.3f ms: %.3f ns per wave instruction on a SIMD (= %.1f cycles at 2.4 GHz), %.1f cycles per element\n", fn[m], fm[m], ns_per_instr,
                   ns_per_instr * 2.4, ns_per_instr * 2.4 * instr[m] / 16);
        }
    }
    acc_kernel<<<1, 256>>>(out + 4); CK(hipDeviceSynchronize());
    float h[4]; CK(hipMemcpy(h, out + 4, 16, hipMemcpyDeviceToHost));
    printf("packed forms vs exp2f on [-16, 0]: max relative error (results >= 2^-14) cubic %.2e quadratic %.2e; max absolute error cubic %.2e quadratic %.2e\n", h[0], h[1], h[2], h[3]);
    return 0;
}
