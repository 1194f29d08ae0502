// Probe: the packed-fp32 sequence that went wrong in the 128-row GEMM family's residual epilogue (LABNOTES rounds 1-4, 2.4, "A wrong bit the soak
// found"), in isolation.  Every lane loads four residual floats, a (mean, rstd) pair and gamma / beta vectors from large buffers (so
// that the loads miss and return at uneven times), evaluates  (r - mean) * rstd * gamma + beta  as the vector expression hipcc turned
// into v_sub_f32 x 4, v_pk_mul_f32 ... op_sel:[0,1], v_pk_fma_f32, and again component by component with every value pinned to its
// own VGPR; any lane whose two results differ in a bit is counted.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/pk_after_load_probe.hip -o scratch/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
typedef float f4v __attribute__((ext_vector_type(4)));
#ifndef MFMA_WAVES
#define MFMA_WAVES 0   // waves of each 4-wave workgroup that run an MFMA loop instead (0 .. 3)
#endif

__global__ __launch_bounds__(256) void pk_kernel(const float *__restrict__ resid, const float2 *__restrict__ stats, const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, size_t rows, int iters, unsigned *mism, float *sink) {
    const int lane = threadIdx.x & 63;
    if (MFMA_WAVES && (threadIdx.x >> 6) >= 4 - MFMA_WAVES) {   // matrix-pipe traffic from the other waves of the SIMDs (the GEMM's situation)
        typedef float f16v __attribute__((ext_vector_type(16)));
        typedef __bf16 b8v __attribute__((ext_vector_type(8)));
        f16v c0 = {}, c1 = {};
        b8v a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
        for (int it = 0; it < iters * 24; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        }
        if (c0[0] + c1[5] == 123.456f) sink[1] = c0[0];
        return;
    }
    size_t row0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    unsigned bad = 0, bad1 = 0, bad2 = 0, bad4 = 0, bad5 = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it, row0 += (size_t)gridDim.x * 64) {
        const size_t mr = row0 % (rows - 16);
        f4v rs[4], rv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rs[i] = *reinterpret_cast<const f4v *>(resid + (mr + i * 4 + (lane >> 4)) * 768 + (lane & 15) * 4);
        const f4v gam = *reinterpret_cast<const f4v *>(gamma + (lane & 15) * 4);
        const f4v bet = *reinterpret_cast<const f4v *>(beta + (lane & 15) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 st = stats[mr + i * 4 + (lane >> 4)];
            // three expressions, each as the vector form hipcc packs and as a pinned scalar twin:
            //   E1 (r - mean) * rstd          v_sub_f32 x 4 + v_pk_mul_f32 ... op_sel:[0,1] (the scalar operand is the HIGH word of a pair)
            //   E2 r * gamma + beta           v_pk_fma_f32, all operands vectors
            //   E3 the whole expression
            const f4v e1 = (rs[i] - st.x) * st.y;
            const f4v e2 = rs[i] * gam + bet;
            const f4v e4 = rs[i] * st.y;              // E4 v_pk_mul_f32 op_sel alone, operands straight from the loads
            const f4v e5 = (rs[i] - st.x) * gam;      // E5 v_sub_f32 x 4 feeding a v_pk_mul_f32 WITHOUT op_sel
            rv[i] = (rs[i] - st.x) * st.y * gam + bet;
            float c[4] = {rs[i].x, rs[i].y, rs[i].z, rs[i].w}, mean = st.x, rstd = st.y;
            const float gg[4] = {gam.x, gam.y, gam.z, gam.w}, bb[4] = {bet.x, bet.y, bet.z, bet.w};
            asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(mean), "+v"(rstd));
            const float v1[4] = {e1.x, e1.y, e1.z, e1.w}, v2[4] = {e2.x, e2.y, e2.z, e2.w}, v3[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
            const float v4[4] = {e4.x, e4.y, e4.z, e4.w}, v5[4] = {e5.x, e5.y, e5.z, e5.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s1 = (c[k] - mean) * rstd, s2 = fmaf(c[k], gg[k], bb[k]);
                asm volatile("" : "+v"(s1), "+v"(s2));
                float s3 = fmaf(s1, gg[k], bb[k]);
                asm volatile("" : "+v"(s3));
                float s4 = c[k] * rstd, s5 = (c[k] - mean) * gg[k];
                asm volatile("" : "+v"(s4), "+v"(s5));
                bad4 += __float_as_uint(s4) != __float_as_uint(v4[k]);
                bad5 += __float_as_uint(s5) != __float_as_uint(v5[k]);
                bad1 += __float_as_uint(s1) != __float_as_uint(v1[k]);
                bad2 += __float_as_uint(s2) != __float_as_uint(v2[k]);
                bad += __float_as_uint(s3) != __float_as_uint(v3[k]);
            }
            acc += rv[i].x + rv[i].y + rv[i].z + rv[i].w;
        }
    }
    if (bad) atomicAdd(mism, bad);
    if (bad1) atomicAdd(mism + 1, bad1);
    if (bad2) atomicAdd(mism + 2, bad2);
    if (bad4) atomicAdd(mism + 3, bad4);
    if (bad5) atomicAdd(mism + 4, bad5);
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const size_t rows = 1 << 20;   // 3 GB of residual rows: every load misses the caches
    float *resid, *gamma, *beta, *sink; float2 *stats; unsigned *mism;
    CK(hipMalloc(&resid, rows * 768 * 4)); CK(hipMalloc(&stats, rows * 8)); CK(hipMalloc(&gamma, 64 * 4)); CK(hipMalloc(&beta, 64 * 4));
    CK(hipMalloc(&mism, 32)); CK(hipMalloc(&sink, 16));
    {
        std::vector<float> h(1 << 22);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 4096.0f - 8.0f;
        for (size_t off = 0; off < rows * 768; off += h.size()) CK(hipMemcpy(resid + off, h.data(), std::min(h.size(), rows * 768 - off) * 4, hipMemcpyHostToDevice));
        std::vector<float2> hs(rows);
        for (size_t i = 0; i < rows; ++i) hs[i] = make_float2((float)(i % 97) * 0.01f - 0.4f, 0.5f + (float)(i % 31) * 0.05f);
        CK(hipMemcpy(stats, hs.data(), rows * 8, hipMemcpyHostToDevice));
        std::vector<float> g(64), b(64);
        for (int i = 0; i < 64; ++i) { g[i] = 0.9f + 0.01f * i; b[i] = 0.02f * i - 0.3f; }
        CK(hipMemcpy(gamma, g.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, b.data(), 256, hipMemcpyHostToDevice));
    }
    unsigned long long total = 0, t1 = 0, t2 = 0, t4 = 0, t5 = 0, lanes = 0;
    for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemset(mism, 0, 32));
        pk_kernel<<<2048, 256>>>(resid, stats, gamma, beta, rows, 64, mism, sink);
        CK(hipDeviceSynchronize());
        unsigned m[8]; CK(hipMemcpy(m, mism, 32, hipMemcpyDeviceToHost));
        total += m[0]; t1 += m[1]; t2 += m[2]; t4 += m[3]; t5 += m[4]; lanes += 2048ull * 256 * 64 * 16;
    }
    printf("MFMA waves per workgroup %d, %llu elements per expression: differing from the scalar twin: E1 (r - mean) * rstd %llu | E2 r * gamma + beta %llu | E3 whole %llu | E4 r * rstd %llu | E5 (r - mean) * gamma %llu\n", MFMA_WAVES, lanes, t1, t2, total, t4, t5);
    return 0;
}
