// bare MFMA + LDS-read loops on random data: 32x32x16 vs 16x16x32 (bf16), same FLOPs and LDS traffic per wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
// wave tile 128x64 per k32: 32x32x16: 2 k16-steps x (4 A + 2 B reads, 8 MFMAs); 16x16x32: 8 A + 4 B reads, 32 MFMAs
template <int SHAPE>
__global__ __launch_bounds__(512) void k(const uint4* src, float* out, int iters) {
    extern __shared__ uint4 lds[];   // 64 KiB of random bf16
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint4* base = lds + w * 256 + lane;
    if (SHAPE == 32) {
        f32x16 acc[4][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 a[4], b[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = base[((it + ks * 4 + i) & 3) * 64 + (i & 1) * 1024];
#pragma unroll
                for (int i = 0; i < 2; ++i) b[i] = base[2048 + ((it + ks + i) & 3) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
            }
        }
        float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x4 acc[8][4] = {};
        for (int it = 0; it < iters; ++it) {
            uint4 a[8], b[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = base[((it + i) & 3) * 64 + (i & 1) * 1024];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = base[2048 + ((it + i) & 3) * 64];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}
int main() {
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<unsigned short> h(4096 * 8);
    for (auto& v : h) { float f = nd(rng); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    uint4* src; float* out; CK(hipMalloc(&src, 65536)); CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) for (int shape : {32, 16}) {
        CK(hipEventRecord(e0));
        if (shape == 32) k<32><<<256, 512, 65536>>>(src, out, iters); else k<16><<<256, 512, 65536>>>(src, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double fl = 256.0 * 8 * iters * 2.0 * 128 * 64 * 32;
        printf("shape %dx%d: %.2f ms  %.0f TF\n", shape, shape, ms, fl / ms / 1e9);
    }
    return 0;
}
