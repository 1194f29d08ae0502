// bare 16x16x32 MFMA + LDS-read loops on random data:
//   A: 8 waves/WG (2 per SIMD), wave tile 128x64  (8 A + 4 B reads, 32 MFMAs per k32)
//   B: 4 waves/WG (1 per SIMD), wave tile 128x128 (8 A + 8 B reads, 64 MFMAs per k32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
template <int NB>
__global__ __launch_bounds__(NB == 4 ? 512 : 256) void k(const uint4* src, float* out, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint4* base = lds + (w & 3) * 256 + lane;
    f32x4 acc[8][NB] = {};
    for (int it = 0; it < iters; ++it) {
        uint4 a[8], b[NB];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = base[((it + i) & 3) * 64 + (i & 1) * 1024];
#pragma unroll
        for (int i = 0; i < NB; ++i) b[i] = base[2048 + ((it + i) & 3) * 64 + (i & 1) * 1024];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < NB; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<unsigned short> h(4096 * 8);
    for (auto& v : h) { float f = nd(rng); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    uint4* src; float* out; CK(hipMalloc(&src, 65536)); CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) for (int nb : {4, 8}) {
        const int iters = nb == 4 ? 20000 : 10000;   // same total flops per CU
        CK(hipEventRecord(e0));
        // LDS sized so that only ONE workgroup fits per CU in both variants
        if (nb == 4) k<4><<<256, 512, 65536 + 32768>>>(src, out, iters); else k<8><<<256, 256, 65536 + 32768>>>(src, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 256.0 * (nb == 4 ? 8 : 4) * iters * 2.0 * 128 * (16.0 * nb) * 32;
        printf("wave tile 128x%d, %d waves/WG: %.2f ms  %.0f TF\n", 16 * nb, nb == 4 ? 8 : 4, ms, fl / ms / 1e9);
    }
    return 0;
}
