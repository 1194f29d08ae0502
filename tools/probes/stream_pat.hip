// streaming read rate of two per-instruction access patterns over a 3 GB buffer:
//   A: lane l reads 16 B at chunk*1024 + l*16           (T64: contiguous KiB per wave instruction)
//   B: lane l reads 16 B at (chunk>>2)*4096 + l*64 + (chunk&3)*16   (64-B pieces per row)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f4* gp4;
template <int PAT>
__global__ __launch_bounds__(256) void k(const f4* src, long n_groups, float* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f4 acc = {0, 0, 0, 0};
    for (long g = (long)blockIdx.x * 4 + w; g < n_groups; g += (long)gridDim.x * 4) {
        gp4 base = (gp4)(src) + g * 192 * 64;
        f4 ring[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ring[i] = PAT == 0 ? base[i * 64 + lane] : base[(i >> 2) * 256 + lane * 4 + (i & 3)];
        for (int tb = 0; tb < 24; ++tb) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc += ring[i];
                const int t = tb * 8 + i + 8;
                const int tt = t < 192 ? t : 191;
                ring[i] = PAT == 0 ? base[tt * 64 + lane] : base[(tt >> 2) * 256 + lane * 4 + (tt & 3)];
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    const long n_groups = 15625;   // 1M rows x 768 floats
    const size_t bytes = (size_t)n_groups * 192 * 1024;
    f4* src; float* out; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 1024 * 256 * 4));
    CK(hipMemset(src, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) for (int pat = 0; pat < 2; ++pat) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 5; ++it) { if (pat == 0) k<0><<<1024, 256>>>(src, n_groups, out); else k<1><<<1024, 256>>>(src, n_groups, out); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("pattern %c: %.3f ms  %.2f TB/s\n", 'A' + pat, ms, bytes / ms / 1e9);
    }
    return 0;
}
