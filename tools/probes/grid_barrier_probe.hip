// Probe (round 5): what would a device-side stage barrier cost against a kernel boundary?
// The small-batch forward (4 x 512 tokens) is ~100 dependent kernels of 5-20 us; an EMPTY dependent launch measures ~4.5 us.
// A persistent kernel would replace each boundary by a grid-wide barrier: every workgroup makes its stores visible to the other
// XCDs (release: L2 write-back), arrives on a device-scope counter, spins until all have arrived, and invalidates its caches
// (acquire) before reading what the others wrote.  This measures exactly that sequence, N times in one launch, on 256 workgroups
// (one per CU, all co-resident), with a little real traffic per stage (each workgroup writes 4 KiB and reads 4 KiB another
// workgroup wrote in the stage before -- checked, so the barrier is a working one).
// Every spin is bounded: a workgroup that waits longer than ~50 ms gives up and flags it (no hang, whatever happens).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/grid_barrier_probe.hip -o scratch/p/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)

__global__ __launch_bounds__(256) void stages_kernel(unsigned *counter, unsigned *flag, unsigned *buf, int n_stage, int with_fences) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    unsigned bad = 0;
    __shared__ unsigned give_up;
    for (int s = 0; s < n_stage; ++s) {
        if (tid == 0) give_up = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u;
        __syncthreads();
        if (give_up) break;      // (workgroup-uniform) somebody hit the spin limit: everybody leaves
        // the stage's "work": 4 KiB written per workgroup ...
        unsigned *mine = buf + ((size_t)(s & 1) * nwg + wg) * 1024;
        for (int i = tid; i < 1024; i += 256) mine[i] = (unsigned)s * 7919u + wg * 1024u + i;
        // ---- the barrier
        if (with_fences) __threadfence();                       // release: this workgroup's stores reach memory the other XCDs see
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(s + 1) * nwg;
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 400000) { atomicOr(flag, 1u); break; }   // bounded: ~25 ms
            }
        }
        __syncthreads();
        if (with_fences) __threadfence();                       // acquire side for the whole workgroup
        // ... and 4 KiB read that ANOTHER workgroup (on another XCD: wg + 1) wrote in this stage
        const unsigned other = (wg + 1) % nwg;
        const unsigned *theirs = buf + ((size_t)(s & 1) * nwg + other) * 1024;
        for (int i = tid; i < 1024; i += 256)
            bad += __builtin_nontemporal_load(theirs + i) != (unsigned)s * 7919u + other * 1024u + i;
    }
    if (bad) atomicOr(flag, 2u);
}
__global__ void empty_kernel(unsigned *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ __launch_bounds__(256) void stage_as_kernel(unsigned *flag, unsigned *buf, int s) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    // the same stage as a kernel of its own: reads what the previous launch wrote, writes for the next
    unsigned bad = 0;
    if (s > 0) {
        const unsigned other = (wg + 1) % nwg;
        const unsigned *theirs = buf + ((size_t)((s - 1) & 1) * nwg + other) * 1024;
        for (int i = tid; i < 1024; i += 256) bad += theirs[i] != (unsigned)(s - 1) * 7919u + other * 1024u + i;
    }
    unsigned *mine = buf + ((size_t)(s & 1) * nwg + wg) * 1024;
    for (int i = tid; i < 1024; i += 256) mine[i] = (unsigned)s * 7919u + wg * 1024u + i;
    if (bad) atomicOr(flag, 2u);
}

int main() {
    const int NWG = 256, NS = 200;
    unsigned *counter, *flag, *buf;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&buf, (size_t)2 * NWG * 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int fences = 1; fences >= 0; --fences) {
        float best = 1e9f; unsigned hflag = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(counter, 0, 4)); CK(hipMemset(flag, 0, 4)); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            stages_kernel<<<NWG, 256>>>(counter, flag, buf, NS, fences);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            unsigned f; CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost)); hflag |= f;
        }
        printf("persistent kernel, %d stages on %d workgroups, %s: %.2f us per stage (flags: %u%s%s)\n", NS, NWG,
               fences ? "release / acquire fences around the barrier" : "NO fences (timing only: reads may be stale)", best * 1e3f / NS, hflag,
               (hflag & 1) ? " SPIN LIMIT HIT" : "", (hflag & 2) ? (fences ? " STALE DATA READ" : " stale data read, as expected without fences") : "");
    }
    {
        float best = 1e9f; unsigned hflag = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(flag, 0, 4)); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int s = 0; s < NS; ++s) stage_as_kernel<<<NWG, 256>>>(flag, buf, s);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            unsigned f; CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost)); hflag |= f;
        }
        printf("the same %d stages as %d dependent kernel launches: %.2f us per stage (flags: %u)\n", NS, NS, best * 1e3f / NS, hflag);
        best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int s = 0; s < NS; ++s) empty_kernel<<<NWG, 256>>>(nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("%d empty dependent launches: %.2f us each\n", NS, best * 1e3f / NS);
    }
    return 0;
}
