// Probe: do the workgroups of gemm8_kernel, which start together and take the same time per tile, lose time because their
// epilogues (the only phase that writes, and for RESID reads, the residual stream) hit HBM all at once?  Starts the
// workgroups in phases (Gemm8Args::stagger / stagger_mode) on the encoder's four shapes at the bench's M = 512000.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/gemm8_stagger_probe.hip -o scratch/g8_stagger
#include "../../haconvdr_amd/csrc/encoder.hip"
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
using namespace hac;
template <int EPI> float run1(Gemm8Args g, int iters){
  const size_t lds = 163840;
  CK(hipFuncSetAttribute((const void*)gemm8_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gemm8_kernel<EPI, true><<<256,512,lds>>>(g);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for(int i=0;i<iters;i++) gemm8_kernel<EPI, true><<<256,512,lds>>>(g);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); return ms/iters;
}
static float run(Gemm8Args g, int epi, int iters){
  if(epi==EPI8_QKV) return run1<EPI8_QKV>(g,iters); if(epi==EPI8_RESID) return run1<EPI8_RESID>(g,iters); return run1<EPI8_GELU>(g,iters);
}
int main(){
  const int M = getenv("G8_M") ? atoi(getenv("G8_M")) : 512000;
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f,1.f);
  auto mk = [&](size_t n, float sc){ std::vector<float> h(n); for(auto&v:h) v=nd(rng)*sc; float* d; CK(hipMalloc(&d,n*4)); CK(hipMemcpy(d,h.data(),n*4,hipMemcpyHostToDevice)); bf16* b; CK(hipMalloc(&b,n*2)); f32_to_bf16_kernel<<<(n+255)/256,256>>>(d,b,n); CK(hipDeviceSynchronize()); CK(hipFree(d)); return b; };
  const size_t poolA = (size_t)8192*3072;
  bf16* Apool = mk(poolA, 1.0f);
  bf16* A; CK(hipMalloc(&A,(size_t)M*3072*2));
  for(size_t off=0; off<(size_t)M*3072; off+=poolA) CK(hipMemcpy(A+off, Apool, std::min(poolA,(size_t)M*3072-off)*2, hipMemcpyDeviceToDevice));
  bf16* W = mk((size_t)3072*3072, 0.02f);
  float* vec; CK(hipMalloc(&vec, 3072*4*4)); CK(hipMemset(vec,0,3072*16));
  int* total; CK(hipMalloc(&total,4)); CK(hipMemcpy(total,&M,4,hipMemcpyHostToDevice));
  bf16 *q,*k,*vt,*h,*yb,*yb2; float2 *stats,*part;
  CK(hipMalloc(&q,(size_t)M*768*2)); CK(hipMalloc(&k,(size_t)M*768*2)); CK(hipMalloc(&vt,(size_t)768*(M+64)*2)); CK(hipMalloc(&h,(size_t)M*3072*2)); CK(hipMalloc(&yb,(size_t)M*768*2)); CK(hipMalloc(&yb2,(size_t)M*768*2)); CK(hipMemset(yb2,0,(size_t)M*768*2));
  CK(hipMalloc(&stats,(size_t)M*8)); CK(hipMalloc(&part,(size_t)M*12*8));
  fill_identity_stats_kernel<<<(M+255)/256,256>>>(stats,(size_t)M); CK(hipDeviceSynchronize());
  Gemm8Args g{}; g.n_groups=1; g.A=A; g.W=W; g.total_rows=total; g.astats=stats; g.wsum=vec; g.cvec=vec+3072; g.q=q; g.k=k; g.v16=vt; g.resid=yb2; g.rstats=stats; g.rgamma=vec+6144; g.rbeta=vec+9216; g.yb=yb; g.part=part; g.h=h;
  struct Cfg{const char* name; int N,K,epi;};
  Cfg cfgs[] = {{"QKV   N=2304 K=768 ",2304,768,EPI8_QKV},{"OUT   N=768  K=768 ",768,768,EPI8_RESID},{"FFN1  N=3072 K=768 ",3072,768,EPI8_GELU},{"FFN2  N=768  K=3072",768,3072,EPI8_RESID}};
  const int staggers[] = {0, 2, 4, 6, 8, 12, 16, 24};
  for(auto&c: cfgs){
    g.N=c.N; g.K=c.K; g.n_groups = c.epi==EPI8_GELU ? 2 : 1;
    if (getenv("G8_BASE_ONLY")) {   // one figure per class and column-group count (A/B of build flags, e.g. -DG8_NT_EPIS=0)
      for (int ng : {1, 2, 4}) {
        if ((c.N / 256) % ng) continue;
        g.n_groups = ng; g.stagger = 0; float best = 1e9f; for (int r = 0; r < 5; ++r) best = std::min(best, run(g, c.epi, 4));
        printf("%s %s n_groups %d: %.3f ms %4.0f TF\n", getenv("G8_BASE_ONLY"), c.name, ng, best, 2.0*M*c.N*c.K/best/1e9); fflush(stdout);
      }
      continue;
    }
    for (int mode = 0; mode < 4; ++mode) {
      printf("%s mode %d:", c.name, mode);
      for (int st : staggers) {
        if (mode && !st) continue;
        g.stagger = st; g.stagger_mode = mode;
        float best = 1e9f; for (int r = 0; r < 3; ++r) best = std::min(best, run(g, c.epi, 4));
        printf("  s%-2d %.3f ms %4.0f TF", st, best, 2.0*M*c.N*c.K/best/1e9);
      }
      printf("\n"); fflush(stdout);
    }
  }
  return 0;
}
namespace hac { std::string &last_error_slot(){ static std::string s; return s; } int fail(int code, const char *fmt, ...){ (void)fmt; return code; } }
