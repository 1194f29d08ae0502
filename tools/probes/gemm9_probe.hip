// Probe of tools/probes/gemm9.inc (producer / consumer waves, epilogue on the loader waves: built in round 4, NOT shipped -- it
// ties gemm8 on the RESID shapes and loses 13 % on FFN-up, profiles/r04_gemm9_breakdown.txt) against gemm8.inc on the encoder's
// shapes at M = 131328 rows (a multiple of 768): same inputs, outputs compared, interleaved timing in one process.
//   -DG9_DBG_NOEPI: hand-over but no epilogue work on the loaders; -DG9_DBG_NOHANDOVER (with NOEPI): the bare k-loop; G9_ONLY=8|9 runs one kernel
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/gemm9_probe.hip -o gemm9_probe
#include "../../haconvdr_amd/csrc/encoder.hip"
namespace hac { namespace {
#include "gemm9.inc"
} }
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
using namespace hac;
template <typename F> float timeit(F f, int iters) {
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for(int i=0;i<iters;i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); return ms/iters;
}
int main(){
  const int M = getenv("G9_M") ? atoi(getenv("G9_M")) : 131328;   // a multiple of 768
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f,1.f);
  auto mk = [&](size_t n, float sc){ std::vector<float> h(n); for(auto&v:h) v=nd(rng)*sc; float* d; CK(hipMalloc(&d,n*4)); CK(hipMemcpy(d,h.data(),n*4,hipMemcpyHostToDevice)); bf16* b; CK(hipMalloc(&b,n*2)); f32_to_bf16_kernel<<<(n+255)/256,256>>>(d,b,n); CK(hipDeviceSynchronize()); CK(hipFree(d)); return b; };
  auto mkf = [&](size_t n, float sc, float off){ std::vector<float> h(n); for(auto&v:h) v=off+nd(rng)*sc; float* d; CK(hipMalloc(&d,n*4)); CK(hipMemcpy(d,h.data(),n*4,hipMemcpyHostToDevice)); return d; };
  const size_t poolA = (size_t)8208*3072;
  bf16* Apool = mk(poolA, 1.0f);
  bf16* A; CK(hipMalloc(&A,(size_t)M*3072*2));
  for(size_t off=0; off<(size_t)M*3072; off+=poolA) CK(hipMemcpy(A+off, Apool, std::min(poolA,(size_t)M*3072-off)*2, hipMemcpyDeviceToDevice));
  bf16* W = mk((size_t)3072*3072, 0.02f);
  bf16* resid = mk((size_t)M*768, 1.0f);
  float *cvec = mkf(3072, 0.1f, 0.f), *wsum = mkf(3072, 0.5f, 0.f), *rgamma = mkf(3072, 0.1f, 1.f), *rbeta = mkf(3072, 0.05f, 0.f);
  std::vector<float> hs((size_t)(M+64)*2); for (size_t i=0;i<(size_t)M+64;++i){ hs[2*i]=nd(rng)*0.3f; hs[2*i+1]=1.0f+0.2f*std::abs(nd(rng)); }
  float2* stats; CK(hipMalloc(&stats,(size_t)(M+64)*8)); CK(hipMemcpy(stats,hs.data(),(size_t)(M+64)*8,hipMemcpyHostToDevice));   // (64 rows of slack: gemm9's loaders fetch whole pieces)
  int* total; CK(hipMalloc(&total,4)); CK(hipMemcpy(total,&M,4,hipMemcpyHostToDevice));
  bf16 *h8,*h9,*yb8,*yb9; float2 *part8,*part9;
  CK(hipMalloc(&h8,(size_t)M*3072*2)); CK(hipMalloc(&h9,(size_t)M*3072*2)); CK(hipMalloc(&yb8,(size_t)M*768*2)); CK(hipMalloc(&yb9,(size_t)M*768*2));
  CK(hipMalloc(&part8,(size_t)M*12*8)); CK(hipMalloc(&part9,(size_t)M*12*8));
  CK(hipFuncSetAttribute((const void*)gemm8_kernel<EPI8_RESID,true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  CK(hipFuncSetAttribute((const void*)gemm8_kernel<EPI8_GELU,true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  CK(hipFuncSetAttribute((const void*)gemm9_kernel<EPI9_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, G9_LDS));
  CK(hipFuncSetAttribute((const void*)gemm9_kernel<EPI9_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, G9_LDS));
  Gemm8Args g{}; g.n_groups=1; g.A=A; g.W=W; g.total_rows=total; g.astats=stats; g.wsum=wsum; g.cvec=cvec; g.resid=resid; g.rstats=stats; g.rgamma=rgamma; g.rbeta=rbeta;
  struct Cfg{const char* name; int N,K,epi;};
  Cfg cfgs[] = {{"OUT   N=768  K=768 ",768,768,EPI8_RESID},{"FFN1  N=3072 K=768 ",3072,768,EPI8_GELU},{"FFN2  N=768  K=3072",768,3072,EPI8_RESID}};
  for(auto&c: cfgs){
    g.N=c.N; g.K=c.K; g.n_groups = c.epi==EPI8_GELU ? 2 : 1;
    Gemm8Args g8a = g, g9a = g; g8a.yb = yb8; g8a.part = part8; g8a.h = h8; g9a.yb = yb9; g9a.part = part9; g9a.h = h9;
    const size_t nout = (size_t)M*c.N;
    bf16 *o8 = c.epi==EPI8_GELU ? h8 : yb8, *o9 = c.epi==EPI8_GELU ? h9 : yb9;
    CK(hipMemset(o8,0xff,nout*2)); CK(hipMemset(o9,0xff,nout*2)); CK(hipMemset(part8,0xff,(size_t)M*12*8)); CK(hipMemset(part9,0xff,(size_t)M*12*8));
    auto run8 = [&]{ if(c.epi==EPI8_RESID) gemm8_kernel<EPI8_RESID,true><<<256,512,163840>>>(g8a); else gemm8_kernel<EPI8_GELU,true><<<256,512,163840>>>(g8a); };
    auto run9 = [&]{ if(c.epi==EPI8_RESID) gemm9_kernel<EPI9_RESID><<<256,512,G9_LDS>>>(g9a); else gemm9_kernel<EPI9_GELU><<<256,512,G9_LDS>>>(g9a); };
    const char *only = getenv("G9_ONLY");   // "8": gemm8 alone, "9": gemm9 alone
    if (!only || only[0] == '8') { run8(); CK(hipDeviceSynchronize()); printf("%s : gemm8 ran\n", c.name); fflush(stdout); }
    if (!only || only[0] == '9') { run9(); CK(hipDeviceSynchronize()); printf("%s : gemm9 ran\n", c.name); fflush(stdout); }
    if (only) {
      float t=1e9f; for (int r=0;r<3;++r) t=std::min(t, only[0]=='8' ? timeit(run8,5) : timeit(run9,5));
      printf("   gemm%c alone %.3f ms %.0f TF\n", only[0], t, 2.0*M*c.N*c.K/t/1e9); fflush(stdout);
      continue;
    }
    std::vector<unsigned short> a(nout), b(nout);
    CK(hipMemcpy(a.data(),o8,nout*2,hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(),o9,nout*2,hipMemcpyDeviceToHost));
    auto f = [](unsigned short h){ unsigned u = (unsigned)h << 16; float v; memcpy(&v, &u, 4); return v; };
    size_t diff=0, first=(size_t)-1, big=0; double worstd=0;
    for(size_t i=0;i<nout;++i) if(a[i]!=b[i]){ if(first==(size_t)-1) first=i; ++diff; const double fa=f(a[i]), fb=f(b[i]); const double d=std::abs(fa-fb)/(std::abs(fa)+std::abs(fb)+1e-2); if(!(d<=worstd)) worstd = d==d ? std::max(worstd,d) : 1e30; if (d > 0.02) ++big; }
    printf("%s : outputs differ in %zu of %zu elements (gemm9 rounds the tile to bf16 once more), worst relative difference %.3g, %zu above 2 %%", c.name, diff, nout, worstd, big);
    if (big) printf(" (first difference at row %zu col %zu: %04x vs %04x)", first/c.N, first%c.N, a[first], b[first]);
    if (c.epi==EPI8_RESID) {
      std::vector<float> pa((size_t)M*12*2), pb((size_t)M*12*2);
      CK(hipMemcpy(pa.data(),part8,pa.size()*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(pb.data(),part9,pb.size()*4,hipMemcpyDeviceToHost));
      double worst=0; for(size_t i=0;i<pa.size();++i){ double d=std::abs((double)pa[i]-pb[i])/(1.0+std::abs((double)pa[i])); if(!(d<=worst)) worst = d==d ? std::max(worst,d) : 1e30; }
      printf("; partial statistics: worst relative difference %.2e", worst);
    }
    printf("\n");
    float t8=1e9f, t9=1e9f;
    for (int r=0;r<4;++r){ t8=std::min(t8,timeit(run8,5)); t9=std::min(t9,timeit(run9,5)); }
    printf("   gemm8 %.3f ms %.0f TF | gemm9 %.3f ms %.0f TF  (%.1f %%)\n", t8, 2.0*M*c.N*c.K/t8/1e9, t9, 2.0*M*c.N*c.K/t9/1e9, (t8/t9-1)*100);
  }
  return 0;
}
namespace hac { std::string &last_error_slot(){ static std::string s; return s; } int fail(int code, const char *fmt, ...){ (void)fmt; return code; } }
