// Probe of the large-batch GEMM (haconvdr_amd/csrc/gemm8.inc) on the encoder's five shapes at M = 131072: timing, and with
// -DG8_STAMP in-kernel s_memtime stamps around the tile boundary (epilogue, first k-tiles); -DG8_NO_EPI times the k-loop alone.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DG8_STAMP] [-DG8_NO_EPI] tools/probes/gemm8_probe.hip -o gemm8_probe
#include "../../haconvdr_amd/csrc/encoder.hip"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
using namespace hac;
static int GRID = getenv("G8_GRID") ? atoi(getenv("G8_GRID")) : 256;   // workgroups (G8_GRID=16: two per XCD -- is an epilogue slow by itself or because every CU runs one?)
// SPLIT = true: the shipped form (operand-split DMA roles, 160 KiB); false: round 2's form (128 KiB).  Both in one process,
// interleaved rounds (cdna_hip_programming.md 5.4 rule 24).
template <int EPI, bool SPLIT> float run1(Gemm8Args g, int iters){
  const size_t lds = SPLIT ? 163840 : 131072;
  CK(hipFuncSetAttribute((const void*)gemm8_kernel<EPI, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gemm8_kernel<EPI, SPLIT><<<GRID,512,lds>>>(g);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for(int i=0;i<iters;i++) gemm8_kernel<EPI, SPLIT><<<GRID,512,lds>>>(g);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); return ms/iters;
}
template <int EPI> float run(Gemm8Args g, int iters, float* t_old){
  float best_new = 1e9f, best_old = 1e9f;
  for (int r = 0; r < 4; ++r) { best_old = std::min(best_old, run1<EPI,false>(g, iters)); best_new = std::min(best_new, run1<EPI,true>(g, iters)); }
  *t_old = best_old; return best_new;
}
int main(){
  const int M = 131072;
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f,1.f);
  auto mk = [&](size_t n, float sc){ std::vector<float> h(n); for(auto&v:h) v=nd(rng)*sc; float* d; CK(hipMalloc(&d,n*4)); CK(hipMemcpy(d,h.data(),n*4,hipMemcpyHostToDevice)); bf16* b; CK(hipMalloc(&b,n*2)); f32_to_bf16_kernel<<<(n+255)/256,256>>>(d,b,n); CK(hipDeviceSynchronize()); CK(hipFree(d)); return b; };
  const size_t poolA = (size_t)8192*3072;
  bf16* Apool = mk(poolA, 1.0f);
  bf16* A; CK(hipMalloc(&A,(size_t)M*3072*2));
  for(size_t off=0; off<(size_t)M*3072; off+=poolA) CK(hipMemcpy(A+off, Apool, std::min(poolA,(size_t)M*3072-off)*2, hipMemcpyDeviceToDevice));
  bf16* W = mk((size_t)3072*3072, 0.02f);
  float* vec; CK(hipMalloc(&vec, 3072*4*4)); CK(hipMemset(vec,0,3072*16));
  int* total; CK(hipMalloc(&total,4)); CK(hipMemcpy(total,&M,4,hipMemcpyHostToDevice));
  bf16 *q,*k,*vt,*h,*yb,*yb2; float *y,*resid; float2 *stats,*part;
  CK(hipMalloc(&q,(size_t)M*768*2)); CK(hipMalloc(&k,(size_t)M*768*2)); CK(hipMalloc(&vt,(size_t)768*(M+64)*2)); CK(hipMalloc(&h,(size_t)M*3072*2)); CK(hipMalloc(&yb,(size_t)M*768*2)); CK(hipMalloc(&yb2,(size_t)M*768*2)); CK(hipMemset(yb2,0,(size_t)M*768*2));
  CK(hipMalloc(&y,(size_t)M*768*4)); CK(hipMalloc(&resid,(size_t)M*768*4)); CK(hipMemset(resid,0,(size_t)M*768*4));
  CK(hipMalloc(&stats,(size_t)M*8)); CK(hipMalloc(&part,(size_t)M*12*8));
  fill_identity_stats_kernel<<<(M+255)/256,256>>>(stats,(size_t)M); CK(hipDeviceSynchronize());
  if (getenv("G8_RANDOM_EPI")) {   // a correctness run: random residual rows, row statistics and column vectors (the timing runs keep zeros)
    CK(hipMemcpy(yb2, Apool, (size_t)std::min(poolA, (size_t)M*768)*2, hipMemcpyDeviceToDevice));
    for (size_t off = poolA; off < (size_t)M*768; off += poolA) CK(hipMemcpy(yb2+off, Apool, std::min(poolA,(size_t)M*768-off)*2, hipMemcpyDeviceToDevice));
    std::vector<float> hs((size_t)M*2), hv(3072*4); std::uniform_real_distribution<float> ud(0.5f, 1.5f);
    for (size_t i = 0; i < (size_t)M; ++i) { hs[2*i] = nd(rng)*0.3f; hs[2*i+1] = ud(rng); }
    for (auto &v : hv) v = nd(rng)*0.5f;
    CK(hipMemcpy(stats, hs.data(), hs.size()*4, hipMemcpyHostToDevice)); CK(hipMemcpy(vec, hv.data(), hv.size()*4, hipMemcpyHostToDevice));
  }
  Gemm8Args g{}; g.n_groups=1; g.A=A; g.W=W; g.total_rows=total; g.astats=stats; g.wsum=vec; g.cvec=vec+3072; g.q=q; g.k=k; g.v16=vt; g.resid=yb2; g.rstats=stats; g.rgamma=vec+6144; g.rbeta=vec+9216; g.yb=yb; g.part=part; g.h=h;
  struct Cfg{const char* name; int N,K,epi;};
  Cfg cfgs[] = {{"QKV   N=2304 K=768 ",2304,768,EPI8_QKV},{"OUT   N=768  K=768 ",768,768,EPI8_RESID},{"FFN1  N=3072 K=768 ",3072,768,EPI8_GELU},{"FFN2  N=768  K=3072",768,3072,EPI8_RESID}};
  for(auto&c: cfgs){
    g.N=c.N; g.K=c.K; g.n_groups = c.epi==EPI8_GELU ? 2 : 1; float t=0, t0=0;
    if(c.epi==EPI8_QKV) t=run<EPI8_QKV>(g,5,&t0); if(c.epi==EPI8_RESID) t=run<EPI8_RESID>(g,5,&t0); if(c.epi==EPI8_GELU) t=run<EPI8_GELU>(g,5,&t0);
    printf("%s : split %.3f ms %.0f TF | round-2 form %.3f ms %.0f TF\n", c.name, t, 2.0*M*c.N*c.K/t/1e9, t0, 2.0*M*c.N*c.K/t0/1e9);
#ifdef G8_NGSWEEP
    for (int ng : {1, 2, 4}) { if ((c.N / 256) % ng) continue; Gemm8Args gg = g; gg.n_groups = ng; float a0 = 0, a1 = 0;
      if(c.epi==EPI8_QKV) a1=run<EPI8_QKV>(gg,5,&a0); if(c.epi==EPI8_RESID) a1=run<EPI8_RESID>(gg,5,&a0); if(c.epi==EPI8_GELU) a1=run<EPI8_GELU>(gg,5,&a0);
      printf("   n_groups %d: split %.3f ms\n", ng, a1); }
#endif
    {   // same bits from both forms (same MFMA order per output element)
      const size_t nb = c.epi==EPI8_GELU ? (size_t)M*3072*2 : (size_t)M*768*2; bf16* out = c.epi==EPI8_GELU ? h : (c.epi==EPI8_RESID ? yb : k);
      std::vector<unsigned short> a(nb/2), b(nb/2);
      CK(hipMemset(out, 0xff, nb));
      if(c.epi==EPI8_QKV) gemm8_kernel<EPI8_QKV,false><<<GRID,512,131072>>>(g); if(c.epi==EPI8_RESID) gemm8_kernel<EPI8_RESID,false><<<GRID,512,131072>>>(g); if(c.epi==EPI8_GELU) gemm8_kernel<EPI8_GELU,false><<<GRID,512,131072>>>(g);
      CK(hipDeviceSynchronize()); CK(hipMemcpy(a.data(), out, nb, hipMemcpyDeviceToHost));
      CK(hipMemset(out, 0xff, nb));
      if(c.epi==EPI8_QKV) gemm8_kernel<EPI8_QKV,true><<<GRID,512,163840>>>(g); if(c.epi==EPI8_RESID) gemm8_kernel<EPI8_RESID,true><<<GRID,512,163840>>>(g); if(c.epi==EPI8_GELU) gemm8_kernel<EPI8_GELU,true><<<GRID,512,163840>>>(g);
      CK(hipDeviceSynchronize()); CK(hipMemcpy(b.data(), out, nb, hipMemcpyDeviceToHost));
      size_t diff = 0, big = 0; double worst = 0;
      auto tof = [](unsigned short h){ unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
      for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) { ++diff; const float fa = tof(a[i]), fb = tof(b[i]); const double rel = fabs(fa - fb) / (fabs(fa) + fabs(fb) + 1e-30);
        worst = std::max(worst, rel); big += rel > 0.005; }      // one bf16 ulp is a relative 2^-8 .. 2^-7 of the value: rel (of the sum) <= 0.004
      printf("   outputs of the two forms differ in %zu of %zu elements (more than one bf16 ulp apart: %zu, worst relative difference %.4f)\n", diff, a.size(), big, worst); }
#ifdef G8_STAMP2
    for (int form = 0; form < 2; ++form) {
      CK(hipMemset(part, 0, 2048));
      if (form) { if(c.epi==EPI8_QKV) gemm8_kernel<EPI8_QKV,true><<<GRID,512,163840>>>(g); if(c.epi==EPI8_RESID) gemm8_kernel<EPI8_RESID,true><<<GRID,512,163840>>>(g); if(c.epi==EPI8_GELU) gemm8_kernel<EPI8_GELU,true><<<GRID,512,163840>>>(g); }
      else { if(c.epi==EPI8_QKV) gemm8_kernel<EPI8_QKV,false><<<GRID,512,131072>>>(g); if(c.epi==EPI8_RESID) gemm8_kernel<EPI8_RESID,false><<<GRID,512,131072>>>(g); if(c.epi==EPI8_GELU) gemm8_kernel<EPI8_GELU,false><<<GRID,512,131072>>>(g); }
      CK(hipDeviceSynchronize());
      unsigned long long hs[96]; CK(hipMemcpy(hs, part, sizeof hs, hipMemcpyDeviceToHost));
      for (int gq = 0; gq < 2; ++gq) { unsigned long long* h = hs + 64 + gq*16;
        if (form) printf("   split   group %d: R1 pieces %llu reads %llu | R2 pieces %llu reads %llu\n", gq, h[12]-h[0], h[1]-h[12], h[13]-h[6], h[7]-h[13]);
        printf("   %s group %d k-tile 6 of tile 3: R1 reads %llu | stage %llu | lgkm wait %llu | barrier %llu | M1 %llu | barrier %llu | R2 reads %llu | stage+waits %llu | barrier %llu | M2 %llu | barrier %llu | total %llu\n",
          form ? "split  " : "round-2", gq, h[1]-h[0], h[2]-h[1], h[3]-h[2], h[4]-h[3], h[5]-h[4], h[6]-h[5], h[7]-h[6], h[8]-h[7], h[9]-h[8], h[10]-h[9], h[11]-h[10], h[11]-h[0]); } }
#endif
#ifdef G8_STAMP
    if (c.epi==EPI8_RESID) { unsigned long long hs[64]; CK(hipMemcpy(hs, part, sizeof hs, hipMemcpyDeviceToHost));
      for (int gq = 0; gq < 2; ++gq) { unsigned long long* e = hs + 32 + gq*16; if (!e[0]) continue; unsigned long long* h = hs + gq*16;
        printf("   group %d LDS epilogue (cycles from the k-loop's end): sb3 sb4 issued %llu | stats cols sb0 landed %llu | sb0 + S0 done %llu | barrier + A staged %llu | sb1 ready %llu | sb2 ready %llu | sb3 ready %llu | sb4 ready %llu | sb5 ready %llu | sb6 ready %llu | sb6 done %llu | barrier + W staged %llu | sb7 ready %llu | sb7 + S7 done %llu | A / W landed %llu\n", gq,
          e[0]-h[1], e[1]-h[1], e[2]-h[1], e[3]-h[1], e[4]-h[1], e[5]-h[1], e[6]-h[1], e[7]-h[1], e[8]-h[1], e[9]-h[1], e[10]-h[1], e[11]-h[1], e[12]-h[1], e[13]-h[1], e[14]-h[1]); } }
    { unsigned long long hs[64]; CK(hipMemcpy(hs, part, sizeof hs, hipMemcpyDeviceToHost));
      for (int gq = 0; gq < 2; ++gq) { unsigned long long* h = hs + gq*16; if (h[10]) printf("   group %d RESID epilogue: loads issued %llu | band0 wait+compute %llu | band1 load+compute %llu | stores issued %llu\n", gq, h[10]-h[1], h[11]-h[10], h[12]-h[11], h[13]-h[12]); printf("   group %d: kloop-end->aligned %llu | +2 stages & drain %llu | epilogue issue %llu | ->k0 barrier %llu | k0->k1 %llu | k1->k2 %llu | k2->k3 %llu\n", gq,
        h[0]-h[8], h[1]-h[0], h[2]-h[1], h[3]-h[2], h[4]-h[3], h[5]-h[4], h[6]-h[5]); } }
#endif
  }
  return 0;
}
namespace hac { std::string &last_error_slot(){ static std::string s; return s; } int fail(int code, const char *fmt, ...){ (void)fmt; return code; } }
