// Probe of the streaming attention kernels (haconvdr_amd/csrc/encoder.hip: attention_stream_kernel<16 | 8>) on 512 sequences of
// L rows with gaussian q (scaled by argv[2], default log2(e)/8 = what the QKV epilogue folds in; 1.0 = logits with a standard
// deviation of 8: the reference is raised on most steps), k, v: timing, and with -DATT_STAMP in-kernel s_memtime stamps of one
// workgroup's 4th item (barrier wait, DMA issue, arithmetic per chunk).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DATT_STAMP] tools/probes/attn_stream_probe.hip -o attn_stream_probe; ./attn_stream_probe 512 0.18
#include "../../haconvdr_amd/csrc/encoder.hip"
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} }while(0)
namespace hac { std::string &last_error_slot(){ static std::string s; return s; } int fail(int code, const char *fmt, ...){ (void)fmt; return code; } }
using namespace hac;
int main(int argc, char** argv){
  const int B = 512, L = argc > 1 ? atoi(argv[1]) : 512;
  const int M = B * L;
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f,1.f);
  auto mk = [&](size_t n, float sc){ std::vector<float> hh(n); for(auto&v:hh) v=nd(rng)*sc; float* d; CK(hipMalloc(&d,n*4)); CK(hipMemcpy(d,hh.data(),n*4,hipMemcpyHostToDevice)); bf16* b; CK(hipMalloc(&b,n*2)); f32_to_bf16_kernel<<<(n+255)/256,256>>>(d,b,n); CK(hipDeviceSynchronize()); CK(hipFree(d)); return b; };
  const size_t pool = (size_t)8192*768;
  const float qs = argc > 2 ? atof(argv[2]) : 0.18f;   // log2(e)/8: what the QKV epilogue folds into Q
  bf16* P = mk(pool, 1.0f);
  bf16* PQ = mk(pool, qs);
  bf16 *q,*k,*vt,*ctx;
  CK(hipMalloc(&q,(size_t)M*768*2)); CK(hipMalloc(&k,(size_t)(M+64)*768*2)); CK(hipMalloc(&vt,(size_t)768*(M+64)*2)); CK(hipMalloc(&ctx,(size_t)M*768*2));
  for(size_t off=0; off<(size_t)M*768; off+=pool){ size_t n=std::min(pool,(size_t)M*768-off)*2; CK(hipMemcpy(q+off,PQ,n,hipMemcpyDeviceToDevice)); CK(hipMemcpy(k+off,P,n,hipMemcpyDeviceToDevice)); CK(hipMemcpy(vt+off,P,n,hipMemcpyDeviceToDevice)); }
  std::vector<int> lens(B,L), len32(B,L), off(B+1), order(B), ncls(2);
  for(int i=0;i<=B;i++) off[i]=i*L;
  for(int i=0;i<B;i++) order[i]=i;
  ncls[0] = L > 256 ? B : 0; ncls[1] = L > 256 ? 0 : B;
  SeqInfo s{}; CK(hipMalloc(&s.desc,(size_t)B*16));
  { std::vector<int> d((size_t)B*4); for (int i=0;i<B;i++){ d[4*i]=L; d[4*i+1]=L; d[4*i+2]=i*L; d[4*i+3]=i; } CK(hipMemcpy(s.desc,d.data(),(size_t)B*16,hipMemcpyHostToDevice)); }
  CK(hipMalloc(&s.lens,B*4)); CK(hipMalloc(&s.len32,B*4)); CK(hipMalloc(&s.off,(B+1)*4)); CK(hipMalloc(&s.order,B*4)); CK(hipMalloc(&s.ncls,8));
  CK(hipMemcpy(s.lens,lens.data(),B*4,hipMemcpyHostToDevice)); CK(hipMemcpy(s.len32,len32.data(),B*4,hipMemcpyHostToDevice)); CK(hipMemcpy(s.off,off.data(),(B+1)*4,hipMemcpyHostToDevice));
  CK(hipMemcpy(s.order,order.data(),B*4,hipMemcpyHostToDevice)); CK(hipMemcpy(s.ncls,ncls.data(),8,hipMemcpyHostToDevice));
  AttnArgs a{}; a.q=q; a.k=k; a.v16=vt; a.ctx=ctx; a.s=s; a.cls_only=0; a.qsplit=1; a.one_class=0;
  const int form = argc > 3 ? atoi(argv[3]) : 0;   // 0: one block per wave (attention_stream_kernel), 1: two blocks per wave, woven (attention_pipe_kernel)
  int *redo; CK(hipMalloc(&redo, (size_t)(B * 12 + 16) * 4)); CK(hipMemset(redo, 0, (size_t)(B * 12 + 16) * 4));
  a.redo_count = redo; a.redo_flags = redo + 16; a.fixup = 0;
  CK(hipFuncSetAttribute((const void *)attention_pipe_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  CK(hipFuncSetAttribute((const void *)attention_pipe_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  CK(hipFuncSetAttribute((const void *)attention_stream_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  CK(hipFuncSetAttribute((const void *)attention_stream_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  auto launch=[&]{
    if (form == 1) { if (L > 256) attention_pipe_kernel<8><<<256,512,163840>>>(a); else attention_pipe_kernel<4><<<512,256,81920>>>(a); }
    else { if (L > 256) attention_stream_kernel<16><<<256,1024,163840>>>(a); else attention_stream_kernel<8><<<512,512,81920>>>(a); } };
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int i=0;i<2;i++) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int it=10;
  for(int i=0;i<it;i++) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=it;
  double fl = 4.0*B*12*(double)L*L*64;
  { int h[16]; CK(hipMemcpy(h, redo, 64, hipMemcpyDeviceToHost)); printf("form %d (flagged items %d) ", form, h[0] / (it + 2)); }
  printf("L=%d qscale %.2f: %.3f ms  %.0f TF (useful)  %.0f cycles@2GHz per item per CU\n", L, qs, ms, fl/ms/1e9, ms*1e-3*2e9/(B*12/(L>256?256.0:512.0)));
#ifdef ATTP_STAMP
  if (form == 1) { unsigned long long st[512]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(hac::g_attp_stamps), sizeof(st)));
    for (int w : {0, 3, 4, 7}) { const unsigned long long *q = st + w * 64;
      printf(" wave %d: wait+barrier %lld | to prologue %lld | first pair %lld | loop iterations [Y + X]:", w, (long long)(q[1]-q[0]), (long long)(q[2]-q[1]), (long long)(q[4]-q[2]));
      for (int i = 1; i < 14; ++i) if (q[4+i] > q[3+i]) printf(" %lld[%lld+%lld]", (long long)(q[4+i]-q[3+i]), (long long)(q[44+i]-q[3+i]), (long long)(q[4+i]-q[44+i]));
      printf(" | boundaries (wait+barrier):"); for (int c = 1; c < 4; ++c) printf(" %lld", (long long)(q[21+2*c]-q[20+2*c]));
      printf(" | last PV + stores %lld | item %lld\n", (long long)(q[19]-q[18]), (long long)(q[19]-q[0])); }
    for (int w : {0, 4}) { const unsigned long long *q = st + w * 64; const unsigned long long t0 = st[0];
      printf(" wave %d timeline (cycles from wave 0's item start): start %lld, past first barrier %lld, prologue %lld, iterations at", w, (long long)(q[0]-t0), (long long)(q[1]-t0), (long long)(q[2]-t0));
      for (int i = 1; i < 14; ++i) printf(" %lld", (long long)(q[3+i]-t0));
      printf(" | boundary waits begin/end:"); for (int c = 1; c < 4; ++c) printf(" %lld-%lld", (long long)(q[20+2*c]-t0), (long long)(q[21+2*c]-t0));
      printf(" | drain done %lld, item done %lld\n", (long long)(q[18]-t0), (long long)(q[19]-t0)); } }
#endif
#ifdef ATT_STAMP
  { unsigned long long st[512]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(hac::g_att_stamps), sizeof(st)));
    const int nch = L > 256 ? (L+127)/128 : (L+63)/64;
    for (int w : {0, 1, 5, 10, 15}) { if (w >= (L > 256 ? 16 : 8)) continue; printf(" wave %2d:", w);
      for (int c = 0; c < nch; ++c) printf(" [c%d wait+bar %lld dma %lld steps %lld]", c, (long long)(st[w*32+2+4*c]-st[w*32+1+4*c]), (long long)(st[w*32+3+4*c]-st[w*32+2+4*c]), (long long)(st[w*32+4+4*c]-st[w*32+3+4*c]));
      printf(" stores %lld total %lld\n", (long long)(st[w*32+20]-st[w*32+4*nch]), (long long)(st[w*32+20]-st[w*32])); } }
#endif
  return 0;
}
