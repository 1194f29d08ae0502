// Probe: HBM read bandwidth with the "tiled group" access pattern: each wave streams 192 x 1KiB chunks per 64-row group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template<int PF>
__global__ void stream(const float4* __restrict__ x, float* out, long ngroups, int P, int W){
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float4 s = {0,0,0,0};
  float4 a[PF];
  long g0 = (long)blockIdx.x*W + w;
  long stride = (long)P*W;
  if (g0 >= ngroups) return;
  const float4* gp = x + g0*12288 + lane;
  #pragma unroll
  for(int i=0;i<PF;i++) a[i]=gp[i*64];
  for(long g=g0; g<ngroups; g+=stride){
    const float4* np = (g+stride<ngroups) ? x + (g+stride)*12288 + lane : gp;
    for(int tb=0; tb<192/PF-1; tb++){
      #pragma unroll
      for(int i=0;i<PF;i++){ float4 v=a[i]; a[i]=gp[(tb*PF+i+PF)*64]; s.x+=v.x; s.y+=v.y; s.z+=v.z; s.w+=v.w; }
    }
    #pragma unroll
    for(int i=0;i<PF;i++){ float4 v=a[i]; a[i]=np[i*64]; s.x+=v.x; s.y+=v.y; s.z+=v.z; s.w+=v.w; }
    gp = np;
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=s.x+s.y+s.z+s.w;
}
template<int PF> void run(const float4* x, float* out, long ngroups, int P, int W){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  stream<PF><<<P,W*64>>>(x,out,ngroups,P,W); CK(hipDeviceSynchronize());
  float best=1e9;
  for(int r=0;r<3;r++){ CK(hipEventRecord(e0)); stream<PF><<<P,W*64>>>(x,out,ngroups,P,W); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
  double bytes=(double)ngroups*196608.0;
  printf("PF=%2d P=%4d W=%2d : %.3f ms  %.2f TB/s\n", PF,P,W,best,bytes/(best*1e-3)/1e12);
}
int main(){
  long nrows = 1000000; long ngroups=(nrows+63)/64; size_t bytes=ngroups*196608;
  float4* x; CK(hipMalloc(&x,bytes)); CK(hipMemset(x,0,bytes)); float* out; CK(hipMalloc(&out,4096*1024*4));
  int Ps[]={256,512,1024,2048}; int Ws[]={4,8};
  for(int W:Ws) for(int P:Ps){ if(P*W*64>4096*1024) continue; run<4>(x,out,ngroups,P,W); run<8>(x,out,ngroups,P,W); run<16>(x,out,ngroups,P,W); }
  return 0;
}
