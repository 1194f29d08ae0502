// Probe: exactness + issue rate of the f32 multi-block MFMA forms on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// A: rows[64][K] (lane l -> row l), B: q[16][K] (lane l -> query l&15). out[64 rows][16 queries]
__global__ void k16(const float* rows, const float* q, float* out, int K){
  int lane = threadIdx.x;
  f32x16 acc = {0};
  for(int k=0;k<K;k++){
    float a = rows[lane*K+k]; float b = q[(lane&15)*K+k];
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a,b,acc,0,0,0);
  }
  for(int r=0;r<16;r++){
    int row = 16*(r>>2) + 4*(lane>>4) + (r&3);
    out[row*16 + (lane&15)] = acc[r];
  }
}
// 32x32x1_2b: A rows[64][K] lane l -> row l (block l>>5, i=l&31); B q[32][K] lane l-> query l&31
__global__ void k32(const float* rows, const float* q, float* out, int K){
  int lane = threadIdx.x;
  f32x32 acc = {0};
  for(int k=0;k<K;k++){
    float a = rows[lane*K+k]; float b = q[(lane&31)*K+k];
    acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a,b,acc,0,0,0);
  }
  for(int r=0;r<32;r++){
    int blk = r>>4; int rr = r&15;
    int row = 32*blk + (rr&3) + 8*(rr>>2) + 4*(lane>>5);
    out[row*32 + (lane&31)] = acc[r];
  }
}
// 16x16x4: A rows[16][K], lane l: row l&15, k = 4s + (l>>4)
__global__ void k16x4(const float* rows, const float* q, float* out, int K){
  int lane = threadIdx.x;
  f32x4 acc = {0};
  for(int s=0;s<K/4;s++){
    float a = rows[(lane&15)*K+4*s+(lane>>4)]; float b = q[(lane&15)*K+4*s+(lane>>4)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a,b,acc,0,0,0);
  }
  for(int r=0;r<4;r++){ int row = 4*(lane>>4)+r; out[row*16+(lane&15)] = acc[r]; }
}

template<int MODE, int NCHAIN>
__global__ void rate(float* out, int iters, float a0, float b0){
  // MODE 0: 16x16x1_4b, 1: 32x32x1_2b, 2: 16x16x4, 3: 32x32x2
  float a = a0 + threadIdx.x*1e-3f, b = b0;
  if constexpr (MODE==0){
    f32x16 acc[NCHAIN]; for(int c=0;c<NCHAIN;c++) acc[c]=(f32x16){0};
    for(int i=0;i<iters;i++){
      #pragma unroll
      for(int c=0;c<NCHAIN;c++) acc[c]=__builtin_amdgcn_mfma_f32_16x16x1f32(a,b,acc[c],0,0,0);
    }
    float s=0; for(int c=0;c<NCHAIN;c++) for(int r=0;r<16;r++) s+=acc[c][r]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  } else if constexpr (MODE==1){
    f32x32 acc[NCHAIN]; for(int c=0;c<NCHAIN;c++) acc[c]=(f32x32){0};
    for(int i=0;i<iters;i++){
      #pragma unroll
      for(int c=0;c<NCHAIN;c++) acc[c]=__builtin_amdgcn_mfma_f32_32x32x1f32(a,b,acc[c],0,0,0);
    }
    float s=0; for(int c=0;c<NCHAIN;c++) for(int r=0;r<32;r++) s+=acc[c][r]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  } else if constexpr (MODE==2){
    f32x4 acc[NCHAIN]; for(int c=0;c<NCHAIN;c++) acc[c]=(f32x4){0};
    for(int i=0;i<iters;i++){
      #pragma unroll
      for(int c=0;c<NCHAIN;c++) acc[c]=__builtin_amdgcn_mfma_f32_16x16x4f32(a,b,acc[c],0,0,0);
    }
    float s=0; for(int c=0;c<NCHAIN;c++) for(int r=0;r<4;r++) s+=acc[c][r]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  } else {
    f32x16 acc[NCHAIN]; for(int c=0;c<NCHAIN;c++) acc[c]=(f32x16){0};
    for(int i=0;i<iters;i++){
      #pragma unroll
      for(int c=0;c<NCHAIN;c++) acc[c]=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,acc[c],0,0,0);
    }
    float s=0; for(int c=0;c<NCHAIN;c++) for(int r=0;r<16;r++) s+=acc[c][r]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  }
}

template<int MODE,int NCHAIN>
void bench(const char* name, int waves_per_cu, double flops_per_mfma){
  int iters = 20000;
  int threads = 64*waves_per_cu; int blocks = 256;
  float* out; CK(hipMalloc(&out, blocks*threads*4));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate<MODE,NCHAIN><<<blocks,threads>>>(out, 100, 1.0f, 1e-6f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rate<MODE,NCHAIN><<<blocks,threads>>>(out, iters, 1.0f, 1e-6f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  double n_mfma = (double)blocks*waves_per_cu*iters*NCHAIN;
  double tf = n_mfma*flops_per_mfma/(ms*1e-3)/1e12;
  // cycles per mfma per SIMD assuming 2.4GHz: waves per SIMD = waves_per_cu/4
  double cyc = (ms*1e-3*2.4e9) / ((double)iters*NCHAIN*(waves_per_cu/4.0 > 1 ? waves_per_cu/4.0 : 1));
  printf("%-14s chains=%d waves/CU=%d : %.3f ms  %.1f TF  (~%.1f cyc/mfma/SIMD @2.4GHz)\n", name, NCHAIN, waves_per_cu, ms, tf, cyc);
  CK(hipFree(out));
}

int main(){
  const int K=768;
  std::vector<float> rows(64*K), q(32*K);
  srand(1);
  for(auto&v:rows) v = (float)((rand()/(double)RAND_MAX*2-1)*1.7);
  for(auto&v:q) v = (float)((rand()/(double)RAND_MAX*2-1)*1.7);
  // a few denormal / large values
  rows[5]=1e-41f; q[5]=3.0f; rows[K+7]=1e30f; q[K+7]=1e-30f;
  float *dr,*dq,*dout; CK(hipMalloc(&dr,rows.size()*4)); CK(hipMalloc(&dq,q.size()*4)); CK(hipMalloc(&dout,64*32*4));
  CK(hipMemcpy(dr,rows.data(),rows.size()*4,hipMemcpyHostToDevice)); CK(hipMemcpy(dq,q.data(),q.size()*4,hipMemcpyHostToDevice));
  std::vector<float> out(64*32);
  auto ref=[&](int row,int qq){ float acc=0; for(int k=0;k<K;k++) acc=fmaf(rows[row*K+k],q[qq*K+k],acc); return acc; };
  auto refsep=[&](int row,int qq){ float acc=0; for(int k=0;k<K;k++){ volatile float p=rows[row*K+k]*q[qq*K+k]; acc=acc+p;} return acc; };
  {
    k16<<<1,64>>>(dr,dq,dout,K); CK(hipDeviceSynchronize()); CK(hipMemcpy(out.data(),dout,64*16*4,hipMemcpyDeviceToHost));
    int bad=0,badsep=0; for(int r=0;r<64;r++)for(int c=0;c<16;c++){ float g=out[r*16+c]; float e=ref(r,c); if(memcmp(&g,&e,4)) bad++; float e2=refsep(r,c); if(memcmp(&g,&e2,4)) badsep++; }
    printf("16x16x1_4b vs fmaf chain: %d/1024 mismatches (vs mul+add: %d)\n", bad, badsep);
  }
  {
    k32<<<1,64>>>(dr,dq,dout,K); CK(hipDeviceSynchronize()); CK(hipMemcpy(out.data(),dout,64*32*4,hipMemcpyDeviceToHost));
    int bad=0; for(int r=0;r<64;r++)for(int c=0;c<32;c++){ float g=out[r*32+c]; float e=ref(r,c); if(memcmp(&g,&e,4)) bad++; }
    printf("32x32x1_2b vs fmaf chain: %d/2048 mismatches\n", bad);
  }
  {
    k16x4<<<1,64>>>(dr,dq,dout,K); CK(hipDeviceSynchronize()); CK(hipMemcpy(out.data(),dout,16*16*4,hipMemcpyDeviceToHost));
    int bad=0; for(int r=0;r<16;r++)for(int c=0;c<16;c++){ float g=out[r*16+c]; float e=ref(r,c); if(memcmp(&g,&e,4)) bad++; }
    printf("16x16x4 vs fmaf chain: %d/256 mismatches\n", bad);
  }
  bench<0,1>("16x16x1_4b",4, 2.0*4*16*16); bench<0,2>("16x16x1_4b",4, 2.0*4*16*16); bench<0,1>("16x16x1_4b",8, 2.0*4*16*16);  bench<0,1>("16x16x1_4b",16, 2.0*4*16*16);
  bench<1,1>("32x32x1_2b",4, 2.0*2*32*32); bench<1,2>("32x32x1_2b",4, 2.0*2*32*32); bench<1,1>("32x32x1_2b",8, 2.0*2*32*32);
  bench<2,1>("16x16x4",4, 2.0*16*16*4); bench<2,2>("16x16x4",4, 2.0*16*16*4); bench<2,1>("16x16x4",8, 2.0*16*16*4);
  bench<3,1>("32x32x2",4, 2.0*32*32*2); bench<3,2>("32x32x2",4, 2.0*32*32*2);
  return 0;
}
