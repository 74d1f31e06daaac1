// Microbenchmark + layout probe for v_mfma_f64_16x16x4_f64 on gfx950 (MI355X).
// Measures the dense FP64 MFMA ceiling that every roofline fraction in this
// repo is priced against, and verifies the A/B/C/D fragment layout and the
// neg:[a,b,c] modifier (BLGP bits on the f64 MFMA) that the complex GEMM uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template<int NACC>
__global__ __launch_bounds__(256) void cyc_kernel(double* out, long long* cyc, int iters, double a0, double b0){
  d4 acc[NACC];
  #pragma unroll
  for(int i=0;i<NACC;i++) acc[i] = d4{0,0,0,0};
  double a = a0 + threadIdx.x*1e-9, b = b0;
  long long t0 = __builtin_readcyclecounter();
  for(int it=0; it<iters; ++it){
    #pragma unroll
    for(int i=0;i<NACC;i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0,0,0);
  }
  double s=0;
  #pragma unroll
  for(int i=0;i<NACC;i++) s += acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
  if((threadIdx.x&63)==0) cyc[blockIdx.x*4 + (threadIdx.x>>6)] = t1-t0;
}

template<int NACC>
__global__ __launch_bounds__(256) void peak_kernel(double* out, int iters, double a0, double b0){
  d4 acc[NACC];
  #pragma unroll
  for(int i=0;i<NACC;i++) acc[i] = d4{0,0,0,0};
  double a = a0 + threadIdx.x*1e-9, b = b0;
  for(int it=0; it<iters; ++it){
    #pragma unroll
    for(int i=0;i<NACC;i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0,0,0);
  }
  double s=0;
  #pragma unroll
  for(int i=0;i<NACC;i++) s += acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
}

__global__ void layout_kernel(const double* A, const double* B, double* C, double* Cneg){
  // A is 16x4 row-major (i,k), B is 4x16 row-major (k,j); C 16x16 row-major.
  int l = threadIdx.x;
  double a = A[(l&15)*4 + (l>>4)];
  double b = B[(l>>4)*16 + (l&15)];
  d4 acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0,0,0);
  for(int r=0;r<4;r++) C[((l>>4)+4*r)*16 + (l&15)] = acc[r];
  d4 acc2 = {1,1,1,1};
  acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0,0,1);  // neg A
  for(int r=0;r<4;r++) Cneg[((l>>4)+4*r)*16 + (l&15)] = acc2[r];
}

template<int NACC> double run_peak(int blocks, int iters){
  double* out; CK(hipMalloc(&out, sizeof(double)*blocks*256));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  peak_kernel<NACC><<<blocks,256>>>(out, 10, 1.0, 1.0);
  CK(hipDeviceSynchronize());
  double best=0;
  for(int rep=0;rep<5;rep++){
    CK(hipEventRecord(e0));
    peak_kernel<NACC><<<blocks,256>>>(out, iters, 1.000001, 0.999999);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    double flops = (double)blocks*4 /*waves*/ * iters * NACC * 2.0*16*16*4;
    double tf = flops/(ms*1e-3)/1e12;
    if(tf>best) best=tf;
  }
  CK(hipFree(out));
  return best;
}

// Hand-pinned variant: 4x4 accumulators in VGPRs, distinct A/B operands, MFMAs issued from inline asm so
// that hipcc cannot shuttle the accumulators through AGPRs between iterations (which is what limits the
// intrinsic-based kernels above).  This is the ceiling a register-resident GEMM inner loop can see.
__global__ __launch_bounds__(256) void asm_kernel(double* out, long long* cyc, int iters, double a0, double b0){
  d4 acc[4][4];
  #pragma unroll
  for(int i=0;i<4;i++)
  #pragma unroll
  for(int j=0;j<4;j++) acc[i][j] = d4{0,0,0,0};
  double a[4], b[4];
  #pragma unroll
  for(int i=0;i<4;i++){ a[i] = a0 + threadIdx.x*1e-9 + i*1e-3; b[i] = b0 - i*1e-3; }
  long long t0 = __builtin_readcyclecounter();
  for(int it=0; it<iters; ++it){
    #pragma unroll
    for(int i=0;i<4;i++)
    #pragma unroll
    for(int j=0;j<4;j++)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  double s=0;
  #pragma unroll
  for(int i=0;i<4;i++)
  #pragma unroll
  for(int j=0;j<4;j++) s += acc[i][j][0]+acc[i][j][1]+acc[i][j][2]+acc[i][j][3];
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
  if((threadIdx.x&63)==0) cyc[blockIdx.x*4 + (threadIdx.x>>6)] = t1-t0;
}

void run_asm(int blocks, int iters, double a0, double b0, const char* tag){
  double* out; long long* cyc; CK(hipMalloc(&out, sizeof(double)*blocks*256)); CK(hipMalloc(&cyc, sizeof(long long)*blocks*4));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  asm_kernel<<<blocks,256>>>(out, cyc, 10, a0, b0); CK(hipDeviceSynchronize());
  double best=0, bper=0, bclk=0;
  for(int rep=0;rep<3;rep++){
    CK(hipEventRecord(e0));
    asm_kernel<<<blocks,256>>>(out, cyc, iters, a0, b0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    std::vector<long long> h(blocks*4); CK(hipMemcpy(h.data(), cyc, sizeof(long long)*blocks*4, hipMemcpyDeviceToHost));
    double avg=0; for(auto v: h) avg += v; avg/=h.size();
    double tf = (double)blocks*4*iters*16*2048.0/(ms*1e-3)/1e12;
    if(tf>best){best=tf; bper=avg/((double)iters*16); bclk=avg/(ms*1e3);}
  }
  printf("asm-pinned %s blocks=%d: %.2f TF, ticks per MFMA per wave %.1f, ticks/us %.1f\n", tag, blocks, best, bper, bclk);
  CK(hipFree(out)); CK(hipFree(cyc));
}

template<int NACC> void run_cyc(int blocks, int iters, double a0, double b0, const char* tag){
  double* out; long long* cyc; CK(hipMalloc(&out, sizeof(double)*blocks*256)); CK(hipMalloc(&cyc, sizeof(long long)*blocks*4));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  cyc_kernel<NACC><<<blocks,256>>>(out, cyc, 10, a0, b0); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  cyc_kernel<NACC><<<blocks,256>>>(out, cyc, iters, a0, b0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  std::vector<long long> h(blocks*4); CK(hipMemcpy(h.data(), cyc, sizeof(long long)*blocks*4, hipMemcpyDeviceToHost));
  double avg=0; for(auto v: h) avg += v; avg/=h.size();
  double per = avg/((double)iters*NACC);
  double tf = (double)blocks*4*iters*NACC*2048.0/(ms*1e-3)/1e12;
  printf("%s blocks=%d NACC=%d: %.2f TF, s_memtime ticks per MFMA per wave %.1f, ticks/us %.1f\n", tag, blocks, NACC, tf, per, avg/(ms*1e3));
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  // layout
  std::vector<double> A(64),B(64),C(256),Cn(256),R(256,0.0);
  for(int i=0;i<64;i++){A[i]=std::sin(1.0+i*0.37); B[i]=std::cos(0.3+i*0.91);}
  for(int i=0;i<16;i++)for(int j=0;j<16;j++){double s=0;for(int k=0;k<4;k++)s+=A[i*4+k]*B[k*16+j];R[i*16+j]=s;}
  double *dA,*dB,*dC,*dCn; CK(hipMalloc(&dA,512));CK(hipMalloc(&dB,512));CK(hipMalloc(&dC,2048));CK(hipMalloc(&dCn,2048));
  CK(hipMemcpy(dA,A.data(),512,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),512,hipMemcpyHostToDevice));
  layout_kernel<<<1,64>>>(dA,dB,dC,dCn); CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(),dC,2048,hipMemcpyDeviceToHost));CK(hipMemcpy(Cn.data(),dCn,2048,hipMemcpyDeviceToHost));
  double e1=0,e2=0; for(int i=0;i<256;i++){e1=fmax(e1,fabs(C[i]-R[i])); e2=fmax(e2,fabs(Cn[i]-(1.0-R[i])));}
  printf("layout max err %.3e   neg-A (blgp=1) max err %.3e\n", e1, e2);
  // peak
  for(int bpc : {1,2,4}){
    int blocks = p.multiProcessorCount*bpc;
    printf("blocks/CU=%d  NACC=1: %.2f TF  NACC=2: %.2f TF  NACC=4: %.2f TF  NACC=8: %.2f TF  NACC=16: %.2f TF\n", bpc,
      run_peak<1>(blocks,20000), run_peak<2>(blocks,10000), run_peak<4>(blocks,5000), run_peak<8>(blocks,2500), run_peak<16>(blocks,1250));
  }
  int cus = p.multiProcessorCount;
  run_cyc<8>(cus, 4000, 1.000001, 0.999999, "1wave/SIMD nonzero");
  run_cyc<8>(cus*2, 4000, 1.000001, 0.999999, "2wave/SIMD nonzero");
  run_cyc<8>(cus*2, 4000, 0.0, 0.0, "2wave/SIMD zeros  ");
  run_cyc<8>(cus*2, 4000, 0.7312893127, -1.3371237, "2wave/SIMD random-ish");
  run_asm(cus, 2000, 1.000001, 0.999999, "1 wave/SIMD ");
  run_asm(cus*2, 2000, 1.000001, 0.999999, "2 waves/SIMD");
  run_asm(cus*2, 2000, 0.731, -1.337, "2 waves/SIMD random-ish");
  run_asm(cus*4, 2000, 1.000001, 0.999999, "4 waves/SIMD");
  run_asm(1, 2000, 1.000001, 0.999999, "single block");
  run_cyc<8>(1, 4000, 1.000001, 0.999999, "single block     ");
  run_cyc<8>(8, 4000, 1.000001, 0.999999, "8 blocks         ");
  run_cyc<8>(64, 4000, 1.000001, 0.999999, "64 blocks        ");
  return 0;
}
