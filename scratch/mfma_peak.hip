// Sustained MFMA rate under power: 4 waves/CU (one per SIMD), dependent 32x32x16 bf16 chains, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a,b,c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a),(b),(c),0,0,0)
template<int NACC>
__global__ __launch_bounds__(256,1) void k(const float* seed, float* out, int iters) {
  const int tid = blockIdx.x*256 + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int q=0;q<4;q++) for (int e=0;e<8;e++){ a[q][e]=(__bf16)(seed[(tid*64+q*16+e)&65535]); b[q][e]=(__bf16)(seed[(tid*64+q*16+8+e)&65535]); }
  f32x16 acc[NACC]; for (int i=0;i<NACC;i++) acc[i] = f32x16{0};
  for (int it=0; it<iters; ++it) {
    #pragma unroll
    for (int j=0;j<16;j++) {
      #pragma unroll
      for (int i=0;i<NACC;i++) acc[i] = MFMA(a[(j+i)&3], b[(j*3+i)&3], acc[i]);
    }
  }
  float s=0; for (int i=0;i<NACC;i++) for (int r=0;r<16;r++) s+=acc[i][r];
  out[tid]=s;
}
template<int NACC> void run(const float* d_seed, float* d_out, int iters, const char* name){
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, d_seed, d_out, iters/10); hipDeviceSynchronize();
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, d_seed, d_out, iters); hipEventRecord(e1);
  hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms,e0,e1);
  double mf = 1024.0 * iters * 16.0 * NACC;   // MFMAs total (1024 waves)
  printf("%-28s %.2f ms  %.1f TFLOP/s issued  (%.1f%% of 2500)  cycles/MFMA@2.4GHz=%.1f\n", name, ms, mf*32768/ (ms*1e-3)/1e12, mf*32768/(ms*1e-3)/1e12/25, ms*1e-3*2.4e9/(iters*16.0*NACC));
}
int main(){
  std::vector<float> h(65536); srand(1); for (auto& x: h) x = ((float)rand()/RAND_MAX*2-1)*0.05f;
  std::vector<float> z(65536, 0.f);
  float *d_seed, *d_zero, *d_out; hipMalloc(&d_seed, 65536*4); hipMalloc(&d_zero, 65536*4); hipMalloc(&d_out, 256*256*4);
  hipMemcpy(d_seed, h.data(), 65536*4, hipMemcpyHostToDevice); hipMemcpy(d_zero, z.data(), 65536*4, hipMemcpyHostToDevice);
  run<1>(d_seed, d_out, 30000, "1 acc dependent, random");
  run<2>(d_seed, d_out, 15000, "2 accs, random");
  run<4>(d_seed, d_out, 8000,  "4 accs, random");
  run<1>(d_zero, d_out, 30000, "1 acc dependent, zeros");
  run<4>(d_zero, d_out, 8000,  "4 accs, zeros");
  return 0;
}
