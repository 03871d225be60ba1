// Micro-benchmarks of MFMA / VALU / LDS interleave patterns, one wave per SIMD (256-thread WG, 1 WG per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a,b,c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a),(b),(c),0,0,0)
#define SB __builtin_amdgcn_sched_barrier(0)

template<int PAT>
__global__ __launch_bounds__(256,1) void k(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  bf16x8 a, b; for (int e=0;e<8;e++){ a[e]=(__bf16)(0.01f*(lane+e)); b[e]=(__bf16)(0.02f*(lane-e)); }
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0};
  float v0 = lane, v1 = lane*2.f, v2 = 3.f, v3 = 1.f;
  const char* lp = lds + lane*16;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    #pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (PAT == 0) {            // 3 dependent MFMAs back to back
        acc0 = MFMA(a,b,acc0); SB; acc0 = MFMA(a,b,acc0); SB; acc0 = MFMA(a,b,acc0); SB;
      } else if (PAT == 1) {     // same acc, 3 VALU between MFMAs
        acc0 = MFMA(a,b,acc0); SB; v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
        acc0 = MFMA(a,b,acc0); SB; v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
        acc0 = MFMA(a,b,acc0); SB; v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
      } else if (PAT == 2) {     // 3 independent accs, 3 VALU between
        acc0 = MFMA(a,b,acc0); SB; v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
        acc1 = MFMA(a,b,acc1); SB; v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
        acc2 = MFMA(a,b,acc2); SB; v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
      } else if (PAT == 3) {     // same acc back-to-back x3 then 9 VALU
        acc0 = MFMA(a,b,acc0); SB; acc0 = MFMA(a,b,acc0); SB; acc0 = MFMA(a,b,acc0); SB;
        v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0);
        v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
      } else if (PAT == 4) {     // PAT0 + 2 ds_read_b128 per 3 MFMAs consumed next iteration
        bf16x8 na = *(const bf16x8*)(lp + ((j*2048) & 32767)); bf16x8 nb = *(const bf16x8*)(lp + ((j*2048+1024)&32767)); SB;
        acc0 = MFMA(a,b,acc0); SB; acc0 = MFMA(a,b,acc0); SB; acc0 = MFMA(a,b,acc0); SB;
        a = na; b = nb;
      } else if (PAT == 5) {     // two accs alternating (dep distance 2), 3 VALU between
        acc0 = MFMA(a,b,acc0); SB; v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
        acc1 = MFMA(a,b,acc1); SB; v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
        acc0 = MFMA(a,b,acc0); SB; v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
      } else if (PAT == 6) {     // VALU only: 9 per step
        v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0);
        v0 = fmaf(v0,v1,v2); v1 = fmaf(v1,v2,v3); v2 = fmaf(v2,v3,v0); SB;
      } else if (PAT == 7) {     // same acc, 1 VALU between MFMAs
        acc0 = MFMA(a,b,acc0); SB; v0 = fmaf(v0,v1,v2); SB;
        acc0 = MFMA(a,b,acc0); SB; v1 = fmaf(v1,v2,v3); SB;
        acc0 = MFMA(a,b,acc0); SB; v2 = fmaf(v2,v3,v0); SB;
      } else if (PAT == 8) {     // same acc, 6 independent VALU between MFMAs
        float w0=v0,w1=v1,w2=v2,w3=v3,w4=v0+1,w5=v1+1;
        acc0 = MFMA(a,b,acc0); SB; w0*=1.1f; w1*=1.1f; w2*=1.1f; w3*=1.1f; w4*=1.1f; w5*=1.1f; SB;
        acc0 = MFMA(a,b,acc0); SB; w0*=1.2f; w1*=1.2f; w2*=1.2f; w3*=1.2f; w4*=1.2f; w5*=1.2f; SB;
        acc0 = MFMA(a,b,acc0); SB; w0*=1.3f; w1*=1.3f; w2*=1.3f; w3*=1.3f; w4*=1.3f; w5*=1.3f; SB;
        v0=w0+w4; v1=w1+w5; v2=w2; v3=w3;
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = v0+v1+v2+v3; for (int r=0;r<16;r++) s += acc0[r]+acc1[r]+acc2[r];
  out[blockIdx.x*256+threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template<int PAT> void run(const char* name, float* d_out, long long* d_cyc) {
  const int iters = 200;
  hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(256), 0, 0, d_out, d_cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(256), 0, 0, d_out, d_cyc, iters); hipEventRecord(e1);
  hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> c(256); hipMemcpy(c.data(), d_cyc, 256*8, hipMemcpyDeviceToHost);
  double avg=0; for (auto x: c) avg += x; avg/=256;
  double steps = iters*16.0;
  printf("%-48s %8.1f ticks/step  wall %.3f ms  -> %.1f ns/step\n", name, avg/steps, ms, ms*1e6/steps);
}
int main(){
  float* d_out; long long* d_cyc; hipMalloc(&d_out, 256*256*4); hipMalloc(&d_cyc, 256*8);
  run<0>("P0 3 dep MFMA b2b", d_out, d_cyc);
  run<1>("P1 same acc, 3 dep-VALU between", d_out, d_cyc);
  run<2>("P2 3 accs, 3 VALU between", d_out, d_cyc);
  run<3>("P3 3 MFMA b2b then 9 VALU", d_out, d_cyc);
  run<4>("P4 P0 + 2 ds_read_b128 prefetch", d_out, d_cyc);
  run<5>("P5 2 accs alternating, 3 VALU between", d_out, d_cyc);
  run<6>("P6 9 dep VALU only", d_out, d_cyc);
  run<7>("P7 same acc, 1 VALU between", d_out, d_cyc);
  run<8>("P8 same acc, 6 indep VALU between", d_out, d_cyc);
  return 0;
}
