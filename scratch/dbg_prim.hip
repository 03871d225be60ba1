#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256,1) void k_glds(const char* src, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63; const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned voff = lane*16 + wave*8192;
  const unsigned dst = lds_ring + 32768 /*slot 1*/ + wave*8192;
  unsigned keep, t;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "v_mov_b32 %1, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "v_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "v_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "v_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "v_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "v_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "v_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "v_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %4\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "=&v"(t) : "v"(voff), "s"(dst), "s"(src) : "memory", "scc");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 8192; i += 256) out[i] = *(unsigned*)(smem + 32768 + i*4);
}

// MFMA layout probe: A[i][k] = i*100 + k (small ints exact in bf16? up to 31*100+15=3115 needs 12 bits -> not exact). use A[i][k]= (i==probe_i && k==probe_k), B[k][j] = (k==probe_k && j==probe_j)
__global__ void k_mfma(float* out /*[64][16]*/, int pi, int pk, int pj, int ah, int ae, int bh, int be) {
  const int lane = threadIdx.x & 63; const int i = lane & 31, h = lane >> 5;
  bf16x8 a, b;
  for (int e=0;e<8;e++){ a[e] = (__bf16)((i==pi && h==ah && e==ae) ? 1.0f : 0.0f); b[e] = (__bf16)((i==pj && h==bh && e==be) ? 1.0f : 0.0f); }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0,0,0);
  for (int r=0;r<16;r++) out[lane*16+r] = acc[r];
}

int main(){
  // glds test
  std::vector<unsigned> h(8192); for (int i=0;i<8192;i++) h[i]=i;
  char* d_src; unsigned* d_out; hipMalloc(&d_src, 32768); hipMalloc(&d_out, 32768);
  hipMemcpy(d_src, h.data(), 32768, hipMemcpyHostToDevice); hipMemset(d_out, 0xff, 32768);
  hipFuncSetAttribute((const void*)k_glds, hipFuncAttributeMaxDynamicSharedMemorySize, 123008);
  hipLaunchKernelGGL(k_glds, dim3(1), dim3(256), 123008, 0, d_src, d_out);
  hipError_t e = hipDeviceSynchronize(); printf("glds sync: %s\n", hipGetErrorString(e));
  std::vector<unsigned> o(8192); hipMemcpy(o.data(), d_out, 32768, hipMemcpyDeviceToHost);
  int bad=0; for (int i=0;i<8192;i++) if (o[i]!=(unsigned)i){ if(bad<10) printf("  out[%d]=%u\n", i, o[i]); bad++; }
  printf("glds mismatches: %d\n", bad);
  // mfma probe: A row pi, lane-half ah elem ae ; B col pj half bh elem be -> nonzero only if (ah,ae)==(bh,be)
  float* d_m; hipMalloc(&d_m, 64*16*4); std::vector<float> m(1024);
  int tests[][7] = {{5,0,9, 0,3, 0,3},{5,0,9, 1,6, 1,6},{17,0,30, 1,2, 1,2},{5,0,9, 0,3, 1,3},{5,0,9,0,3,0,4}};
  for (auto& t : tests){
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, d_m, t[0],t[1],t[2],t[3],t[4],t[5],t[6]);
    hipMemcpy(m.data(), d_m, 4096, hipMemcpyDeviceToHost);
    printf("A(row %d,h%d,e%d) B(col %d,h%d,e%d): nonzeros:", t[0],t[3],t[4],t[2],t[5],t[6]);
    for (int l=0;l<64;l++) for(int r=0;r<16;r++) if (m[l*16+r]!=0) printf(" [lane %d (col %d,h %d) reg %d -> row %d]=%g", l, l&31, l>>5, r, (r&3)+8*(r>>2)+4*(l>>5), m[l*16+r]);
    printf("\n");
  }
  return 0;
}
