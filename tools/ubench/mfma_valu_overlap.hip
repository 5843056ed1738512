// Microbenchmark (diagnostic, not product): how much independent VALU / LDS work a lone wave (one per SIMD) can issue in the
// shadow of its own MFMAs.  Loop body = 2 x v_mfma_f32_32x32x16_fp8_fp8 (64 cycles of the matrix pipe) + N other instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int KIND, int NM>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  f32x16 c0, c1;
  for (int j = 0; j < 16; ++j) c0[j] = c1[j] = 0.f;
  long a8 = threadIdx.x * 0x0101010101010101LL, b8 = 0x3838383838383838LL;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j;
  lds[threadIdx.x] = 1.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (NM > 0) asm volatile("v_mfma_f32_32x32x16_fp8_fp8 %0, %1, %2, %0" : "+v"(c0) : "v"(a8), "v"(b8));
#pragma unroll
    for (int u = 0; u < NV / 2; ++u) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[u & 7]));
      if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u & 7]));
      if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v[u & 7]) : "v"((int)(threadIdx.x * 4)));
      if (KIND == 3) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
    }
    if (NM > 1) asm volatile("v_mfma_f32_32x32x16_fp8_fp8 %0, %1, %2, %0" : "+v"(c1) : "v"(a8), "v"(b8));
#pragma unroll
    for (int u = NV / 2; u < NV; ++u) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[u & 7]));
      if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u & 7]));
      if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v[u & 7]) : "v"((int)(threadIdx.x * 4)));
      if (KIND == 3) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NV, int KIND, int NM>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  k<NV, KIND, NM><<<256, 256>>>(out, cyc, 10);
  k<NV, KIND, NM><<<256, 256>>>(out, cyc, 2000);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-10s N=%2d  MFMAs=%d : %.1f ticks per iteration\n", name, NV, NM, h / 2000.0);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 0, 2>("v_fma"); run<8, 0, 2>("v_fma"); run<12, 0, 2>("v_fma"); run<16, 0, 2>("v_fma"); run<24, 0, 2>("v_fma"); run<32, 0, 2>("v_fma");
  run<16, 0, 0>("v_fma"); run<32, 0, 0>("v_fma");
  run<2, 1, 2>("v_exp"); run<4, 1, 2>("v_exp"); run<8, 1, 2>("v_exp"); run<8, 1, 0>("v_exp");
  run<4, 2, 2>("ds_read"); run<8, 2, 2>("ds_read"); run<8, 2, 0>("ds_read");
  run<16, 3, 2>("s_add"); run<32, 3, 2>("s_add"); run<32, 3, 0>("s_add");
  return 0;
}
