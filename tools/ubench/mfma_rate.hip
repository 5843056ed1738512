// Microbenchmark (diagnostic, not product): issue rate of v_mfma_f32_32x32x16_{bf16,fp8_fp8} with one wave per SIMD,
// as a dependent chain on ONE accumulator and round-robin over 2 / 4 accumulators.  Prints shader cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  long a8 = threadIdx.x * 0x0101010101010101LL, b8 = 0x3838383838383838LL;
  bf16x8 ah, bh;
  for (int j = 0; j < 8; ++j) { ah[j] = (__bf16)(float)(threadIdx.x & 7); bh[j] = (__bf16)1.0f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, b8, acc[u % NACC], 0, 0, 0);
      else acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[u % NACC], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND, int NACC>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND, NACC><<<256, 256>>>(out, cyc, 10);
  hipEventRecord(e0);
  k<KIND, NACC><<<256, 256>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = 16.0 * iters;
  printf("%-34s s_memtime ticks / MFMA %.2f   ns / MFMA %.3f   chip %.1f TFLOP/s\n", name, h[0] / n, ms * 1e6 / n,
         256.0 * 4 * n * 32768 / (ms * 1e-3) / 1e12);
}
int main() {
  run<0, 1>("fp8 32x32x16, 1 accumulator");
  run<0, 2>("fp8 32x32x16, 2 accumulators");
  run<0, 4>("fp8 32x32x16, 4 accumulators");
  run<1, 1>("bf16 32x32x16, 1 accumulator");
  run<1, 2>("bf16 32x32x16, 2 accumulators");
  run<1, 4>("bf16 32x32x16, 4 accumulators");
  return 0;
}
