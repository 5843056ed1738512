// Microbenchmark (diagnostic, not product): operand lane map of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, found by
// exact small-integer products: hypothesis p puts the lane half h = lane >> 5 at bit p of k (k = bits of byte index j around it).
// Also times the instruction (one wave per SIMD, dependent chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ void k1(const i32x8* a, const i32x8* b, float* d) {
  f32x16 c;
  for (int j = 0; j < 16; ++j) c[j] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 16; ++i) d[((i & 3) + 8 * (i >> 2) + 4 * (threadIdx.x >> 5)) * 32 + (threadIdx.x & 31)] = c[i];
}
__global__ __launch_bounds__(256) void krate(float* out, unsigned long long* cyc, int iters) {
  f32x16 c;
  for (int j = 0; j < 16; ++j) c[j] = 0.f;
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = 0x38383838; b[j] = 0x30303030 + (int)threadIdx.x; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += c[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
static uint8_t e4m3_of_small_int(int v) {          // 0..8 exactly: 0, 1 = 0x38, 2 = 0x40, 3 = 0x44, 4 = 0x48, 5 = 0x4a, 6 = 0x4c, 7 = 0x4e, 8 = 0x50
  static const uint8_t t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
  return t[v];
}
int main() {
  int A[32][64], B[64][32];
  srand(1);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i][k] = rand() % 9;
  for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = rand() % 9;
  float want[32][32];
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 64; ++k) s += A[i][k] * B[k][j]; want[i][j] = (float)s; }
  i32x8 *da, *db; float* dd;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dd, 32 * 32 * 4);
  for (int p = 0; p <= 5; ++p) {
    uint8_t ha[64][32], hb[64][32];
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j) {
        const int h = l >> 5, lo = j & ((1 << p) - 1), hi = j >> p;
        const int k = (hi << (p + 1)) | (h << p) | lo;
        ha[l][j] = e4m3_of_small_int(A[l & 31][k]);
        hb[l][j] = e4m3_of_small_int(B[k][l & 31]);
      }
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    k1<<<1, 64>>>(da, db, dd);
    float got[32][32]; hipMemcpy(got, dd, sizeof(got), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) bad += got[i][j] != want[i][j];
    printf("hypothesis h at bit %d of k: %d of 1024 outputs differ (got[0][0] %.0f want %.0f)\n", p, bad, got[0][0], want[0][0]);
  }
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  krate<<<256, 256>>>(out, cyc, 10);
  hipEventRecord(e0); krate<<<256, 256>>>(out, cyc, 2000); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h0; hipMemcpy(&h0, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 8.0 * 2000;
  printf("32x32x64 f8f6f4 (e4m3): s_memtime ticks / MFMA %.2f  ns / MFMA %.3f  chip %.1f TFLOP/s\n", h0 / n, ms * 1e6 / n,
         256.0 * 4 * n * 131072 / (ms * 1e-3) / 1e12);
  return 0;
}
