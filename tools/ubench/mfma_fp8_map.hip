// Development check (not product): operand maps of v_mfma_f32_32x32x16_fp8_fp8 and the fp8 conversion builtins on gfx950,
// with exact small-integer data (cdna_hip_programming.md 3: "check the map with exact integer data").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ unsigned pack4(float a, float b, float c, float d) {
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);   // low 16 bits
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);    // high 16 bits
  return (unsigned)v;
}

// A[32][16], B[16][32] row-major floats (small integers) -> C[32][32] via ONE MFMA, assuming lane (r = l & 31, h = l >> 5)
// holds A[r][8h + j] and B[8h + j][r] in byte j of its 64-bit operand
__global__ void k(const float* A, const float* B, float* C, float* cvt) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  unsigned a0 = pack4(A[r * 16 + 8 * h + 0], A[r * 16 + 8 * h + 1], A[r * 16 + 8 * h + 2], A[r * 16 + 8 * h + 3]);
  unsigned a1 = pack4(A[r * 16 + 8 * h + 4], A[r * 16 + 8 * h + 5], A[r * 16 + 8 * h + 6], A[r * 16 + 8 * h + 7]);
  unsigned b0 = pack4(B[(8 * h + 0) * 32 + r], B[(8 * h + 1) * 32 + r], B[(8 * h + 2) * 32 + r], B[(8 * h + 3) * 32 + r]);
  unsigned b1 = pack4(B[(8 * h + 4) * 32 + r], B[(8 * h + 5) * 32 + r], B[(8 * h + 6) * 32 + r], B[(8 * h + 7) * 32 + r]);
  const long a = (long)(((unsigned long long)a1 << 32) | a0), b = (long)(((unsigned long long)b1 << 32) | b0);
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
  if (l == 0) {   // conversion behaviour: round trip, saturation / overflow
    const float t[8] = {1.f, -2.5f, 448.f, 449.f, 1000.f, -1e6f, 0.001f, 0.0625f};
    for (int i = 0; i < 8; i += 2) {
      int v = __builtin_amdgcn_cvt_pk_fp8_f32(t[i], t[i + 1], 0, false);
      f32x2 back = __builtin_amdgcn_cvt_pk_f32_fp8(v, false);
      cvt[i] = back[0]; cvt[i + 1] = back[1];
    }
  }
}

int main() {
  float hA[32 * 16], hB[16 * 32], hC[32 * 32], ref[32 * 32], hcvt[8];
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (float)((i * 3 + k * 5) % 7 - 3);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((k * 2 + j * 7) % 5 - 2);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dC, *dcvt;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dcvt, sizeof hcvt);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dcvt);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost); hipMemcpy(hcvt, dcvt, sizeof hcvt, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) if (hC[i] != ref[i]) ++bad;
  printf("fp8 32x32x16 operand map: %d mismatches of 1024\n", bad);
  printf("cvt round trips (1, -2.5, 448, 449, 1000, -1e6, 0.001, 0.0625): ");
  for (int i = 0; i < 8; ++i) printf("%g ", hcvt[i]);
  printf("\n");
  return bad != 0;
}
