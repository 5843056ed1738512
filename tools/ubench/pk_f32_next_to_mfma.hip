// Microbenchmark (development tool, not product): do packed-FP32 VALU instructions (v_pk_add_f32) give the same sums when the
// other wave of their SIMD is issuing MFMAs?  (moe_router_kernel's LayerNorm sums went wrong about once in 100
// forwards next to work-groups of another launch that were in their MFMA phase.)
//   hipcc -O3 --offload-arch=gfx950 -o pk_f32_next_to_mfma pk_f32_next_to_mfma.hip && ./pk_f32_next_to_mfma
// One kernel, two roles by work-group parity, 70 KB of LDS each so that exactly two work-groups share a CU (one wave of each
// per SIMD): "sum" work-groups add the same 8 floats per lane over and over with packed adds and count every result that
// differs from their first; "mfma" work-groups run v_mfma_f32_16x16x4_f32 back to back.  Modes: sums next to sums, sums next to
// MFMAs, and the same with scalar (unpacked) adds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool PACKED>
__global__ __launch_bounds__(256, 2) void roles_kernel(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, int mode) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const bool mfma_role = mode >= 1 && (blockIdx.x & 1);
  if (mfma_role) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (mode == 1) {                                       // fp32 MFMA, registers only
      float a = in[threadIdx.x], b = in[threadIdx.x + 256];
      for (int i = 0; i < iters * 4; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
    } else {                                               // bf16 MFMA; mode 3: operands re-read from LDS, which is re-written from memory
      bf16x8 a, b;
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[threadIdx.x + e]; b[e] = (__bf16)in[threadIdx.x + 8 + e]; }
      bf16x8* l = reinterpret_cast<bf16x8*>(lds);
      l[threadIdx.x] = a; l[256 + threadIdx.x] = b;
      __syncthreads();
      for (int i = 0; i < iters; ++i) {
        if (mode == 3) {
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(in + ((size_t)((i * 256 + threadIdx.x) % 4096)) * 4);
          bf16x8 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) { w[e] = (__bf16)g0[e]; w[4 + e] = (__bf16)g0[e]; }
          l[512 + ((i & 7) * 256) + threadIdx.x] = w;
          a = l[(threadIdx.x + i) & 255];
          b = l[256 + ((threadIdx.x + 2 * i) & 255)];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
      }
    }
    if (acc[0] == 1234567.f) bad[1] = 1;                  // (keeps the chain alive)
    return;
  }
  const float* src = in + ((size_t)(blockIdx.x % 64) * 256 + threadIdx.x) * 8;
  f32x2 v0 = *reinterpret_cast<const f32x2*>(src), v1 = *reinterpret_cast<const f32x2*>(src + 2);
  f32x2 v2 = *reinterpret_cast<const f32x2*>(src + 4), v3 = *reinterpret_cast<const f32x2*>(src + 6);
  float ref = 0.f;
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));      // recomputed every iteration
    float s;
    if (PACKED) {
      f32x2 t = (v0 + v1) + (v2 + v3);                     // v_pk_add_f32 x3
      asm volatile("" : "+v"(t));
      f32x2 u = t * f32x2{1.0f, 1.0f} + f32x2{0.f, 0.f};   // v_pk_mul_f32 / v_pk_add_f32 (exact)
      asm volatile("" : "+v"(u));
      s = u[0] + u[1];
    } else {
      float t0 = (v0[0] + v1[0]) + (v2[0] + v3[0]), t1 = (v0[1] + v1[1]) + (v2[1] + v3[1]);
      asm volatile("" : "+v"(t0), "+v"(t1));
      s = t0 + t1;
    }
    if (it == 0) ref = s;
    else if (s != ref) ++wrong;
  }
  if (wrong) atomicAdd(&bad[0], wrong);
  if (lane == 0 && lds[0] == 3.f) bad[2] = 1;
}

int main() {
  const size_t n = (size_t)64 * 256 * 8 + 1024;
  std::vector<float> h(n);
  srand(1);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 4.f - 2.f;
  float* in;
  unsigned* bad;
  CHECK(hipMalloc(&in, n * 4));
  CHECK(hipMalloc(&bad, 16));
  CHECK(hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice));
  const int lds = 70 * 1024;
  CHECK(hipFuncSetAttribute((const void*)roles_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHECK(hipFuncSetAttribute((const void*)roles_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const char* names[4] = {"sums next to sums", "sums next to fp32 MFMAs", "sums next to bf16 MFMAs", "... + LDS / loads"};
  for (int packed = 1; packed >= 0; --packed)
    for (int mode = 0; mode < 4; ++mode) {
      CHECK(hipMemset(bad, 0, 16));
      for (int rep = 0; rep < 20; ++rep) {
        if (packed) hipLaunchKernelGGL(roles_kernel<true>, dim3(2048), dim3(256), lds, 0, in, bad, 20000, mode);
        else hipLaunchKernelGGL(roles_kernel<false>, dim3(2048), dim3(256), lds, 0, in, bad, 20000, mode);
      }
      CHECK(hipDeviceSynchronize());
      unsigned hb[4];
      CHECK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
      printf("%-8s adds, %-24s: %u sums differ from the first one of their lane (%.1f G sums)\n", packed ? "packed" : "scalar", names[mode], hb[0],
             20.0 * 2048 * (mode ? 0.5 : 1.0) * 256 * 20000 / 1e9);
    }
  return 0;
}
