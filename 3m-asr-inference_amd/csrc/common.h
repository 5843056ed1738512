// Shared host/device helpers for libm3asr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

namespace m3 {

// ---- error convention: int status + thread-local last-error string --------------------
// (reference: status int from enqueue, CUDA errors only logged, common/common.h:26-38)
void set_error(const char* fmt, ...);
const char* last_error();

#define M3_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      m3::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return -1;                                                                        \
    }                                                                                   \
  } while (0)

#define M3_REQUIRE(cond, ...)                  \
  do {                                         \
    if (!(cond)) {                             \
      m3::set_error(__VA_ARGS__);              \
      return -2;                               \
    }                                          \
  } while (0)

#define M3_LAUNCH_CHECK() M3_CHECK_HIP(hipGetLastError())

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// 1-D grid of 256-thread blocks, capped (kernels grid-stride the rest)
static inline unsigned grid1d(size_t n, size_t cap) {
  size_t g = (n + 255) / 256;
  return (unsigned)(g < cap ? g : cap);
}

// ---- device helpers ---------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Cross-lane reductions.  Inside a 16-lane row they are pure DPP moves (no LDS round trip, unlike the
// ds_bpermute that __shfl_xor lowers to): xor-1 / xor-2 quad permutes, then row_half_mirror and row_mirror.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = group16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = group16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- host-side per-device state.  hipFuncSetAttribute and the CU count belong to a device, not to the process: a process that
//      drives several devices (tests, a server with one engine per GPU) must set kernel attributes once on EACH of them. ----
struct PerDeviceOnce {                              // "has this initialisation run on the current device yet?"
  unsigned long long mask = 0;
  static int current() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 ? dev : 0;
  }
  bool done() const { return (mask >> current()) & 1ull; }
  void mark() { mask |= 1ull << current(); }
};
inline int device_cu_count() {                      // CUs of the current device (256 when there is none: host-side size queries)
  static int cached[64] = {0};
  const int dev = PerDeviceOnce::current();
  if (!cached[dev]) {
    int n = 0;
    cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cached[dev];
}

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stg4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// Weights that a launch reads exactly once (the skinny GEMMs and the B=1 expert kernel stream 1-107 MB of them per launch).
// -DM3_NT_WEIGHTS=1 makes these loads non-temporal (L2-served, bypass L1: MI355X_MICROARCH.md).  A/B at configs[1], same device,
// same process order (profiles/r03_ab_headline.txt): one forward alone 2.377 vs 2.390 ms (-0.5 %), but 187 k vs 207 k frames/s
// with four execution contexts (-10 %): the four row-tile work-groups that share a weight tile through their XCD's L2 no
// longer find it there once their timing is skewed by the other contexts.  Off by default.
#ifndef M3_NT_WEIGHTS
#define M3_NT_WEIGHTS 0
#endif
__device__ __forceinline__ f32x4 ldg4_w(const float* p) {
#if M3_NT_WEIGHTS
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
  return *reinterpret_cast<const f32x4*>(p);
#endif
}

// D = A(16x4) * B(4x16) + C, exact f32 (v_mfma_f32_16x16x4_f32).
// lane l: a = A[l&15][l>>4], b = B[l>>4][l&15]; c[r] = C[(l>>4)*4 + r][l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- bf16 operands (weights stored bf16, activations rounded at the MFMA input, fp32 accumulate) ----
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 ldg8h(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 ldg8h_w(const bf16_t* p) {   // once-read weights, see ldg4_w
#if M3_NT_WEIGHTS
  return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
#else
  return *reinterpret_cast<const bf16x8*>(p);
#endif
}
// 8 consecutive floats -> 8 bf16 (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16x8 cvt8(f32x4 lo, f32x4 hi) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[j] = (bf16_t)lo[j];
    r[4 + j] = (bf16_t)hi[j];
  }
  return r;
}
// D = A(16x32) * B(32x16) + C (v_mfma_f32_16x16x32_bf16).
// lane l: a[j] = A[l&15][8*(l>>4)+j], b[j] = B[8*(l>>4)+j][l&15]; c[r] = C[(l>>4)*4 + r][l&15].
__device__ __forceinline__ f32x4 mfma16h(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- fp8 (OCP e4m3) weights: dequantised to bf16 on their way to the MFMA (every e4m3 value is exact in bf16) ----
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 ldg16b(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x4 ldg16b_w(const void* p) {     // once-read weights, see ldg4_w
#if M3_NT_WEIGHTS
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#else
  return *reinterpret_cast<const u32x4*>(p);
#endif
}
// 8 consecutive fp8 (two dwords) -> bf16x8, element order preserved
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(unsigned int lo, unsigned int hi) {
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
  const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
  bf16x8 r;
  r[0] = (bf16_t)a[0]; r[1] = (bf16_t)a[1]; r[2] = (bf16_t)b[0]; r[3] = (bf16_t)b[1];
  r[4] = (bf16_t)c[0]; r[5] = (bf16_t)c[1]; r[6] = (bf16_t)d[0]; r[7] = (bf16_t)d[1];
  return r;
}

}  // namespace m3
