// The router gate on one lane (shared by moe_index.hip and the self-routing expert kernel of moe_expert.hip).
#pragma once
#include "common.h"

namespace m3 {

// Router gate on one lane: softmax + top-1 over WIDTH logits with the reference's arg-max tree
// (SoftmaxAndTop1KernelSmall, softmax_topk_kernel.cu:55-64: stride tree, strict '<').
// want_value = false: the arg-max only (val_out untouched): one pass over the row, no exponentials.
template <int WIDTH>
__device__ __forceinline__ void gate_top1_lane(const float* __restrict__ row, int* idx_out, float* val_out, bool want_value = true) {
  constexpr int H = WIDTH / 2;
  float v[H];
  int id[H];
  // first tree stage (stride WIDTH/2) while loading, so only WIDTH/2 candidates stay in registers
#pragma unroll
  for (int j = 0; j < H; j += 4) {
    const f32x4 lo = ldg4(row + j), hi = ldg4(row + H + j);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool take_hi = lo[q] < hi[q];
      v[j + q] = take_hi ? hi[q] : lo[q];
      id[j + q] = take_hi ? H + j + q : j + q;
    }
  }
#pragma unroll
  for (int stride = H >> 1; stride > 0; stride >>= 1)
#pragma unroll
    for (int t = 0; t < stride; ++t)
      if (v[t] < v[t + stride]) {
        v[t] = v[t + stride];
        id[t] = id[t + stride];
      }
  *idx_out = id[0];
  if (!want_value) return;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < WIDTH; j += 4) {   // second pass over the row (L1-resident)
    const f32x4 t = ldg4(row + j);
#pragma unroll
    for (int q = 0; q < 4; ++q) sum += expf(t[q] - v[0]);
  }
  *val_out = 1.f / sum;
}

// The same gate on a row that is already in registers (row[j] = logits 4j .. 4j+3): lets the caller put other loads between
// the row's loads and its first use.
template <int WIDTH>
__device__ __forceinline__ void gate_top1_regs(const f32x4* row, int* idx_out, float* val_out, bool want_value) {
  constexpr int H = WIDTH / 2;
  float v[H];
  int id[H];
#pragma unroll
  for (int j = 0; j < H; j += 4) {
    const f32x4 lo = row[j / 4], hi = row[(H + j) / 4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool take_hi = lo[q] < hi[q];
      v[j + q] = take_hi ? hi[q] : lo[q];
      id[j + q] = take_hi ? H + j + q : j + q;
    }
  }
#pragma unroll
  for (int stride = H >> 1; stride > 0; stride >>= 1)
#pragma unroll
    for (int t = 0; t < stride; ++t)
      if (v[t] < v[t + stride]) {
        v[t] = v[t + stride];
        id[t] = id[t + stride];
      }
  *idx_out = id[0];
  if (!want_value) return;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < WIDTH; j += 4)
#pragma unroll
    for (int q = 0; q < 4; ++q) sum += expf(row[j / 4][q] - v[0]);
  *val_out = 1.f / sum;
}

}  // namespace m3
