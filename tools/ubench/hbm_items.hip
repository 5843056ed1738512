// Microbenchmark (diagnostic, not product): how long N equal work items of 256 KB (one 256-thread work-group each, 16-byte
// loads, 8 in flight per lane) take to stream from HBM, for N around the B=1 expert launch's 410 items on 256 CUs
// (2 work-groups resident per CU) -- is the makespan set by the CUs that host two items?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// PATTERN 0: fully coalesced (a wave instruction = 1 KB contiguous).  PATTERN 1: the operand pattern of the fp32 expert
// kernel's phase 1 (moe_expert.hip): wave w owns 16 rows of 2 KB, lane (col = lane & 15, kq = lane >> 4) reads 16 B at
// k-step s of row 16 w + col: an instruction touches 16 rows x 64 B, eight k-steps per group.
template <int PATTERN>
__global__ __launch_bounds__(256, 2) void stream_items(const f32x4* __restrict__ w, float* out, int item_f4, int lds_pad) {
  extern __shared__ float pad[];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (PATTERN == 0) {
    const f32x4* p = w + (size_t)blockIdx.x * item_f4 + threadIdx.x;
    for (int i = 0; i < item_f4; i += 256 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + i + 256 * j);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
  } else {
    // the item = 2 x (64 rows x 2 KB): two 128-KB halves walked like W1 rows (512 floats = 128 float4 per row, 32 k-steps)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
    for (int half = 0; half < 2; ++half) {
      const f32x4* row = w + (size_t)blockIdx.x * item_f4 + (size_t)half * (item_f4 / 2) + (size_t)(16 * wave + col) * 128 + kq;
      for (int g = 0; g < 4; ++g) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(row + 4 * (8 * g + j));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
      }
    }
  }
  if (lds_pad) pad[threadIdx.x] = acc[0];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}
int main() {
  const int item_bytes = 256 * 1024, item_f4 = item_bytes / 16, max_items = 2048;
  f32x4* w; float* out;
  hipMalloc(&w, (size_t)max_items * item_bytes * 4); hipMalloc(&out, 4096 * 4);     // 2 GB: rotate through 4 regions, no cache reuse
  hipMemset(w, 0, (size_t)max_items * item_bytes * 4);
  hipFuncSetAttribute((const void*)stream_items<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipFuncSetAttribute((const void*)stream_items<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ns[] = {256, 410, 512, 1024};
  for (int pat = 0; pat < 2; ++pat)
  for (int lds : {0, 40 * 1024}) {      // 40 KB of LDS per work-group keeps the residency at the expert kernel's 2 per CU ... 0: register-limited only
    for (int n : ns) {
      float best = 1e9f;
      for (int rep = 0; rep < 8; ++rep) {
        const f32x4* base = w + (size_t)(rep & 3) * max_items / 4 * item_f4 * 0 + (size_t)(rep & 3) * (size_t)512 * item_f4;
        hipEventRecord(e0);
        if (pat == 0) stream_items<0><<<n, 256, lds>>>(base, out, item_f4, lds ? 1 : 0);
        else stream_items<1><<<n, 256, lds>>>(base, out, item_f4, lds ? 1 : 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
      }
      printf("pattern %d lds %5d B  items %4d : %7.2f us   %6.2f TB/s\n", pat, lds, n, best * 1e3, (double)n * item_bytes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
