// Microbenchmark (diagnostic, not product): how long N equal work items of 256 KB (one 256-thread work-group each, 16-byte
// loads, 8 in flight per lane) take to stream from HBM, for N around the B=1 expert launch's 410 items on 256 CUs
// (2 work-groups resident per CU) -- is the makespan set by the CUs that host two items?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void stream_items(const f32x4* __restrict__ w, float* out, int item_f4, int lds_pad) {
  extern __shared__ float pad[];
  const f32x4* p = w + (size_t)blockIdx.x * item_f4 + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < item_f4; i += 256 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + i + 256 * j);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  if (lds_pad) pad[threadIdx.x] = acc[0];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}
int main() {
  const int item_bytes = 256 * 1024, item_f4 = item_bytes / 16, max_items = 2048;
  f32x4* w; float* out;
  hipMalloc(&w, (size_t)max_items * item_bytes * 4); hipMalloc(&out, 4096 * 4);     // 2 GB: rotate through 4 regions, no cache reuse
  hipMemset(w, 0, (size_t)max_items * item_bytes * 4);
  hipFuncSetAttribute((const void*)stream_items, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ns[] = {128, 256, 320, 384, 410, 448, 512, 640, 768, 1024};
  for (int lds : {0, 40 * 1024}) {      // 40 KB of LDS per work-group keeps the residency at the expert kernel's 2 per CU ... 0: register-limited only
    for (int n : ns) {
      float best = 1e9f;
      for (int rep = 0; rep < 8; ++rep) {
        const f32x4* base = w + (size_t)(rep & 3) * max_items / 4 * item_f4 * 0 + (size_t)(rep & 3) * (size_t)512 * item_f4;
        hipEventRecord(e0);
        stream_items<<<n, 256, lds>>>(base, out, item_f4, lds ? 1 : 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
      }
      printf("lds %5d B  items %4d : %7.2f us   %6.2f TB/s\n", lds, n, best * 1e3, (double)n * item_bytes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
