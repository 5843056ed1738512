// Experiment: what does COLD CODE cost a short kernel?  A launch-bound forward runs 275 kernels of
// 0.5-1.5 k instructions each, every one of them entering an instruction cache that the previous kernels have filled with
// their own code.  Two kernels execute the same number of scalar no-ops, one as straight-line code (N x 4 bytes of
// instructions, each fetched once), the other as a loop over a 64-instruction body (256 bytes, fetched once, reused);
// `evict` runs a third, large kernel between repetitions so that the measured launch finds the cache cold.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_exp_icache tools/exp_icache.hip && tools/_exp_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int N>
struct Nops {
  static __device__ __forceinline__ void run() {
    asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\n"
                 "s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0");
    Nops<N - 16>::run();
  }
};
template <>
struct Nops<0> {
  static __device__ __forceinline__ void run() {}
};

template <int N>
__global__ __launch_bounds__(256) void straight(int* out) {
  Nops<N>::run();
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = N;
}

template <int N>
__global__ __launch_bounds__(256) void looped(int* out) {
  for (int i = 0; i < N / 64; ++i) {
    Nops<64>::run();
    asm volatile("" ::: "memory");
  }
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = N;
}

// second family, so that "the previous kernel" is different code of the same size
template <int N>
__global__ __launch_bounds__(256) void other(int* out) {
  Nops<N>::run();
  asm volatile("s_nop 1");
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = -N;
}

template <typename F>
static float time_launches(F&& launch, hipStream_t s, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) launch();
  hipStreamSynchronize(s);
  // capture `reps` launches in a graph: the same submission path as the engine's forward
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < reps; ++i) launch();
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  std::vector<float> t;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(a, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    t.push_back(ms * 1000.f / reps);
  }
  std::sort(t.begin(), t.end());
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return t[t.size() / 2];
}

int main() {
  hipStream_t s; CHECK(hipStreamCreate(&s));
  int* out; CHECK(hipMalloc(&out, 64));
  const int reps = 200;
  const dim3 grid(256), block(256);
  printf("grid 256 x 256 threads, %d launches per graph, us per launch (median of 7)\n", reps);
  printf("%8s %12s %12s %22s\n", "instrs", "looped", "straight", "straight, alternating");
#define ROW(N)                                                                                          \
  {                                                                                                     \
    const float tl = time_launches([&] { hipLaunchKernelGGL(looped<N>, grid, block, 0, s, out); }, s, reps);   \
    const float ts = time_launches([&] { hipLaunchKernelGGL(straight<N>, grid, block, 0, s, out); }, s, reps); \
    int k = 0;                                                                                          \
    const float ta = time_launches([&] {                                                                \
      if ((k++ & 1) == 0) hipLaunchKernelGGL(straight<N>, grid, block, 0, s, out);                      \
      else hipLaunchKernelGGL(other<N>, grid, block, 0, s, out);                                        \
    }, s, reps);                                                                                        \
    printf("%8d %12.2f %12.2f %22.2f\n", N, tl, ts, ta);                                                \
  }
  ROW(64) ROW(512) ROW(1024) ROW(2048) ROW(4096) ROW(8192) ROW(16384)
  // the same with ONE work-group of one wave: no sharing of fetched lines between waves
  printf("grid 1 x 64 threads\n");
#define ROW1(N)                                                                                         \
  {                                                                                                     \
    const float tl = time_launches([&] { hipLaunchKernelGGL(looped<N>, dim3(1), dim3(64), 0, s, out); }, s, reps);   \
    const float ts = time_launches([&] { hipLaunchKernelGGL(straight<N>, dim3(1), dim3(64), 0, s, out); }, s, reps); \
    printf("%8d %12.2f %12.2f\n", N, tl, ts);                                                           \
  }
  ROW1(64) ROW1(1024) ROW1(4096) ROW1(16384)
  return 0;
}
