// Microbenchmark (development tool, not product): how fast can a work-group of 4 waves stream weights into an LDS ring by
// LDS-DMA, as the fused expert kernel does -- isolates the fill path from MFMA / fragment reads.
//   hipcc -O3 --offload-arch=gfx950 -o ldsdma_fill ldsdma_fill.hip && ./ldsdma_fill
// Each work-group streams `bytes_per_wg` (2 MB) of a buffer through a ring of R slots x P bytes: wait own fills (counted
// vmcnt), s_barrier, [read one 16-B value per lane from the slot so the data is consumed], refill the slot freed a step ago.
// Variants: share = work-groups reading the SAME region (1 = private regions, 16 = as the fused kernel: 16 token tiles of
// one expert), swz = source-side XOR swizzle on/off, form = buffer_load..lds / global_load_lds, lockstep offset on/off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PIECE bytes per slot, RING slots, INSTR = PIECE / 1024 / 4 fills per wave per piece
template <int PIECE, int RING, int FORM, bool SWZ>
__global__ __launch_bounds__(256) void fill_kernel(const char* __restrict__ w, size_t region_bytes, int share, int dephase,
                                                   int bytes_per_wg, float* __restrict__ sink, int gap,
                                                   unsigned long long* __restrict__ issue_cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IPW = PIECE / 1024 / 4;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per = gridDim.x >> 3;
  const int logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int region = logical / share, member = logical % share;
  const char* base = w + (size_t)region * region_bytes;
  const int npieces = bytes_per_wg / PIECE;
  const int phase = dephase ? (member * 5) % npieces : 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region_bytes, 0x00020000);
  unsigned voff[IPW];
#pragma unroll
  for (int ii = 0; ii < IPW; ++ii) {
    const int i = IPW * wv + ii;                       // 1-KB instruction index inside the piece
    const int c = SWZ ? ((lane & 48) | ((lane ^ i) & 15)) : lane;
    voff[ii] = (unsigned)(i * 1024 + c * 16);
  }
  auto issue = [&](int t) {
    const int pc = (t + phase) % npieces;
    const unsigned soff = (unsigned)pc * PIECE;
    char* dst = smem + (t % RING) * PIECE + wv * (PIECE / 4);
#pragma unroll
    for (int ii = 0; ii < IPW; ++ii) {
      if (FORM == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + ii * 1024), 16, (int)voff[ii], (int)soff, 0, 0);
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + soff + voff[ii]),
                                         (__attribute__((address_space(3))) void*)(dst + ii * 1024), 16, 0, 0);
      }
    }
  };
  float acc = 0.f;
  unsigned long long t_issue = 0;
#pragma unroll
  for (int t = 0; t < RING - 1; ++t) issue(t);
  for (int t = 0; t < npieces; ++t) {
    if (npieces - 1 - t >= RING - 2) wait_vmcnt<(RING - 2) * IPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (t + RING - 1 < npieces) issue(t + RING - 1);
    asm volatile("" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    t_issue += c1 - c0;
    acc += *reinterpret_cast<const float*>(smem + (t % RING) * PIECE + threadIdx.x * 16);
    if (gap > 0) {                                       // a compute phase between bursts, as in the fused kernel
      while (__builtin_amdgcn_s_memtime() - c1 < (unsigned long long)gap) __builtin_amdgcn_s_sleep(2);
    }
  }
  if (acc == 123.456f) sink[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 17) issue_cycles[0] = t_issue;
}

template <int PIECE, int RING, int FORM, bool SWZ>
static void run(const char* name, const char* w, size_t region_bytes, int nregions, int share, int dephase, float* sink, int gap = 0) {
  static unsigned long long* d_ic = nullptr;
  if (!d_ic) CHECK(hipMalloc(&d_ic, 8));
  const int nwg = nregions * share;
  const int bytes_per_wg = 2 << 20;
  CHECK(hipFuncSetAttribute((const void*)fill_kernel<PIECE, RING, FORM, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, PIECE * RING));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it)
    hipLaunchKernelGGL((fill_kernel<PIECE, RING, FORM, SWZ>), dim3(nwg), dim3(256), PIECE * RING, 0, w, region_bytes, share, dephase, bytes_per_wg, sink, gap, d_ic);
  CHECK(hipEventRecord(e0));
  const int iters = 10;
  for (int it = 0; it < iters; ++it)
    hipLaunchKernelGGL((fill_kernel<PIECE, RING, FORM, SWZ>), dim3(nwg), dim3(256), PIECE * RING, 0, w, region_bytes, share, dephase, bytes_per_wg, sink, gap, d_ic);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double rounds = (nwg + 255) / 256;
  unsigned long long ic = 0;
  CHECK(hipMemcpy(&ic, d_ic, 8, hipMemcpyDeviceToHost));
  const int n_issue = bytes_per_wg / 1024 / 4;
  printf("%-30s wgs %4d share %2d dephase %d gap %5d : %8.1f us  %6.1f GB/s per CU (%.2f TB/s chip), %.0f cycles per fill issue\n", name, nwg,
         share, dephase, gap, us, (double)bytes_per_wg * rounds / us / 1e3, (double)bytes_per_wg * nwg / us / 1e6, (double)ic / n_issue);
}

int main() {
  const size_t region = 2 << 20;
  const int nregions = 256;                        // 512 MB: private regions for 256 work-groups
  char* w; float* sink;
  CHECK(hipMalloc(&w, region * nregions));
  CHECK(hipMemset(w, 1, region * nregions));
  CHECK(hipMalloc(&sink, 4));
  // private regions (HBM / Infinity-Cache streaming), one work-group per CU
  run<16384, 8, 0, true>("16K x8 buffer swz", w, region, 256, 1, 0, sink);
  run<32768, 4, 0, true>("32K x4 buffer swz", w, region, 256, 1, 0, sink);
  run<16384, 8, 0, false>("16K x8 buffer linear", w, region, 256, 1, 0, sink);
  run<16384, 8, 1, true>("16K x8 global_load_lds swz", w, region, 256, 1, 0, sink);
  // 16 work-groups share a region (the fused kernel's situation): 16 regions x 16 = 256 work-groups
  run<16384, 8, 0, true>("16K x8 buffer swz", w, region, 16, 16, 0, sink);
  run<16384, 8, 0, true>("16K x8 buffer swz", w, region, 16, 16, 1, sink);
  run<32768, 4, 0, true>("32K x4 buffer swz", w, region, 16, 16, 0, sink);
  run<32768, 4, 0, true>("32K x4 buffer swz", w, region, 16, 16, 1, sink);
  run<16384, 8, 0, false>("16K x8 buffer linear", w, region, 16, 16, 1, sink);
  run<16384, 8, 1, true>("16K x8 global_load_lds swz", w, region, 16, 16, 1, sink);
  // 32 regions x 16 = 512 work-groups (two rounds), as S = 65536
  run<16384, 8, 0, true>("16K x8 buffer swz", w, region, 32, 16, 1, sink);
  // everything from L2: all 256 work-groups share ONE 2 MB region / 8 regions (one per XCD)
  run<16384, 8, 0, true>("16K x8 buffer swz", w, region, 1, 256, 1, sink);
  run<16384, 8, 0, true>("16K x8 buffer swz", w, region, 8, 32, 1, sink);
  // bursts separated by a compute-like gap (cycles), shared regions as in the fused kernel
  run<32768, 4, 0, true>("32K x4 buffer swz", w, region, 16, 16, 1, sink, 1000);
  run<32768, 4, 0, true>("32K x4 buffer swz", w, region, 16, 16, 1, sink, 2200);
  run<32768, 4, 0, true>("32K x4 buffer swz", w, region, 16, 16, 1, sink, 4400);
  run<16384, 8, 0, true>("16K x8 buffer swz", w, region, 16, 16, 1, sink, 1100);
  run<32768, 4, 0, true>("32K x4 buffer swz (private)", w, region, 256, 1, 0, sink, 2200);
  return 0;
}
