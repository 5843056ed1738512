// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns this library uses (diagnostic, not product).
// MI355X_MICROARCH.md (HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B reads exactly 1/2 of the bytes of a wide coalesced
// streaming read (128-B requests tallied at 64 B); "other access widths are uncalibrated".  The weight loads of the skinny
// GEMMs and of the B=1 expert kernel are NOT full lines per instruction: a wave instruction reads 16 rows x 64 B (one half
// line per row; the other half comes with the next k-step).  Every kernel below reads a KNOWN byte count exactly once:
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// then FETCH_SIZE(KB) x 1024 / bytes_read per kernel name is the factor to apply (0.5 -> double it, 1.0 -> take it as is).
//   full_line<NT>   : lane i reads 16 B at base + 16 i: 1 KB contiguous per wave instruction
//   half_line<NT>   : lane (col = lane & 15, kq = lane >> 4) reads 16 B at row (16 w + col), byte 64 s + 16 kq: 16 rows x 64 B
//                     per instruction, rows 2 KB apart (fp32 W1 rows of D = 512), k-steps s = 0..31 back to back
//   quarter_line<NT>: 16 rows x 64 B with rows 256 B apart (the slice-major W2 of the plan: [D][64] floats)
// NT = 1: __builtin_nontemporal_load, 0: plain load.  Each launch walks its own 512-MB region (nothing is re-read, the
// 256-MB Infinity Cache cannot serve a second touch).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ f32x4 ld(const f32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <int NT>
__global__ __launch_bounds__(256) void full_line(const f32x4* __restrict__ w, float* out, size_t f4_per_block) {
  const f32x4* p = w + (size_t)blockIdx.x * f4_per_block + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = 0; i < f4_per_block; i += 256 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ld<NT>(p + i + 256 * j);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

// a block owns `rows_per_block` rows of ROW_F4 float4 each (4 waves x 16 rows per pass)
template <int NT, int ROW_F4>
__global__ __launch_bounds__(256) void part_line(const f32x4* __restrict__ w, float* out, int rows_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = 0; r0 < rows_per_block; r0 += 64) {
    const f32x4* row = w + ((size_t)blockIdx.x * rows_per_block + r0 + 16 * wave + col) * ROW_F4 + kq;
    for (int s = 0; s < ROW_F4 / 4; s += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ld<NT>(row + 4 * (s + j < ROW_F4 / 4 ? s + j : 0));
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

int main() {
  const size_t region = (size_t)512 << 20;         // bytes per launch
  const int nreg = 6;
  f32x4* w; float* out;
  if (hipMalloc(&w, region * nreg) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(w, 0, region * nreg);
  hipDeviceSynchronize();
  const int blocks = 2048;
  const size_t f4_per_block = region / 16 / blocks;            // 16384 float4 = 256 KB per block
  int k = 0;
  auto base = [&]() { return w + (size_t)(k++ % nreg) * (region / 16); };
  for (int rep = 0; rep < 3; ++rep) {
    full_line<0><<<blocks, 256>>>(base(), out, f4_per_block);
    full_line<1><<<blocks, 256>>>(base(), out, f4_per_block);
    part_line<0, 128><<<blocks, 256>>>(base(), out, (int)(f4_per_block / 128));   // 2-KB rows: "half_line"
    part_line<1, 128><<<blocks, 256>>>(base(), out, (int)(f4_per_block / 128));
    part_line<0, 16><<<blocks, 256>>>(base(), out, (int)(f4_per_block / 16));     // 256-B rows: "quarter_line"
    part_line<1, 16><<<blocks, 256>>>(base(), out, (int)(f4_per_block / 16));
  }
  hipDeviceSynchronize();
  printf("{\"bytes_per_launch\": %zu, \"launches_per_kernel\": 3, \"kernels\": [\"full_line<0>\", \"full_line<1>\", \"part_line<0,128>\", "
         "\"part_line<1,128>\", \"part_line<0,16>\", \"part_line<1,16>\"]}\n", region);
  return 0;
}
