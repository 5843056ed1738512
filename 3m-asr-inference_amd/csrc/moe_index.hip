// MoE top-1 routing index math + row scatter / gather.
//
// Replaces (reference files under TRTAPI++/plugin/fmoe_expert_plugin/):
//   ScatterMappingKernel       fmoe_expert_kernel.cu:25-90    -> moe_index_kernel
//   ScatterMappingCopyKernel   fmoe_expert_kernel.cu:92-118   -> row_scatter_kernel   (FastMoE local_scatter)
//   GatherrMappingCopyKernel   fmoe_expert_kernel.cu:191-217  -> row_gather_kernel    (FastMoE local_gather)
//
// Contract (integer, bit-exact):  cnt[e] = #{i : g_i = e};
// acc[0] = 0, acc[e+1] = acc[e] + cnt[e];  mapping[i] = acc[g_i] + #{j < i : g_j = g_i};
// pos = mapping^-1.  The reference gets the within-expert rank from shared-memory atomicAdd
// arrival order (nondeterministic); here the rank is the STABLE one, computed without atomics on
// the rank path: a wave matches equal expert ids with log2(E) ballots, ranks them by popcount of
// the lower-lane mask, per-wave counts are prefix-summed over waves, and the per-expert offsets
// come from a wavefront prefix sum (shfl_up) over the histogram.  g_i < 0 (padded frame) or
// g_i >= E is dropped: mapping -1, not counted.  One workgroup; any S (1024-token chunks).
#include "common.h"
#include "kernels.h"

namespace m3 {

constexpr int kIdxMaxE = 256;
constexpr int kIdxMaxWaves = 16;

__device__ __forceinline__ unsigned long long match_expert(int key, bool active, int nbits) {
  unsigned long long mask = __ballot(active);
  for (int b = 0; b < nbits; ++b) {
    const bool bit = (key >> b) & 1;
    const unsigned long long bal = __ballot(bit);
    mask &= bit ? bal : ~bal;
  }
  return active ? mask : 0ull;
}

// Router gate on one lane: softmax + top-1 over WIDTH logits with the reference's arg-max tree
// (SoftmaxAndTop1KernelSmall, softmax_topk_kernel.cu:55-64: stride tree, strict '<').
template <int WIDTH>
__device__ __forceinline__ void gate_top1_lane(const float* __restrict__ row, int* idx_out, float* val_out) {
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
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < WIDTH; j += 4) {   // second pass over the row (L1-resident)
    const f32x4 t = ldg4(row + j);
#pragma unroll
    for (int q = 0; q < 4; ++q) sum += expf(t[q] - v[0]);
  }
  *idx_out = id[0];
  *val_out = 1.f / sum;
}

// GATE_W > 0: the kernel first forms gate_idx / gate_value from router logits [S][GATE_W] (one lane per
// token; padded frames t >= len[b] get idx -1 / value 0), i.e. SoftmaxTopK + ScatterMapping in one launch.
template <int GATE_W>
__global__ __launch_bounds__(1024) void moe_index_kernel(const int32_t* gate_in, int S, int E, int nbits,
                                                         int32_t* __restrict__ mapping, int32_t* __restrict__ acc_hist,
                                                         int32_t* __restrict__ pos, const float* __restrict__ logits,
                                                         const int32_t* __restrict__ row_len, int rows_per_batch,
                                                         int32_t* gate_out, float* __restrict__ gate_value) {
  __shared__ int hist[kIdxMaxE];
  __shared__ int running[kIdxMaxE];
  __shared__ int wcnt[kIdxMaxWaves][kIdxMaxE];
  __shared__ int woff[kIdxMaxWaves][kIdxMaxE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nwaves = nthreads >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int32_t* gate = gate_in;

  for (int e = tid; e < E; e += nthreads) hist[e] = 0;
  if (GATE_W > 0) {
    for (int i = tid; i < S; i += nthreads) {
      int gi = -1;
      float gv = 0.f;
      const bool live = row_len == nullptr || (i % rows_per_batch) < row_len[i / rows_per_batch];
      if (live) gate_top1_lane<(GATE_W > 0 ? GATE_W : 8)>(logits + (size_t)i * GATE_W, &gi, &gv);
      gate_out[i] = gi;
      gate_value[i] = gv;
    }
    gate = gate_out;   // re-read below by the same workgroup, after the barrier
  }
  __syncthreads();

  // pass 1: histogram (order-independent -> LDS atomics by the wave leaders are fine)
  for (int base = 0; base < S; base += nthreads) {
    const int i = base + tid;
    int g = (i < S) ? gate[i] : -1;
    const bool active = (g >= 0) && (g < E);
    const unsigned long long m = match_expert(g, active, nbits);
    if (active && (m & lt_mask) == 0ull) atomicAdd(&hist[g], __popcll(m));
  }
  __syncthreads();

  // wavefront prefix sum over experts -> acc_hist[0..E]; running[e] = exclusive offset
  if (wave == 0) {
    int carry = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int e = e0 + lane;
      const int v = (e < E) ? hist[e] : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      if (e < E) {
        running[e] = carry + incl - v;
        acc_hist[e + 1] = carry + incl;
      }
      carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) acc_hist[0] = 0;
  }
  __syncthreads();

  // pass 2: stable rank = (rows of earlier chunks) + (rows of earlier waves) + (lower lanes in my wave)
  for (int base = 0; base < S; base += nthreads) {
    const int i = base + tid;
    int g = (i < S) ? gate[i] : -1;
    const bool active = (g >= 0) && (g < E);
    const unsigned long long m = match_expert(g, active, nbits);
    const int rank = __popcll(m & lt_mask);
    for (int e = lane; e < E; e += 64) wcnt[wave][e] = 0;
    if (active && rank == 0) wcnt[wave][g] = __popcll(m);
    __syncthreads();
    for (int e = tid; e < E; e += nthreads) {
      int run = running[e];
      for (int w = 0; w < nwaves; ++w) {
        woff[w][e] = run;
        run += wcnt[w][e];
      }
      running[e] = run;
    }
    __syncthreads();
    if (i < S) {
      if (active) {
        const int dst = woff[wave][g] + rank;
        mapping[i] = dst;
        if (pos) pos[dst] = i;
      } else {
        mapping[i] = -1;
      }
    }
  }
}

int launch_moe_index(const int32_t* gate_idx, int S, int E, int32_t* mapping, int32_t* acc_hist, int32_t* pos,
                     hipStream_t stream) {
  M3_REQUIRE(E >= 1 && E <= kIdxMaxE, "moe_index: num_expert=%d out of range [1,%d]", E, kIdxMaxE);
  M3_REQUIRE(S >= 0, "moe_index: negative S");
  int nbits = 0;
  while ((1 << nbits) < E) ++nbits;
  int threads = S >= 1024 ? 1024 : (int)align_up(S > 0 ? S : 1, 64);
  hipLaunchKernelGGL(moe_index_kernel<0>, dim3(1), dim3(threads), 0, stream, gate_idx, S, E, nbits, mapping,
                     acc_hist, pos, nullptr, nullptr, 0, nullptr, nullptr);
  M3_LAUNCH_CHECK();
  return 0;
}

int launch_moe_gate_index(const float* logits, int width, const int32_t* row_len, int rows_per_batch, int S,
                          int32_t* gate_idx, float* gate_value, int32_t* mapping, int32_t* acc_hist, int32_t* pos,
                          hipStream_t stream) {
  M3_REQUIRE(S > 0, "moe_gate_index: empty batch");
  M3_REQUIRE(row_len == nullptr || rows_per_batch > 0, "moe_gate_index: rows_per_batch missing");
  int nbits = 0;
  while ((1 << nbits) < width) ++nbits;
  int threads = S >= 1024 ? 1024 : (int)align_up(S, 64);
#define M3_GI_CASE(W_)                                                                                        \
  hipLaunchKernelGGL(moe_index_kernel<W_>, dim3(1), dim3(threads), 0, stream, nullptr, S, W_, nbits, mapping, \
                     acc_hist, pos, logits, row_len, rows_per_batch, gate_idx, gate_value)
  switch (width) {
    case 8: M3_GI_CASE(8); break;
    case 16: M3_GI_CASE(16); break;
    case 32: M3_GI_CASE(32); break;
    case 64: M3_GI_CASE(64); break;
    default: M3_REQUIRE(false, "moe_gate_index: num_expert=%d must be 8/16/32/64", width);
  }
#undef M3_GI_CASE
  M3_LAUNCH_CHECK();
  return 0;
}

// ---- row scatter / gather: one wave per row, 16 B per lane per access, grid-stride over rows ----
template <bool GATHER>
__global__ __launch_bounds__(256) void row_permute_kernel(const uint4* __restrict__ in, const int32_t* __restrict__ mapping,
                                                          int S, int row16, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = blockIdx.x * 4 + wave; s < S; s += gridDim.x * 4) {
    const int m = mapping[s];
    if (GATHER) {
      uint4* dst = out + (size_t)s * row16;
      if (m >= 0) {
        const uint4* src = in + (size_t)m * row16;
        for (int c = lane; c < row16; c += 64) dst[c] = src[c];
      } else {
        for (int c = lane; c < row16; c += 64) dst[c] = uint4{0, 0, 0, 0};
      }
    } else if (m >= 0) {
      const uint4* src = in + (size_t)s * row16;
      uint4* dst = out + (size_t)m * row16;
      for (int c = lane; c < row16; c += 64) dst[c] = src[c];
    }
  }
}

static int launch_row_permute(bool gather, const void* in, const int32_t* mapping, int S, int row_bytes, void* out,
                              hipStream_t stream) {
  M3_REQUIRE(row_bytes > 0 && (row_bytes & 15) == 0, "row scatter/gather: row_bytes=%d must be a multiple of 16",
             row_bytes);
  if (S == 0) return 0;
  const int grid = min(cdiv(S, 4), 2048);
  if (gather)
    hipLaunchKernelGGL(row_permute_kernel<true>, dim3(grid), dim3(256), 0, stream, (const uint4*)in, mapping, S,
                       row_bytes / 16, (uint4*)out);
  else
    hipLaunchKernelGGL(row_permute_kernel<false>, dim3(grid), dim3(256), 0, stream, (const uint4*)in, mapping, S,
                       row_bytes / 16, (uint4*)out);
  M3_LAUNCH_CHECK();
  return 0;
}

int launch_local_scatter(const void* x, const int32_t* mapping, int S, int row_bytes, void* out, hipStream_t stream) {
  return launch_row_permute(false, x, mapping, S, row_bytes, out, stream);
}
int launch_local_gather(const void* buf, const int32_t* mapping, int S, int row_bytes, void* out, hipStream_t stream) {
  return launch_row_permute(true, buf, mapping, S, row_bytes, out, stream);
}

}  // namespace m3
