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
// g_i >= E is dropped: mapping -1, not counted.  One workgroup (1024-token chunks) up to 4096 tokens,
// one work-group per 1024 tokens beyond (moe_index_multi_kernel).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "moe_gate.h"

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

// Stable counting sort of the tokens by expert (see the header comment).  `gate` may point to global memory or to
// LDS (flat addressing); all threads of the (single) workgroup call this.
__device__ __forceinline__ void moe_index_body(const int32_t* gate, int S, int E, int nbits, int32_t* __restrict__ mapping,
                                               int32_t* __restrict__ acc_hist, int32_t* __restrict__ pos) {
  __shared__ int hist[kIdxMaxE];
  __shared__ int running[kIdxMaxE];
  __shared__ int wcnt[kIdxMaxWaves][kIdxMaxE];
  __shared__ int woff[kIdxMaxWaves][kIdxMaxE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nwaves = nthreads >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  for (int e = tid; e < E; e += nthreads) hist[e] = 0;
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

// GATE_W > 0: the kernel first forms gate_idx / gate_value from router logits [S][GATE_W] (one lane per
// token; padded frames t >= len[b] get idx -1 / value 0), i.e. SoftmaxTopK + ScatterMapping in one launch.
template <int GATE_W>
__global__ __launch_bounds__(1024) void moe_index_kernel(const int32_t* gate_in, int S, int E, int nbits,
                                                         int32_t* __restrict__ mapping, int32_t* __restrict__ acc_hist,
                                                         int32_t* __restrict__ pos, const float* __restrict__ logits,
                                                         const int32_t* __restrict__ row_len, int rows_per_batch,
                                                         int32_t* gate_out, float* __restrict__ gate_value) {
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int32_t* gate = gate_in;
  if (GATE_W > 0) {
    for (int i = tid; i < S; i += nthreads) {
      int gi = -1;
      float gv = 0.f;
      const bool live = row_len == nullptr || (i % rows_per_batch) < row_len[i / rows_per_batch];
      if (live) gate_top1_lane<(GATE_W > 0 ? GATE_W : 8)>(logits + (size_t)i * GATE_W, &gi, &gv);
      gate_out[i] = gi;
      gate_value[i] = gv;
    }
    gate = gate_out;   // re-read by the same workgroup after the barrier inside moe_index_body
    __syncthreads();
  }
  moe_index_body(gate, S, E, nbits, mapping, acc_hist, pos);
}

// Long batches: the same stable counting sort spread over S/1024 work-groups with no scratch memory and no
// inter-work-group synchronisation.  Work-group b owns tokens [1024 b, 1024 (b+1)); it histograms the WHOLE gate array
// (S * 4 bytes, L2-resident after the first work-group) into total[e] and, on the way, the part before its own chunk into
// before[e]; acc = exclusive scan of total, and its tokens land at acc[g] + before[g] + (stable rank inside the chunk).
// Every work-group derives the same acc; work-group 0 writes acc_hist.  Reads grow as S^2 / 1024 tokens (16 k tokens:
// 1 MB; 262 k tokens: 268 MB from L2), against a single work-group walking the array twice.
__global__ __launch_bounds__(1024) void moe_index_multi_kernel(const int32_t* __restrict__ gate, int S, int E, int nbits,
                                                               int32_t* __restrict__ mapping, int32_t* __restrict__ acc_hist,
                                                               int32_t* __restrict__ pos) {
  __shared__ int total[kIdxMaxE];
  __shared__ int before[kIdxMaxE];
  __shared__ int wcnt[kIdxMaxWaves][kIdxMaxE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int my0 = blockIdx.x * 1024;
  for (int e = tid; e < E; e += 1024) total[e] = before[e] = 0;
  __syncthreads();
  for (int base = 0; base < S; base += 1024) {
    const int i = base + tid;
    const int g = (i < S) ? gate[i] : -1;
    const bool active = (g >= 0) && (g < E);
    const unsigned long long m = match_expert(g, active, nbits);
    if (active && (m & lt_mask) == 0ull) {           // the lowest lane of each group of equal ids adds the group
      const int c = __popcll(m);
      atomicAdd(&total[g], c);
      if (base < my0) atomicAdd(&before[g], c);      // chunks are 1024-aligned: a whole iteration is before or not
    }
  }
  __syncthreads();
  if (wave == 0) {                                    // exclusive scan over experts; before[e] becomes this chunk's base
    int carry = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int e = e0 + lane;
      const int v = (e < E) ? total[e] : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      if (e < E) {
        before[e] += carry + incl - v;
        if (blockIdx.x == 0) acc_hist[e + 1] = carry + incl;
      }
      carry += __shfl(incl, 63, 64);
    }
    if (blockIdx.x == 0 && lane == 0) acc_hist[0] = 0;
  }
  // stable rank inside the chunk: earlier waves' counts + lower lanes of my wave
  const int i = my0 + tid;
  const int g = (i < S) ? gate[i] : -1;
  const bool active = (g >= 0) && (g < E);
  const unsigned long long m = match_expert(g, active, nbits);
  const int rank = __popcll(m & lt_mask);
  for (int e = lane; e < E; e += 64) wcnt[wave][e] = 0;
  if (active && rank == 0) wcnt[wave][g] = __popcll(m);
  __syncthreads();
  if (i < S) {
    if (active) {
      int dst = before[g] + rank;
      for (int w = 0; w < wave; ++w) dst += wcnt[w][g];
      mapping[i] = dst;
      if (pos) pos[dst] = i;
    } else {
      mapping[i] = -1;
    }
  }
}

// rows from which the multi-work-group form is used (env M3_INDEX_MULTI_MIN_ROWS; below it one work-group is faster)
static int index_multi_min_rows() {
  static const int v = [] {
    const char* e = getenv("M3_INDEX_MULTI_MIN_ROWS");
    return e ? atoi(e) : 4096;
  }();
  return v;
}

int launch_moe_index(const int32_t* gate_idx, int S, int E, int32_t* mapping, int32_t* acc_hist, int32_t* pos,
                     hipStream_t stream) {
  M3_REQUIRE(E >= 1 && E <= kIdxMaxE, "moe_index: num_expert=%d out of range [1,%d]", E, kIdxMaxE);
  M3_REQUIRE(S >= 0, "moe_index: negative S");
  int nbits = 0;
  while ((1 << nbits) < E) ++nbits;
  if (S >= index_multi_min_rows()) {
    hipLaunchKernelGGL(moe_index_multi_kernel, dim3(cdiv(S, 1024)), dim3(1024), 0, stream, gate_idx, S, E, nbits, mapping,
                       acc_hist, pos);
    M3_LAUNCH_CHECK();
    return 0;
  }
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


// ------------------------------------------------------------------------------------------------
// moe_route_kernel: the whole routing decision of one MoE layer in ONE single-workgroup launch
//   logits = embed_part[s] + LayerNorm(x[s]) . Wx^T (+ bias)      router matmul on cat([embed, x])
//                                                                   (positionwise_feed_forward.py:169-180,225)
//   gate_idx, gate_value = softmax-top1(logits)                    SoftmaxTopK plugin (softmax_topk_kernel.cu:26-120)
//   mapping, acc_histogram, pos                                    ScatterMapping (fmoe_expert_kernel.cu:25-90)
// for S <= 256 tokens.  The embed half of the router product does not depend on the layer's input and is
// computed for all layers at once by one GEMM (engine stage "router_e_all"); here only the K = D half is
// left: each wave = (16-token tile) x (K part), MFMA 16x16x4, LayerNorm applied on the output side with the
// affine folded into Wx (as gemm.hip LN_EPI), row sums taken from the A fragments.
template <int E, int NWV>
__global__ __launch_bounds__(64 * NWV) void moe_route_kernel(const float* __restrict__ x, int ldx, int D,
                                                         const float* __restrict__ wx, const float* __restrict__ wsum,
                                                         const float* __restrict__ bias, const float* __restrict__ eall,
                                                         int ld_e, float ln_eps, const int32_t* __restrict__ row_len,
                                                         int rows_per_batch, int S, int nbits,
                                                         int32_t* __restrict__ gate_idx, float* __restrict__ gate_value,
                                                         int32_t* __restrict__ mapping, int32_t* __restrict__ acc_hist,
                                                         int32_t* __restrict__ pos) {
  constexpr int NT = E / 16;
  __shared__ float part[NWV][16][E + 1];  // [unit][row][expert] partial logits
  __shared__ float psum[NWV][16][2];      // [unit][row] partial (sum x, sum x^2)
  __shared__ float s_wsum[E], s_bias[E];
  __shared__ int s_gate[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int ntile = (S + 15) >> 4;               // <= NWV
  const int KP = NWV / ntile;                    // K parts per tile (>= 1)
  const int nsteps = D >> 4;
  if (tid < E) {
    s_wsum[tid] = wsum[tid];
    s_bias[tid] = bias ? bias[tid] : 0.f;
  }
  // the embed half of this thread's logits (phase B) is requested now, so its latency hides under phase A
  constexpr int EPL0 = E / 16;
  float epre[EPL0];
  {
    const int tk = min(tid >> 4, S - 1);
#pragma unroll
    for (int j = 0; j < EPL0; ++j) epre[j] = eall ? eall[(size_t)tk * ld_e + (tid & 15) + 16 * j] : 0.f;
  }
  // ---- phase A: partial logits of unit (tile, kpart) = this wave ----
  const int tile = wave / KP, kp = wave - tile * KP;
  if (tile < ntile) {
    const int m = min(16 * tile + col, S - 1);
    const float* arow = x + (size_t)m * ldx + 4 * kq;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
    for (int s0 = kp; s0 < nsteps; s0 += 4 * KP) {   // 4 K-steps per trip, all loads issued before the MFMAs
      f32x4 a[4], b[4][NT];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sidx = min(s0 + i * KP, nsteps - 1);
        a[i] = ldg4(arow + (sidx << 4));
#pragma unroll
        for (int t = 0; t < NT; ++t) b[i][t] = ldg4(wx + (size_t)(16 * t + col) * D + (sidx << 4) + 4 * kq);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (s0 + i * KP < nsteps) {
          s1 += (a[i][0] + a[i][1]) + (a[i][2] + a[i][3]);
          s2 += (a[i][0] * a[i][0] + a[i][1] * a[i][1]) + (a[i][2] * a[i][2] + a[i][3] * a[i][3]);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t] = mfma16(a[i][j], b[i][t][j], acc[t]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][4 * kq + r][16 * t + col] = acc[t][r];
    s1 += __shfl_xor(s1, 16, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (kq == 0) {
      psum[wave][col][0] = s1;
      psum[wave][col][1] = s2;
    }
  }
  __syncthreads();
  // ---- phase B: 16 lanes per token (lane l owns experts l, l+16, ..): finish the logits, softmax + top-1 with
  //      the reference's arg-max tree (strides >= 16 inside the lane, 8/4/2/1 across lanes by DPP row shifts) ----
  constexpr int EPL = E / 16;                    // experts per lane
  for (int tok0 = 0; tok0 < S; tok0 += 4 * NWV) {
    const int tok = tok0 + (tid >> 4);
    const int l16 = tid & 15;
    const int tk = min(tok, S - 1);
    const int t = tk >> 4, r = tk & 15;
    float t1 = 0.f, t2 = 0.f;
    for (int p = 0; p < KP; ++p) {
      t1 += psum[t * KP + p][r][0];
      t2 += psum[t * KP + p][r][1];
    }
    const float mean = t1 / (float)D;
    const float rstd = rsqrtf(fmaxf(t2 / (float)D - mean * mean, 0.f) + ln_eps);
    float v[EPL], c[EPL];
    int id[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      const int e = l16 + 16 * j;
      float a = 0.f;
      for (int p = 0; p < KP; ++p) a += part[t * KP + p][r][e];
      const float ev = (tok0 == 0) ? epre[j] : (eall ? eall[(size_t)tk * ld_e + e] : 0.f);
      v[j] = rstd * (a - mean * s_wsum[e]) + s_bias[e] + ev;
      c[j] = v[j];
      id[j] = e;
    }
#pragma unroll
    for (int stride = EPL / 2; stride > 0; stride >>= 1)   // tree strides E/2 .. 16: partners live in this lane
#pragma unroll
      for (int j = 0; j < stride; ++j)
        if (c[j] < c[j + stride]) {
          c[j] = c[j + stride];
          id[j] = id[j + stride];
        }
    float best = c[0];
    int bi = id[0];
#define M3_TREE_STEP(STRIDE, CTRL)                                                            \
    {                                                                                         \
      const float ov = dpp_mov<CTRL>(best);                                                   \
      const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xF, 0xF, true);                \
      if (l16 < STRIDE && best < ov) {                                                        \
        best = ov;                                                                            \
        bi = oi;                                                                              \
      }                                                                                       \
    }
    M3_TREE_STEP(8, 0x108)   // row_shl:8  -> lane l reads lane l+8
    M3_TREE_STEP(4, 0x104)
    M3_TREE_STEP(2, 0x102)
    M3_TREE_STEP(1, 0x101)
#undef M3_TREE_STEP
    const float mx = __shfl(best, (tid & 63) & ~15, 64);     // lane 0 of the 16-lane group holds the winner
    const int mi = __shfl(bi, (tid & 63) & ~15, 64);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) sum += expf(v[j] - mx);
    sum = group16_sum(sum);
    if (tok < S && l16 == 0) {
      const bool live = row_len == nullptr || (tok % rows_per_batch) < row_len[tok / rows_per_batch];
      gate_idx[tok] = live ? mi : -1;
      gate_value[tok] = live ? 1.f / sum : 0.f;
      s_gate[tok] = live ? mi : -1;
    }
  }
  __syncthreads();
  // ---- phase C: stable counting sort by expert ----
  moe_index_body(s_gate, S, E, nbits, mapping, acc_hist, pos);
}

int launch_moe_route(const float* x, int ldx, int D, const float* wx, const float* wsum, const float* bias,
                     const float* eall, int ld_e, float ln_eps, const int32_t* row_len, int rows_per_batch, int S, int E,
                     int32_t* gate_idx, float* gate_value, int32_t* mapping, int32_t* acc_hist, int32_t* pos,
                     hipStream_t stream) {
  M3_REQUIRE(S > 0 && S <= 256, "moe_route: S=%d outside [1,256]", S);
  M3_REQUIRE((D & 15) == 0 && (ldx & 3) == 0, "moe_route: bad D / ldx");
  M3_REQUIRE(row_len == nullptr || rows_per_batch > 0, "moe_route: rows_per_batch missing");
  int nbits = 0;
  while ((1 << nbits) < E) ++nbits;
#define M3_ROUTE_CASE(E_)                                                                                          \
  hipLaunchKernelGGL((moe_route_kernel<E_, 16>), dim3(1), dim3(1024), 0, stream, x, ldx, D, wx, wsum, bias,       \
                     eall, ld_e, ln_eps, row_len, rows_per_batch, S, nbits, gate_idx, gate_value, mapping, acc_hist, pos)
  switch (E) {
    case 16: M3_ROUTE_CASE(16); break;
    case 32: M3_ROUTE_CASE(32); break;
    case 64: M3_ROUTE_CASE(64); break;
    default: M3_REQUIRE(false, "moe_route: num_expert=%d must be 16/32/64", E);
  }
#undef M3_ROUTE_CASE
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
