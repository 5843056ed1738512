// Split-K LDS-tiled fp32 GEMM for the two deep-K layers of the subsampling front end at short inputs.
//
//   conv2  (Conv2d 3x3 stride 2, C->C, as implicit GEMM):  M = B*T2*F2 = 450,  N = 512, K = 9*C = 4608
//   out.0  (Linear(C*F2 -> D)):                             M = B*T' = 50,      N = 512, K = 9728
// (reference: TensorRT convolution / matmul layers built by torch_network_helper.py:227-251,573-605 from
// layer/subsampling.py:103-145).  The K-split kernel of gemm.hip gives every 16-column workgroup its own copy of the
// im2col rows: at B=1 conv2 moved ~820 MB through L2 for 2.1 GFLOP and took 58 us, 5 % of a forward.  Here the output is
// cut into 64 x 64 tiles and K into `splits` ranges; one workgroup = (tile, K range) runs the pipeline of
// moe_expert_tiled_f32.hip (coalesced 16-B staging loads one k-step ahead, 2-stage LDS ring, 2x2 waves of 32x32,
// v_mfma_f32_16x16x4_f32) and writes its raw partial tile; splitk_reduce_kernel sums the partials in a fixed order
// and applies bias / ReLU / scale (deterministic, no atomics).  Partials live in the engine workspace.
#include "common.h"
#include "kernels.h"

namespace m3 {

namespace {
constexpr int SBM = 64, SBN = 64, SBK = 64;
constexpr int S_LD = SBK + 4;                        // floats per LDS row (272 B: conflict-free 16-B reads)
constexpr int SC_LD = SBN + 4;
constexpr int kSplitLdsBytes = 2 * (SBM + SBN) * S_LD * 4;   // 69,632 B (the 17 KB epilogue image reuses it)
}  // namespace

template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_f32_splitk_kernel(const GemmParams p, float* __restrict__ part,
                                                                 int ksteps_per_split) {
  constexpr int MT = SBM / 32, NT = SBN / 32;
  constexpr int CA = SBK / 4, RA = 256 / CA, JA = SBM / RA, JB = SBN / RA;
  extern __shared__ __attribute__((aligned(16))) unsigned char splitk_lds[];
  float* As = reinterpret_cast<float*>(splitk_lds);
  float* Bs = As + 2 * SBM * S_LD;
  float* Cs = reinterpret_cast<float*>(splitk_lds);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int n_tile = blockIdx.x % p.n_tiles, m_tile = blockIdx.x / p.n_tiles, split = blockIdx.y;
  const int m0 = m_tile * SBM, n0 = n_tile * SBN;
  const int nsteps_total = p.K / SBK;
  const int s_lo = split * ksteps_per_split, s_hi = min(s_lo + ksteps_per_split, nsteps_total);

  const int ac = tid % CA, ar0 = tid / CA;
  const float* aptr[JA];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int m = min(m0 + ar0 + RA * j, p.M - 1);
    if (CONV) {
      const int f2 = m % p.conv_F2;
      const int t2 = (m / p.conv_F2) % p.conv_T2;
      const int b = m / (p.conv_F2 * p.conv_T2);
      aptr[j] = p.A + ((size_t)(b * p.conv_T1 + 2 * t2) * p.conv_F1 + 2 * f2) * p.conv_C + 4 * ac;
    } else {
      aptr[j] = p.A + (size_t)m * p.lda + 4 * ac;
    }
  }
  const float* bptr[JB];
#pragma unroll
  for (int j = 0; j < JB; ++j) bptr[j] = p.W + (size_t)min(n0 + ar0 + RA * j, p.N - 1) * p.K + 4 * ac;
  auto a_offset = [&](int k) -> int {
    if (CONV) {
      const int seg = k / p.conv_C, c = k - seg * p.conv_C;
      const int kh = seg / 3, kw = seg - kh * 3;
      return (kh * p.conv_F1 + kw) * p.conv_C + c;
    }
    return k;
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 areg[JA], breg[JB];
  auto load_tiles = [&](int s) {
    const int ko = a_offset(s * SBK);
#pragma unroll
    for (int j = 0; j < JA; ++j) areg[j] = ldg4(aptr[j] + ko);
#pragma unroll
    for (int j = 0; j < JB; ++j) breg[j] = ldg4_w(bptr[j] + s * SBK);
  };
  auto store_tiles = [&](int buf) {
    float* a_dst = As + buf * (SBM * S_LD) + ar0 * S_LD + 4 * ac;
    float* b_dst = Bs + buf * (SBN * S_LD) + ar0 * S_LD + 4 * ac;
#pragma unroll
    for (int j = 0; j < JA; ++j) *reinterpret_cast<f32x4*>(a_dst + RA * j * S_LD) = areg[j];
#pragma unroll
    for (int j = 0; j < JB; ++j) *reinterpret_cast<f32x4*>(b_dst + RA * j * S_LD) = breg[j];
  };

  load_tiles(s_lo);
  store_tiles(0);
  __syncthreads();
  for (int s = s_lo; s < s_hi; ++s) {
    const int i = s - s_lo;
    load_tiles(min(s + 1, s_hi - 1));                // clamped, unconditional (see gemm_bf16_tiled.hip)
    __builtin_amdgcn_sched_barrier(0);
    const float* a_lds = As + (i & 1) * (SBM * S_LD) + ((SBM / 2) * wm + col) * S_LD + 4 * kq;
    const float* b_lds = Bs + (i & 1) * (SBN * S_LD) + ((SBN / 2) * wn + col) * S_LD + 4 * kq;
#pragma unroll
    for (int ks = 0; ks < SBK / 16; ++ks) {
      f32x4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(b_lds + 16 * nt * S_LD + 16 * ks);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(a_lds + 16 * mt * S_LD + 16 * ks);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[mt][nt] = mfma16(a[j], b[nt][j], acc[mt][nt]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    store_tiles((i + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[((SBM / 2) * wm + 16 * mt + 4 * kq + r) * SC_LD + (SBN / 2) * wn + 16 * nt + col] = acc[mt][nt][r];
  __syncthreads();
  // raw partial tile -> part[split][m][n] (row-wise, float4 per lane)
  const int c4 = 4 * (lane & 15), n = n0 + c4;
  for (int it = 0; it < SBM / 16; ++it) {
    const int row = (4 * it + wave) * 4 + (lane >> 4);
    const int m = m0 + row;
    if (m < p.M && n < p.N) stg4(part + ((size_t)split * p.M + m) * p.N + n, *reinterpret_cast<const f32x4*>(Cs + row * SC_LD + c4));
  }
}

// y[m][n] = alpha * act(bias[n] + sum_s part[s][m][n]), fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int N,
                                                            const float* __restrict__ bias, int act, float alpha,
                                                            float* __restrict__ y, int ldy) {
  const size_t quads = (size_t)M * (N / 4);
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)gridDim.x * 256) {
    const int m = (int)(q / (N / 4)), n = 4 * (int)(q % (N / 4));
    f32x4 v = bias ? ldg4(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < splits; s0 += 4) {          // 4 partials in flight
      f32x4 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        t[j] = (s0 + j < splits) ? ldg4(part + ((size_t)(s0 + j) * M + m) * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) v += t[j];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = v[e];
      if (act == ACT_RELU) t = fmaxf(t, 0.f);
      if (act == ACT_SILU) t = silu(t);
      v[e] = t * alpha;
    }
    stg4(y + (size_t)m * ldy + n, v);
  }
}

int init_gemm_f32_splitk_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f32_splitk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLdsBytes));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f32_splitk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLdsBytes));
  once.mark();
  return 0;
}

// How many K ranges (0 = do not use this path) and the workspace they need.  Used for deep, narrow problems only:
// few output tiles, long K, plain epilogue (bias / ReLU / SiLU / scale; no LayerNorm, mask, GLU, residual, concat).
int gemm_f32_splitk_plan(const GemmParams& p, size_t* ws_bytes) {
  if (ws_bytes) *ws_bytes = 0;
  if (p.w_bf16 || p.K < 4096 || (p.K & 63) || (p.N & 3) || (p.lda & 3) || (p.ldy & 3)) return 0;
  if (p.mode == GEMM_A_CONCAT2 || p.ln_wsum || p.ln_gamma || p.mask_in || p.mask_out || p.resid || p.act == ACT_GLU) return 0;
  if (p.mode == GEMM_A_CONV3X3S2 && (p.conv_C & 63)) return 0;
  const long tiles = (long)cdiv(p.M, SBM) * cdiv(p.N, SBN);
  if (tiles > 160) return 0;                          // enough tiles: the other kernels fill the chip
  const int nsteps = p.K / SBK;
  int splits = (int)(512 / tiles);                    // ~2 workgroups per CU
  if (splits > nsteps / 4) splits = nsteps / 4;       // >= 4 k-steps per workgroup
  if (splits < 2) return 0;
  const int per = cdiv(nsteps, splits);
  splits = cdiv(nsteps, per);
  if (ws_bytes) *ws_bytes = (size_t)splits * p.M * p.N * sizeof(float);
  return splits;
}

int launch_gemm_f32_splitk(const GemmParams& pin, float* ws, size_t ws_bytes, hipStream_t stream) {
  GemmParams p = pin;
  size_t need = 0;
  const int splits = gemm_f32_splitk_plan(p, &need);
  M3_REQUIRE(splits >= 2 && ws != nullptr && ws_bytes >= need, "gemm split-K: not applicable / workspace %zu < %zu", ws_bytes, need);
  if (int rc = init_gemm_f32_splitk_kernels()) return rc;
  p.m_tiles = cdiv(p.M, SBM);
  p.n_tiles = cdiv(p.N, SBN);
  const int per = cdiv(p.K / SBK, splits);
  dim3 grid(p.m_tiles * p.n_tiles, splits);
  if (p.mode == GEMM_A_CONV3X3S2)
    hipLaunchKernelGGL((gemm_f32_splitk_kernel<true>), grid, dim3(256), kSplitLdsBytes, stream, p, ws, per);
  else
    hipLaunchKernelGGL((gemm_f32_splitk_kernel<false>), grid, dim3(256), kSplitLdsBytes, stream, p, ws, per);
  const size_t quads = (size_t)p.M * (p.N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid1d(quads, 1024)), dim3(256), 0, stream, ws, splits, p.M, p.N, p.bias, p.act,
                     p.alpha, p.Y, p.ldy);
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
