// Skinny fp32 GEMM  Y[M,N] = epilogue( prologue(A)[M,K] . W[N,K]^T )  on v_mfma_f32_16x16x4_f32.
//
// Replaces the TensorRT-native layers of the reference hot path (no reference kernel source):
//   Linear  = constant-weight matmul + bias add   (TRTAPI++/python/trt_helper/torch_network_helper.py:573-605)
//   Conv1d point-wise (k=1)                        (torch_network_helper.py:199-225)
//   Conv2d 3x3 stride 2 (second subsampling conv)  (torch_network_helper.py:227-251) as implicit GEMM
//   router matmul on cat([embed, x])               (trainer_3m_fix/layer/positionwise_feed_forward.py:169-180,225)
// with the surrounding element-wise layers fused as prologue / epilogue:
//   LayerNorm on the A rows (layer_norm_kernel.cu:33-139, but WITH eps),
//   masked_fill(0) of padded frames before / after (masked_fill_kernel.cu:27-54),
//   bias, ReLU / SiLU / GLU (glu_kernel.cu:26-44), uniform scale + residual add
//   (tensor_network_helper.py:406-471).
//
// Shape regime: M = B*T' tokens (50 .. ~2000), N, K in 512..4608: every weight element is used by
// only M rows, so this is a weight-streaming kernel whose run time at M = 50 is a few microseconds,
// i.e. it is bound by memory latency and by the number of instructions a wave issues.  Layout:
// one workgroup = 16 output columns (x2 for GLU) x 16*MT rows; its NW waves split K round-robin in
// 16-deep steps; each wave streams its W rows and A rows straight into VGPRs (float4 per lane = 16
// rows x 64 B per instruction), two groups of G steps in flight, all loads unconditional (clamped
// addresses: a branch around a load makes hipcc wait per load); partial tiles are summed through LDS in
// a fixed order (no atomics -> bitwise reproducible).  Small M is split into 16-row tiles so a
// 50-token utterance spreads over 128-256 workgroups; the tiles of one weight column block sit on
// the same XCD (blockIdx % 8) so the weight tile is fetched from HBM once.
//
// LayerNorm has two forms:
//  * LN_EPI (engine): the affine is folded into W / bias when the plan is packed, and the
//    normalisation moves to the OUTPUT side:  y_n = rstd_m * (acc_mn - mean_m * wsum_n) + bias'_n  with
//    wsum_n = sum_k W'_nk.  The GEMM runs on the raw rows, the row sums (sum a, sum a^2) are taken from
//    the A fragments already in registers, so LayerNorm costs no extra memory round trip and no barrier.
//  * LN_PRO (generic m3_linear with gamma / beta): statistics (two-pass) + affine applied to the A
//    fragments before the MFMAs; can also write the normalised rows out (ln_out).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

// -DM3_GEMM_DIAG: phase stamps (s_memtime) of every work-group of gemm_f32_kernel into a debug buffer (tools/diag_gemm_f32.py)
#ifdef M3_GEMM_DIAG
#define M3_GDIAG(...) __VA_ARGS__
#else
#define M3_GDIAG(...)
#endif

namespace m3 {

M3_GDIAG(__device__ unsigned long long g_gemm_dbg[2048 * 8];)

// K-steps per in-flight load group: 2 buffers x G x (NT + MT) float4 must fit the per-lane register
// budget (512 VGPR+AGPR for 4 waves, 256 for 8, 128 for 16 waves per workgroup) without spilling.
constexpr int gemm_group_steps(int MT, int NT, int NW) {
  if (NW == 16) return MT == 1 ? (NT == 1 ? 4 : 2) : (MT == 2 ? 2 : 1);
  if (NW == 8) return MT == 1 ? 8 : (MT == 2 ? (NT == 1 ? 6 : 4) : (NT == 1 ? 3 : 2));
  return MT == 1 ? 8 : (MT == 2 ? 6 : 4);
}

enum { LN_NONE = 0, LN_EPI = 1, LN_PRO = 2 };

// NW = waves per workgroup = K-split factor (4 / 8 / 16 for K ~ 512 / 1024 / >= 2048).
// NBUF = 1 when a wave's share of K fits one load group (K <= 16*NW*G: every block GEMM of the model): half the
// staging registers -> <= 128 VGPRs -> 4 waves per SIMD, so kernels of other streams can share the CU.
// The kernel body as a device function of (parameter block, work-group id): gemm_f32_kernel runs one problem,
// gemm_f32_dual_kernel two INDEPENDENT problems of the same instantiation in one launch (work-groups [0, n0) the first,
// the rest the second) -- the embed encoder and the main encoder's first block do not depend on each other until that block's
// router, so their GEMMs can share launches (engine.hip: "horizontal fusion"; a launch saved is ~6 us of a B = 1 forward).
template <int MT, bool GLU, int NW, bool CONV, int LN, int NBUF>
__device__ __forceinline__ void gemm_f32_body(const GemmParams& p, const int bid) {
  constexpr int NT = GLU ? 2 : 1;
  constexpr int G = gemm_group_steps(MT, NT, NW);
  constexpr int RW = (16 * MT) / NW;   // rows per wave in the LN_PRO prologue (>= 1)
  static_assert(RW >= 1, "more waves than rows");
  __shared__ float red[NW][MT * NT][256];
  __shared__ float stats[16 * MT][2];
  __shared__ float rsum[LN == LN_EPI ? NW : 1][16 * MT][2];
  __shared__ __attribute__((aligned(16))) float ln_g[LN == LN_PRO ? 1024 : 4], ln_b[LN == LN_PRO ? 1024 : 4];

  M3_GDIAG(unsigned long long dg[8]; dg[0] = __builtin_amdgcn_s_memtime();)
  // ONE batch of kernel-argument loads.  hipcc fetches each field of the by-value parameter block where it is first used: the
  // prologue was ~8 dependent rounds of s_load + s_waitcnt (2 400 cycles = 1.1 us before the first global load was issued,
  // in-kernel stamps, tools/diag_gemm_f32.py).  Naming the fields as scalar operands of one empty asm statement makes it load
  // them together, one wait.
  asm volatile("" ::"s"(p.A), "s"(p.lda), "s"(p.W), "s"(p.bias), "s"(p.Y), "s"(p.ldy), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.mode),
               "s"(p.n_tiles), "s"(p.m_tiles), "s"(p.xcd_swizzle), "s"(p.m_dev), "s"(p.resid), "s"(p.ldr), "s"(p.act), "s"(p.alpha),
               "s"(p.mask_in), "s"(p.mask_out), "s"(p.row_len), "s"(p.rows_per_batch));
  if (LN == LN_EPI) asm volatile("" ::"s"(p.ln_wsum), "s"(p.ln_wbeta), "s"(p.ln_eps));
  if (LN == LN_PRO) asm volatile("" ::"s"(p.ln_gamma), "s"(p.ln_beta), "s"(p.ln_eps), "s"(p.ln_out), "s"(p.ld_ln_out), "s"(p.ln_on_a2));
  if (!CONV) asm volatile("" ::"s"(p.A2), "s"(p.lda2), "s"(p.K1));
  if (CONV) asm volatile("" ::"s"(p.conv_T1), "s"(p.conv_F1), "s"(p.conv_T2), "s"(p.conv_F2), "s"(p.conv_C));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int Nout = GLU ? (p.N >> 1) : p.N;

  // ---- workgroup -> (column tile, row tile); XCD-aware when the column-tile count allows ----
  int n_tile, m_tile;
  {
    const int id = bid;
    if (p.xcd_swizzle) {
      const int j = id >> 3;
      m_tile = j % p.m_tiles;
      n_tile = (j / p.m_tiles) * 8 + (id & 7);
    } else {
      n_tile = id % p.n_tiles;
      m_tile = id / p.n_tiles;
    }
  }
  const int n0 = n_tile * 16;
  const int m0 = m_tile * (16 * MT);
  if (p.m_dev != nullptr && m0 > *p.m_dev) return;   // packed ragged batch: no live row in this tile
  const int ln_k0 = p.ln_on_a2 ? p.K1 : 0;             // first K index the LayerNorm applies to
  const int Kl = p.K - ln_k0;                          // LayerNorm row width

  // ---- per-lane row descriptors ----
  const float* arow[MT];
  const float* arow2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = min(m0 + 16 * mt + col, p.M - 1);
    arow2[mt] = nullptr;
    if (CONV) {
      const int f2 = m % p.conv_F2;
      const int t2 = (m / p.conv_F2) % p.conv_T2;
      const int b = m / (p.conv_F2 * p.conv_T2);
      arow[mt] = p.A + ((size_t)(b * p.conv_T1 + 2 * t2) * p.conv_F1 + 2 * f2) * p.conv_C + 4 * kq;
    } else {
      arow[mt] = p.A + (size_t)m * p.lda + 4 * kq;
      if (p.mode == GEMM_A_CONCAT2) arow2[mt] = p.A2 + (size_t)m * p.lda2 + 4 * kq;
    }
  }
  const float* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    wrow[t] = p.W + (size_t)min(n0 + t * Nout + col, p.N - 1) * p.K + 4 * kq;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr bool DUAL = MT * NT <= 2;
  f32x4 acc2[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc2[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[MT], s2[MT];   // LN_EPI: this lane's share of sum(a), sum(a^2) of row (16*mt + col)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) s1[mt] = s2[mt] = 0.f;

  const int nsteps = p.K >> 4;
  auto a_offset = [&](int k) -> int {  // wave-uniform k (multiple of 16) -> element offset in the A row
    if (CONV) {
      const int seg = k / p.conv_C, c = k - seg * p.conv_C;
      const int kh = seg / 3, kw = seg - kh * 3;
      return (kh * p.conv_F1 + kw) * p.conv_C + c;
    }
    return k;
  };

  // two groups of G K-steps in flight (registers only; nothing is shared between waves)
  f32x4 wbuf[NBUF][G][NT], abuf[NBUF][G][MT];
  auto load_group = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      // steps past the end re-load the last step (results unused): no branch around the loads
      const int s = min(wave + NW * (G * g + i), nsteps - 1);
      const int k = s << 4;
#pragma unroll
      for (int t = 0; t < NT; ++t) wbuf[buf][i][t] = ldg4_w(wrow[t] + k);
      const bool second = !CONV && p.mode == GEMM_A_CONCAT2 && k >= p.K1;
      const int off = second ? (k - p.K1) : a_offset(k);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) abuf[buf][i][mt] = ldg4((second ? arow2[mt] : arow[mt]) + off);
    }
  };

  const int ngroups = (nsteps + NW * G - 1) / (NW * G);
  M3_GDIAG(dg[1] = __builtin_amdgcn_s_memtime();)
  // (Compile-time load offsets for the common exact shape -- one base address per row, no per-load address arithmetic -- were
  //  tried: issue-to-data time unchanged (4 830 vs 4 650 cycles).  The loads are not issue-bound: a CU keeps ~16 KB of misses in
  //  flight whatever the instruction stream, so 64 KB per work-group take ~4 latencies.)
  load_group(0, 0);
  M3_GDIAG(dg[2] = __builtin_amdgcn_s_memtime();)
  // (everything below is issued behind the first group of operand loads: it is needed at epilogue time only)
  // the epilogue wave's bias / residual are requested up front (they would otherwise be a dependent
  // memory round trip at the very end of a microsecond-scale kernel)
  const int ep_mt = wave;              // wave w finishes row sub-tile w (w < MT)
  const bool is_ep = wave < MT;
  const int ep_n = n0 + col;
  float bias0 = 0.f, bias1 = 0.f, wsum0 = 0.f, wsum1 = 0.f, wbeta0 = 0.f, wbeta1 = 0.f, res[4] = {0.f, 0.f, 0.f, 0.f};
  if (is_ep && ep_n < Nout) {
    if (p.bias) {
      bias0 = p.bias[ep_n];
      if (GLU) bias1 = p.bias[ep_n + Nout];
    }
    if (LN == LN_EPI) {
      wsum0 = p.ln_wsum[ep_n];
      if (GLU) wsum1 = p.ln_wsum[ep_n + Nout];
      if (p.mask_in) {
        wbeta0 = p.ln_wbeta[ep_n];
        if (GLU) wbeta1 = p.ln_wbeta[ep_n + Nout];
      }
    }
    if (p.resid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = min(m0 + 16 * ep_mt + 4 * kq + r, p.M - 1);
        res[r] = p.resid[(size_t)m * p.ldr + ep_n];
      }
    }
  }
  // padded frames (masked_fill(0) before / after the layer, masked_fill_kernel.cu:27-54): resolved here for the rows this lane
  // finishes.  An input-masked row has a zero A row, i.e. a zero accumulator: the epilogue writes that instead of zeroing the
  // A fragments in front of every MFMA (a select + hazard nop per MFMA on the critical path of a one-tile wave)
  bool pad4[4] = {false, false, false, false};
  if (is_ep && (p.mask_in || p.mask_out)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = min(m0 + 16 * ep_mt + 4 * kq + r, p.M - 1);
      pad4[r] = (m % p.rows_per_batch) >= p.row_len[m / p.rows_per_batch];
    }
  }


  // ---- LN_PRO: row statistics (two-pass) + gamma/beta to LDS, while the first loads are in flight ----
  float a_mean[MT], a_rstd[MT];
  float* ln_out_row[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    a_mean[mt] = 0.f;
    a_rstd[mt] = 1.f;
    ln_out_row[mt] = nullptr;
  }
  if (LN == LN_PRO) {
    const float* lnA = p.ln_on_a2 ? p.A2 : p.A;
    const int ldl = p.ln_on_a2 ? p.lda2 : p.lda;
    for (int k = threadIdx.x * 4; k < Kl; k += 256 * NW) {
      *reinterpret_cast<f32x4*>(&ln_g[k]) = ldg4(p.ln_gamma + k);
      *reinterpret_cast<f32x4*>(&ln_b[k]) = ldg4(p.ln_beta + k);
    }
    // one 16-lane row group per A row (4 rows per wave pass): 256 B of a row per load instruction
    const int q = lane >> 4, l16 = lane & 15;
    const int nj = (Kl + 63) >> 6;                     // float4 per lane per row (<= 16 -> rows up to 1024 wide)
    for (int c = 0; c < (RW + 3) / 4; ++c) {
      const int rloc = wave * RW + 4 * c + q;
      const bool row_on = (4 * c + q) < RW;
      const int m = min(m0 + min(rloc, 16 * MT - 1), p.M - 1);
      const float* row = lnA + (size_t)m * ldl;
      f32x4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = 4 * (l16 + 16 * j);
        const f32x4 t = ldg4(row + min(k, Kl - 4));
        const bool in = j < nj && k < Kl;
        v[j] = f32x4{in ? t[0] : 0.f, in ? t[1] : 0.f, in ? t[2] : 0.f, in ? t[3] : 0.f};
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
      const float mean = group16_sum(sum) / (float)Kl;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = 4 * (l16 + 16 * j);
        const float on = (j < nj && k < Kl) ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = (v[j][e] - mean) * on;
          sq += d * d;
        }
      }
      const float var = group16_sum(sq) / (float)Kl;
      if (row_on && l16 == 0) {
        stats[rloc][0] = mean;
        stats[rloc][1] = rsqrtf(var + p.ln_eps);
      }
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      a_mean[mt] = stats[16 * mt + col][0];
      a_rstd[mt] = stats[16 * mt + col][1];
      const int m = m0 + 16 * mt + col;
      if (p.ln_out != nullptr && n_tile == 0 && m < p.M) ln_out_row[mt] = p.ln_out + (size_t)m * p.ld_ln_out + 4 * kq;
    }
  }

  auto compute_group = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int s = wave + NW * (G * g + i);
      if (s < nsteps) {
        const int k = s << 4;
        const bool in_ln = k >= ln_k0;
        f32x4 g4, b4;
        if (LN == LN_PRO && in_ln) {
          g4 = *reinterpret_cast<const f32x4*>(&ln_g[(k - ln_k0) + 4 * kq]);
          b4 = *reinterpret_cast<const f32x4*>(&ln_b[(k - ln_k0) + 4 * kq]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 a = abuf[buf][i][mt];
          if (LN == LN_PRO && in_ln) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = (a[j] - a_mean[mt]) * a_rstd[mt] * g4[j] + b4[j];
            if (ln_out_row[mt] != nullptr) stg4(ln_out_row[mt] + (k - ln_k0), a);
          }
          if (LN == LN_EPI && in_ln) {
            s1[mt] += (a[0] + a[1]) + (a[2] + a[3]);
            s2[mt] += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (DUAL) {       // two independent accumulation chains (k even / odd inside the step): the fp32 MFMA has a 40-cycle
              //                 dependent latency against a 32-cycle issue, and a one-tile wave has nothing else to interleave
#pragma unroll
              for (int j = 0; j < 4; j += 2) {
                acc[mt][t] = mfma16(a[j], wbuf[buf][i][t][j], acc[mt][t]);
                acc2[mt][t] = mfma16(a[j + 1], wbuf[buf][i][t][j + 1], acc2[mt][t]);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[mt][t] = mfma16(a[j], wbuf[buf][i][t][j], acc[mt][t]);
            }
          }
        }
      }
    }
  };

  M3_GDIAG(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dg[3] = __builtin_amdgcn_s_memtime();)
  if (NBUF == 1) {
    compute_group(0, 0);
  } else {
    for (int g = 0; g < ngroups; g += 2) {
      if (g + 1 < ngroups) load_group(g + 1, NBUF - 1);
      compute_group(g, 0);
      if (g + 1 < ngroups) {
        if (g + 2 < ngroups) load_group(g + 2, 0);
        compute_group(g + 1, NBUF - 1);
      }
    }
  }

  if (DUAL) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] += acc2[mt][t];
  }
  M3_GDIAG(asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); dg[4] = __builtin_amdgcn_s_memtime();)
  // ---- cross-wave K reduction through LDS, then epilogue (wave w finishes row sub-tile w) ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][mt * NT + t][r * 64 + lane] = acc[mt][t][r];
  if (LN == LN_EPI) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float a1 = s1[mt], a2 = s2[mt];           // fold the 4 k-quarters of this wave (lanes col, col+16, +32, +48)
      a1 += __shfl_xor(a1, 16, 64);
      a2 += __shfl_xor(a2, 16, 64);
      a1 += __shfl_xor(a1, 32, 64);
      a2 += __shfl_xor(a2, 32, 64);
      if (kq == 0) {
        rsum[wave][16 * mt + col][0] = a1;
        rsum[wave][16 * mt + col][1] = a2;
      }
    }
  }
  __syncthreads();
  M3_GDIAG(dg[5] = __builtin_amdgcn_s_memtime();)

  if (is_ep) {
    const int mt = ep_mt;
    f32x4 v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sum = 0.f;   // fixed summation order over the K-split -> bitwise reproducible
#pragma unroll
        for (int w = 0; w < NW; w += 4)
          sum += (red[w][mt * NT + t][r * 64 + lane] + red[w + 1][mt * NT + t][r * 64 + lane]) +
                 (red[w + 2][mt * NT + t][r * 64 + lane] + red[w + 3][mt * NT + t][r * 64 + lane]);
        v[t][r] = sum;
      }
    if (ep_n < Nout) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * mt + 4 * kq + r;
        const int m = m0 + row;
        if (m >= p.M) continue;
        float y0 = v[0][r], y1 = v[NT - 1][r];
        const bool pad = pad4[r];
        if (LN != LN_EPI && p.mask_in && pad) y0 = y1 = 0.f;      // zero A row -> zero accumulator
        if (LN == LN_EPI) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            t1 += rsum[w][row][0];
            t2 += rsum[w][row][1];
          }
          const float mean = t1 / (float)Kl;
          const float var = fmaxf(t2 / (float)Kl - mean * mean, 0.f);
          const float rstd = rsqrtf(var + p.ln_eps);
          if (p.mask_in && pad) {   // masked_fill(0) after the LayerNorm: the row contributes the plain bias only
            y0 = -wbeta0;
            y1 = -wbeta1;
          } else {
            y0 = rstd * (y0 - mean * wsum0);
            y1 = rstd * (y1 - mean * wsum1);
          }
        }
        float y = y0 + bias0;
        if (GLU) y = y * sigmoidf(y1 + bias1);
        if (p.act == ACT_RELU) y = fmaxf(y, 0.f);
        if (p.act == ACT_SILU) y = silu(y);
        if (p.mask_out && pad) y = 0.f;
        y *= p.alpha;
        if (p.resid) y += res[r];
        p.Y[(size_t)m * p.ldy + ep_n] = y;
      }
    }
  }
  M3_GDIAG(dg[6] = __builtin_amdgcn_s_memtime();
           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
           dg[7] = __builtin_amdgcn_s_memtime();
           if (lane == 0 && wave == 0 && bid < 2048) {
             unsigned long long* o = g_gemm_dbg + (size_t)bid * 8;
             for (int i = 0; i < 8; ++i) o[i] = dg[i];
             // where the work-group ran: HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20), packed above the last stamp's low 40 bits
             const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
             o[7] = (dg[7] & 0xffffffffffull) | ((unsigned long long)(hw & 0xffff) << 40) | ((unsigned long long)(xcc & 0xf) << 56);
           })
}

template <int MT, bool GLU, int NW, bool CONV, int LN, int NBUF>
__global__ __launch_bounds__(64 * NW) void gemm_f32_kernel(const GemmParams p) {
  gemm_f32_body<MT, GLU, NW, CONV, LN, NBUF>(p, (int)blockIdx.x);
}
// (the second problem's work-group ids start at a multiple of 8, so that "id % 8 = XCD" holds for its XCD-aware tile order too)
template <int MT, bool GLU, int NW, int LN>
__global__ __launch_bounds__(64 * NW) void gemm_f32_dual_kernel(const GemmParams p0, const GemmParams p1, const int n0) {
  if ((int)blockIdx.x < n0) {
    if ((int)blockIdx.x < p0.n_tiles * p0.m_tiles) gemm_f32_body<MT, GLU, NW, false, LN, 1>(p0, (int)blockIdx.x);
  } else {
    gemm_f32_body<MT, GLU, NW, false, LN, 1>(p1, (int)blockIdx.x - n0);
  }
}

int launch_gemm_f32_tiled(const GemmParams& p, hipStream_t stream);   // gemm_f32_tiled.hip
bool gemm_f32_tiled_supports(const GemmParams& p);
static int f32_tiled_min_rows() {   // below this many rows the K-split kernel fills the chip better (M3_TILED_MIN_ROWS overrides)
  static const int v = [] {
    const char* e = getenv("M3_TILED_MIN_ROWS");
    return e ? atoi(e) : 384;
  }();
  return v;
}

// long batches: LDS-tiled kernel (everything but the affine-LayerNorm / concat router GEMM, whose output is 32 wide)
// (needs enough 64 x 64 tiles to occupy the chip: below ~160 the K-split kernel's many small workgroups win)
static bool gemm_f32_uses_tiled(const GemmParams& p) {
  const bool glu = p.act == ACT_GLU;
  return p.M >= f32_tiled_min_rows() && (long)cdiv(p.M, 64) * cdiv(glu ? p.N / 2 : p.N, 64) >= 160 && gemm_f32_tiled_supports(p);
}

// which kernel launch_gemm_f32 / the engine's split-K front end will run for this problem (sizes / mode only)
const char* gemm_kernel_label(const GemmParams& p, bool splitk) {
  if (splitk) return "gemm_f32_splitk_kernel";
  if (p.w_bf16 && gemm_bf16w_uses_dma(p)) return "gemm_bf16_dma_kernel";
  if (p.w_bf16) return gemm_bf16w_uses_tiled(p) ? "gemm_bf16w_tiled_kernel" : "gemm_bf16w_kernel";
  return gemm_f32_uses_tiled(p) ? "gemm_f32_tiled_kernel" : "gemm_f32_kernel";
}

// The instantiation launch_gemm_f32 runs for a problem (skinny fp32 kernel only): what two problems must share to be launched
// together.  ok = false: not this kernel (bf16 weights, the LDS-tiled form) or not one of the dual instantiations.
struct GemmVariant { bool ok; int mt, nw, ln; bool glu; };
static GemmVariant gemm_f32_variant(const GemmParams& p) {
  GemmVariant v{false, 0, 0, 0, false};
  if (p.w_bf16 || p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K & 15) || p.mode != GEMM_A_PLAIN || p.ln_gamma != nullptr || p.m_dev != nullptr) return v;
  if (gemm_f32_uses_tiled(p)) return v;
  v.glu = p.act == ACT_GLU;
  const int Nout = v.glu ? p.N / 2 : p.N;
  int mt = p.M <= 128 ? 1 : (p.M <= 512 ? 2 : 4);
  while (mt < 4 && 16 * mt < p.M && (long)cdiv(Nout, 16) * cdiv(p.M, 16 * mt) > 512) mt *= 2;
  while (mt > 1 && (long)cdiv(Nout, 16) * cdiv(p.M, 16 * mt) < 256) mt /= 2;
  v.mt = mt;
  v.nw = p.K >= 2048 ? 16 : (p.K >= 1024 ? 8 : 4);
  v.ln = p.ln_wsum ? LN_EPI : LN_NONE;
  // the dual instantiations: 16-row tiles, 4 / 8 waves, one load group per wave (every block GEMM of the model at B = 1)
  v.ok = mt == 1 && (v.nw == 4 || v.nw == 8) && (p.K >> 4) <= v.nw * gemm_group_steps(1, v.glu ? 2 : 1, v.nw);
  return v;
}
bool gemm_f32_dual_fusable(const GemmParams& a, const GemmParams& b) {
  const GemmVariant va = gemm_f32_variant(a), vb = gemm_f32_variant(b);
  return va.ok && vb.ok && va.mt == vb.mt && va.nw == vb.nw && va.ln == vb.ln && va.glu == vb.glu;
}
// two independent problems of one instantiation in ONE launch (gemm_f32_dual_fusable must hold; both validated like single launches)
int launch_gemm_f32_dual(const GemmParams& a_in, const GemmParams& b_in, hipStream_t stream) {
  M3_REQUIRE(gemm_f32_dual_fusable(a_in, b_in), "gemm dual: the two problems do not share an instantiation");
  GemmParams q[2] = {a_in, b_in};
  for (GemmParams& p : q) {
    M3_REQUIRE((p.lda & 3) == 0, "gemm: lda=%d must be a multiple of 4", p.lda);
    M3_REQUIRE(!(p.act == ACT_GLU) || (p.N & 1) == 0, "gemm: GLU needs even N");
    M3_REQUIRE(!(p.ln_wsum && p.mask_in) || p.ln_wbeta, "gemm: folded LayerNorm + input mask needs ln_wbeta");
    if (p.ln_wsum) M3_REQUIRE(p.K <= 1024, "gemm: LayerNorm supports rows up to 1024 wide");
    if (p.mask_in || p.mask_out) M3_REQUIRE(p.row_len && p.rows_per_batch > 0, "gemm: mask needs row_len");
    const int Nout = p.act == ACT_GLU ? p.N / 2 : p.N;
    p.n_tiles = cdiv(Nout, 16);
    p.m_tiles = cdiv(p.M, 16);
    p.xcd_swizzle = (p.n_tiles % 8 == 0) ? 1 : 0;
  }
  const GemmVariant v = gemm_f32_variant(a_in);
  const int n0 = (int)align_up((size_t)q[0].n_tiles * q[0].m_tiles, 8);
  dim3 grid(n0 + q[1].n_tiles * q[1].m_tiles);
#define M3_DUAL(GLU_, NW_, LN_)                                                                                             \
  hipLaunchKernelGGL((gemm_f32_dual_kernel<1, GLU_, NW_, LN_>), grid, dim3(64 * NW_), 0, stream, q[0], q[1], n0)
  if (v.glu) {
    if (v.nw == 8) { if (v.ln == LN_EPI) M3_DUAL(true, 8, LN_EPI); else M3_DUAL(true, 8, LN_NONE); }
    else { if (v.ln == LN_EPI) M3_DUAL(true, 4, LN_EPI); else M3_DUAL(true, 4, LN_NONE); }
  } else {
    if (v.nw == 8) { if (v.ln == LN_EPI) M3_DUAL(false, 8, LN_EPI); else M3_DUAL(false, 8, LN_NONE); }
    else { if (v.ln == LN_EPI) M3_DUAL(false, 4, LN_EPI); else M3_DUAL(false, 4, LN_NONE); }
  }
#undef M3_DUAL
  M3_LAUNCH_CHECK();
  return 0;
}

int launch_gemm_f32(const GemmParams& pin, hipStream_t stream) {
  if (pin.w_bf16) return launch_gemm_bf16w(pin, stream);
  GemmParams p = pin;
  M3_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  M3_REQUIRE((p.K & 15) == 0, "gemm: K=%d must be a multiple of 16", p.K);
  M3_REQUIRE((p.lda & 3) == 0, "gemm: lda=%d must be a multiple of 4", p.lda);
  const bool glu = p.act == ACT_GLU;
  M3_REQUIRE(!glu || (p.N & 1) == 0, "gemm: GLU needs even N");
  const bool conv = p.mode == GEMM_A_CONV3X3S2;
  if (p.mode == GEMM_A_CONCAT2)
    M3_REQUIRE((p.K1 & 15) == 0 && p.A2 != nullptr && (p.lda2 & 3) == 0, "gemm: bad concat operands");
  if (conv) M3_REQUIRE((p.conv_C & 15) == 0 && p.K == 9 * p.conv_C, "gemm: conv mode needs K=9*C, C%%16==0");
  const int ln = p.ln_wsum ? LN_EPI : (p.ln_gamma ? LN_PRO : LN_NONE);
  M3_REQUIRE(!(p.ln_wsum && p.ln_gamma), "gemm: folded (ln_wsum) and affine (ln_gamma) LayerNorm are exclusive");
  if (ln != LN_NONE) {
    M3_REQUIRE(p.mode == GEMM_A_PLAIN || (p.mode == GEMM_A_CONCAT2 && p.ln_on_a2),
               "gemm: LayerNorm needs plain A (or the A2 half of a concat)");
    M3_REQUIRE((p.ln_on_a2 ? p.K - p.K1 : p.K) <= 1024, "gemm: LayerNorm supports rows up to 1024 wide");
    M3_REQUIRE(p.K <= 2047, "gemm: LayerNorm variants are built for K < 2048");
  }
  M3_REQUIRE(!p.ln_on_a2 || p.mode == GEMM_A_CONCAT2, "gemm: ln_on_a2 needs concat mode");
  M3_REQUIRE(p.ln_out == nullptr || ln == LN_PRO, "gemm: ln_out needs the affine LayerNorm prologue");
  M3_REQUIRE(!(ln == LN_EPI && p.mask_in) || p.ln_wbeta, "gemm: folded LayerNorm + input mask needs ln_wbeta");
  if (p.mask_in || p.mask_out) M3_REQUIRE(p.row_len && p.rows_per_batch > 0, "gemm: mask needs row_len");
  if (gemm_f32_uses_tiled(p)) return launch_gemm_f32_tiled(p, stream);
  const int Nout = glu ? p.N / 2 : p.N;
  // row tile: 16*MT rows per workgroup; short inputs are cut into 16-row tiles to fill the chip, but more
  // workgroups than fit at once (2 per CU) only serialise: then prefer fatter tiles
  int mt = p.M <= 128 ? 1 : (p.M <= 512 ? 2 : 4);
  while (mt < 4 && 16 * mt < p.M && (long)cdiv(Nout, 16) * cdiv(p.M, 16 * mt) > 512) mt *= 2;
  // narrow outputs over many rows (the router: N = 32 experts, S ~ 2000 rows): fat row tiles would leave most CUs idle
  while (mt > 1 && (long)cdiv(Nout, 16) * cdiv(p.M, 16 * mt) < 256) mt /= 2;
  p.n_tiles = cdiv(Nout, 16);
  p.m_tiles = cdiv(p.M, 16 * mt);
  p.xcd_swizzle = (p.n_tiles % 8 == 0) ? 1 : 0;
  dim3 grid(p.n_tiles * p.m_tiles);
  int nw = p.K >= 2048 ? 16 : (p.K >= 1024 ? 8 : 4);
  if (nw == 16 && (glu || mt == 4)) nw = 8;   // those 16-wave variants would spill registers
  if (nw == 16 && ln != LN_NONE) nw = 8;

#define M3_GEMM_LAUNCH(MT_, GLU_, NW_, CONV_, LN_)                                                              \
  do {                                                                                                          \
    if ((p.K >> 4) <= NW_ * gemm_group_steps(MT_, GLU_ ? 2 : 1, NW_))                                           \
      hipLaunchKernelGGL((gemm_f32_kernel<MT_, GLU_, NW_, CONV_, LN_, 1>), grid, dim3(64 * NW_), 0, stream, p); \
    else                                                                                                        \
      hipLaunchKernelGGL((gemm_f32_kernel<MT_, GLU_, NW_, CONV_, LN_, 2>), grid, dim3(64 * NW_), 0, stream, p); \
  } while (0)
#define M3_GEMM_MT(GLU_, NW_, CONV_, LN_)                                                              \
  do {                                                                                                 \
    if (mt == 1) M3_GEMM_LAUNCH(1, GLU_, NW_, CONV_, LN_);                                             \
    else if (mt == 2) M3_GEMM_LAUNCH(2, GLU_, NW_, CONV_, LN_);                                        \
    else M3_GEMM_LAUNCH(4, GLU_, NW_, CONV_, LN_);                                                     \
  } while (0)
#define M3_GEMM_LN(GLU_, NW_)                                                                          \
  do {                                                                                                 \
    if (ln == LN_EPI) M3_GEMM_MT(GLU_, NW_, false, LN_EPI);                                            \
    else if (ln == LN_PRO) M3_GEMM_MT(GLU_, NW_, false, LN_PRO);                                       \
    else M3_GEMM_MT(GLU_, NW_, false, LN_NONE);                                                        \
  } while (0)
  if (conv) {                       // implicit 3x3 conv: no GLU, no LayerNorm
    M3_REQUIRE(!glu && ln == LN_NONE, "gemm: conv mode supports neither GLU nor LayerNorm");
    if (nw == 16) M3_GEMM_MT(false, 16, true, LN_NONE); else if (nw == 8) M3_GEMM_MT(false, 8, true, LN_NONE);
    else M3_GEMM_MT(false, 4, true, LN_NONE);
  } else if (nw == 16) {            // K >= 2048: plain only
    M3_GEMM_MT(false, 16, false, LN_NONE);
  } else if (glu) {
    if (nw == 8) M3_GEMM_LN(true, 8); else M3_GEMM_LN(true, 4);
  } else {
    if (nw == 8) M3_GEMM_LN(false, 8); else M3_GEMM_LN(false, 4);
  }
#undef M3_GEMM_LN
#undef M3_GEMM_MT
#undef M3_GEMM_LAUNCH
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3

#ifdef M3_GEMM_DIAG
extern "C" int m3_debug_gemm_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(m3::g_gemm_dbg), bytes < sizeof(m3::g_gemm_dbg) ? bytes : sizeof(m3::g_gemm_dbg));
}
#endif
