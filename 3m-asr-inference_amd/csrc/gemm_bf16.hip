// Skinny GEMM with bf16 weights:  Y[M,N] = epilogue( A[M,K] . W[N,K]^T )  on v_mfma_f32_16x16x32_bf16.
//
// The low-precision form of gemm.hip (reference: HelperConfig.plugin_data_type = 1 / builder.py --fp16,
// TRTAPI++/python/trt_helper/builder_helper.py:47-57,109-123 -- wired in the reference but never finished).
// CDNA4 has no fp16 advantage over bf16 and bf16 keeps the fp32 exponent range, so bf16 is the 16-bit type here.
// Storage / arithmetic split:
//   * weights live in HBM as bf16 (half the bytes of the weight-streaming regime that bounds B=1),
//   * activations stay fp32 in HBM (residual stream, LayerNorm statistics, softmax, biases: all fp32); each
//     lane rounds its 8-float A chunk to bf16 (v_cvt_pk_bf16_f32, round-to-nearest-even) right before the MFMA,
//   * accumulation is fp32, the epilogue (folded LayerNorm, bias, activation, mask, residual) is fp32.
// Same work decomposition as gemm.hip (16 output columns x 16*MT rows per workgroup, K split over NW waves in
// 32-deep steps, fixed-order LDS reduction), so results are bitwise reproducible run to run.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace m3 {

int launch_gemm_bf16w_tiled(const GemmParams& p, hipStream_t stream);   // gemm_bf16_tiled.hip
bool gemm_bf16w_tiled_supports(const GemmParams& p);
// below this many rows the 16-column K-split kernel fills the chip better (M3_TILED_MIN_ROWS overrides, for tuning)
static int tiled_min_rows() {
  static const int v = [] {
    const char* e = getenv("M3_TILED_MIN_ROWS");
    return e ? atoi(e) : 384;
  }();
  return v;
}

constexpr int gemm16_group_steps(int MT, int NW) { return NW == 16 ? 2 : (MT == 4 ? 2 : 4); }

// bf16 activations x bf16 weights from this many rows on: the LDS-DMA fed kernel (gemm_bf16_dma.hip).  Measured against the
// register-staged kernel (tools/bench_gemm_bf16.py --a16, profiles/r03_gemm_dma_vs_staged.txt): +12-16 % at 16 384 rows, +0-16 %
// at 4 480, SLOWER at 1 984 rows (128 x 128 tiles leave most CUs idle there and a CU keeps only ~16 KB of LDS-DMA in
// flight: a k-step is a full ~1.3 us round trip).  M3_DMA_MIN_ROWS overrides (read once).
static int dma_min_rows() {
  static const int v = [] {
    const char* e = getenv("M3_DMA_MIN_ROWS");
    return e ? atoi(e) : 4096;
  }();
  return v;
}
bool gemm_bf16w_uses_dma(const GemmParams& p) { return p.M >= dma_min_rows() && gemm_bf16_dma_supports(p); }

// long batches: LDS-tiled kernel, when there are enough 64 x 64 tiles to occupy the chip
bool gemm_bf16w_uses_tiled(const GemmParams& p) {
  const bool glu = p.act == ACT_GLU;
  return p.M >= tiled_min_rows() && (long)cdiv(p.M, 64) * cdiv(glu ? p.N / 2 : p.N, 64) >= 160 && gemm_bf16w_tiled_supports(p) &&
         p.mode != GEMM_A_CONCAT2 && p.ln_gamma == nullptr;
}

template <int MT, bool GLU, int NW, bool CONV, bool LN, int NBUF>
__global__ __launch_bounds__(64 * NW) void gemm_bf16w_kernel(const GemmParams p) {
  constexpr int NT = GLU ? 2 : 1;
  constexpr int G = gemm16_group_steps(MT, NW);
  __shared__ float red[NW][MT * NT][256];
  __shared__ float rsum[LN ? NW : 1][16 * MT][2];

  // one batch of kernel-argument loads (gemm.hip: the lazily loaded parameter block cost ~8 dependent s_load rounds)
  asm volatile("" ::"s"(p.A), "s"(p.lda), "s"(p.W), "s"(p.bias), "s"(p.Y), "s"(p.ldy), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.mode),
               "s"(p.n_tiles), "s"(p.m_tiles), "s"(p.xcd_swizzle), "s"(p.m_dev), "s"(p.resid), "s"(p.ldr), "s"(p.act), "s"(p.alpha),
               "s"(p.mask_in), "s"(p.mask_out), "s"(p.row_len), "s"(p.rows_per_batch));
  if (LN) asm volatile("" ::"s"(p.ln_wsum), "s"(p.ln_wbeta), "s"(p.ln_eps));
  if (CONV) asm volatile("" ::"s"(p.conv_T1), "s"(p.conv_F1), "s"(p.conv_T2), "s"(p.conv_F2), "s"(p.conv_C));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int Nout = GLU ? (p.N >> 1) : p.N;

  int n_tile, m_tile;
  {
    const int id = blockIdx.x;
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

  const float* arow[MT];
  bool a_zero[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = min(m0 + 16 * mt + col, p.M - 1);
    a_zero[mt] = false;
    if (CONV) {
      const int f2 = m % p.conv_F2;
      const int t2 = (m / p.conv_F2) % p.conv_T2;
      const int b = m / (p.conv_F2 * p.conv_T2);
      arow[mt] = p.A + ((size_t)(b * p.conv_T1 + 2 * t2) * p.conv_F1 + 2 * f2) * p.conv_C + 8 * kq;
    } else {
      arow[mt] = p.A + (size_t)m * p.lda + 8 * kq;
    }
    if (p.mask_in) a_zero[mt] = (m % p.rows_per_batch) >= p.row_len[m / p.rows_per_batch];
  }
  const bf16_t* W = reinterpret_cast<const bf16_t*>(p.W);
  const bf16_t* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wrow[t] = W + (size_t)min(n0 + t * Nout + col, p.N - 1) * p.K + 8 * kq;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[MT], s2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) s1[mt] = s2[mt] = 0.f;

  const int nsteps = p.K >> 5;
  auto a_offset = [&](int k) -> int {   // wave-uniform k (multiple of 32) -> element offset in the A row
    if (CONV) {
      const int seg = k / p.conv_C, c = k - seg * p.conv_C;
      const int kh = seg / 3, kw = seg - kh * 3;
      return (kh * p.conv_F1 + kw) * p.conv_C + c;
    }
    return k;
  };

  bf16x8 wbuf[NBUF][G][NT];
  f32x4 abuf[NBUF][G][MT][2];
  auto load_group = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int s = min(wave + NW * (G * g + i), nsteps - 1);   // clamped: no branch around the loads
      const int k = s << 5;
#pragma unroll
      for (int t = 0; t < NT; ++t) wbuf[buf][i][t] = ldg8h_w(wrow[t] + k);
      const int off = a_offset(k);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        abuf[buf][i][mt][0] = ldg4(arow[mt] + off);
        abuf[buf][i][mt][1] = ldg4(arow[mt] + off + 4);
      }
    }
  };

  const int ep_mt = wave;
  const bool is_ep = wave < MT;
  const int ep_n = n0 + col;
  float bias0 = 0.f, bias1 = 0.f, wsum0 = 0.f, wsum1 = 0.f, wbeta0 = 0.f, wbeta1 = 0.f, res[4] = {0.f, 0.f, 0.f, 0.f};
  if (is_ep && ep_n < Nout) {
    if (p.bias) {
      bias0 = p.bias[ep_n];
      if (GLU) bias1 = p.bias[ep_n + Nout];
    }
    if (LN) {
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

  const int ngroups = (nsteps + NW * G - 1) / (NW * G);
  load_group(0, 0);

  auto compute_group = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int s = wave + NW * (G * g + i);
      if (s < nsteps) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 lo = abuf[buf][i][mt][0], hi = abuf[buf][i][mt][1];
          if (LN) {   // statistics from the fp32 values (before rounding)
            s1[mt] += ((lo[0] + lo[1]) + (lo[2] + lo[3])) + ((hi[0] + hi[1]) + (hi[2] + hi[3]));
            s2[mt] += ((lo[0] * lo[0] + lo[1] * lo[1]) + (lo[2] * lo[2] + lo[3] * lo[3])) +
                      ((hi[0] * hi[0] + hi[1] * hi[1]) + (hi[2] * hi[2] + hi[3] * hi[3]));
          }
          if (a_zero[mt]) lo = hi = f32x4{0.f, 0.f, 0.f, 0.f};
          const bf16x8 a = cvt8(lo, hi);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[mt][t] = mfma16h(a, wbuf[buf][i][t], acc[mt][t]);
        }
      }
    }
  };

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

  // ---- cross-wave K reduction through LDS (fixed order), then the fp32 epilogue ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][mt * NT + t][r * 64 + lane] = acc[mt][t][r];
  if (LN) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float a1 = s1[mt], a2 = s2[mt];
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

  if (is_ep) {
    const int mt = ep_mt;
    f32x4 v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sum = 0.f;
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
        bool pad = false;
        if (p.mask_in || p.mask_out) pad = (m % p.rows_per_batch) >= p.row_len[m / p.rows_per_batch];
        if (LN) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            t1 += rsum[w][row][0];
            t2 += rsum[w][row][1];
          }
          const float mean = t1 / (float)p.K;
          const float var = fmaxf(t2 / (float)p.K - mean * mean, 0.f);
          const float rstd = rsqrtf(var + p.ln_eps);
          if (p.mask_in && pad) {
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
}

int launch_gemm_bf16w(const GemmParams& pin, hipStream_t stream) {
  GemmParams p = pin;
  M3_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm_bf16w: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  M3_REQUIRE((p.K & 31) == 0, "gemm_bf16w: K=%d must be a multiple of 32", p.K);
  M3_REQUIRE((p.lda & 3) == 0, "gemm_bf16w: lda=%d must be a multiple of 4", p.lda);
  const bool glu = p.act == ACT_GLU;
  M3_REQUIRE(!glu || (p.N & 1) == 0, "gemm_bf16w: GLU needs even N");
  const bool conv = p.mode == GEMM_A_CONV3X3S2;
  M3_REQUIRE(p.mode != GEMM_A_CONCAT2, "gemm_bf16w: concat operands are fp32-only (the router stays fp32)");
  M3_REQUIRE(p.ln_gamma == nullptr && p.ln_out == nullptr,
             "gemm_bf16w: only the folded LayerNorm (ln_wsum) is available with bf16 weights");
  if (conv) M3_REQUIRE((p.conv_C & 31) == 0 && p.K == 9 * p.conv_C, "gemm_bf16w: conv mode needs K=9*C, C%%32==0");
  const bool ln = p.ln_wsum != nullptr;
  if (ln) M3_REQUIRE(p.mode == GEMM_A_PLAIN && p.K <= 2047, "gemm_bf16w: LayerNorm needs plain A with K < 2048");
  M3_REQUIRE(!(ln && p.mask_in) || p.ln_wbeta, "gemm_bf16w: folded LayerNorm + input mask needs ln_wbeta");
  if (p.mask_in || p.mask_out) M3_REQUIRE(p.row_len && p.rows_per_batch > 0, "gemm_bf16w: mask needs row_len");
  if (gemm_bf16w_uses_dma(p)) return launch_gemm_bf16_dma(p, stream);
  // only the LDS-DMA kernel writes / reads the row-statistic partials: a consumer of y_copy_stats would read stale numbers
  M3_REQUIRE(p.Yb_stats == nullptr && p.ln_stats == nullptr,
             "gemm_bf16w: y_copy_stats / ln_stats need the LDS-DMA kernel (M >= %d rows, bf16 A); this problem (M=%d) runs on another one",
             dma_min_rows(), p.M);
  if (gemm_bf16w_uses_tiled(p)) return launch_gemm_bf16w_tiled(p, stream);
  M3_REQUIRE(!p.a_bf16 && !p.y_bf16 && p.Yb == nullptr, "gemm_bf16w: bf16 activations are a feature of the tiled kernel");
  const int Nout = glu ? p.N / 2 : p.N;
  int mt = p.M <= 128 ? 1 : (p.M <= 512 ? 2 : 4);
  while (mt < 4 && 16 * mt < p.M && (long)cdiv(Nout, 16) * cdiv(p.M, 16 * mt) > 512) mt *= 2;
  p.n_tiles = cdiv(Nout, 16);
  p.m_tiles = cdiv(p.M, 16 * mt);
  p.xcd_swizzle = (p.n_tiles % 8 == 0) ? 1 : 0;
  dim3 grid(p.n_tiles * p.m_tiles);
  int nw = p.K >= 2048 ? 16 : (p.K >= 1024 ? 8 : 4);
  if (nw == 16 && (glu || mt == 4 || ln)) nw = 8;

#define M3_GEMM_LAUNCH(MT_, GLU_, NW_, CONV_, LN_)                                                                 \
  do {                                                                                                             \
    if ((p.K >> 5) <= NW_ * gemm16_group_steps(MT_, NW_))                                                          \
      hipLaunchKernelGGL((gemm_bf16w_kernel<MT_, GLU_, NW_, CONV_, LN_, 1>), grid, dim3(64 * NW_), 0, stream, p);  \
    else                                                                                                           \
      hipLaunchKernelGGL((gemm_bf16w_kernel<MT_, GLU_, NW_, CONV_, LN_, 2>), grid, dim3(64 * NW_), 0, stream, p);  \
  } while (0)
#define M3_GEMM_MT(GLU_, NW_, CONV_, LN_)                          \
  do {                                                             \
    if (mt == 1) M3_GEMM_LAUNCH(1, GLU_, NW_, CONV_, LN_);         \
    else if (mt == 2) M3_GEMM_LAUNCH(2, GLU_, NW_, CONV_, LN_);    \
    else M3_GEMM_LAUNCH(4, GLU_, NW_, CONV_, LN_);                 \
  } while (0)
#define M3_GEMM_LN(GLU_, NW_)                                      \
  do {                                                             \
    if (ln) M3_GEMM_MT(GLU_, NW_, false, true);                    \
    else M3_GEMM_MT(GLU_, NW_, false, false);                      \
  } while (0)
  if (conv) {
    M3_REQUIRE(!glu && !ln, "gemm_bf16w: conv mode supports neither GLU nor LayerNorm");
    if (nw == 16) M3_GEMM_MT(false, 16, true, false); else if (nw == 8) M3_GEMM_MT(false, 8, true, false);
    else M3_GEMM_MT(false, 4, true, false);
  } else if (nw == 16) {
    M3_GEMM_MT(false, 16, false, false);
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
