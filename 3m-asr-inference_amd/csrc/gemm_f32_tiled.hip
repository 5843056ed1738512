// LDS-tiled fp32 GEMM for long batches:  Y[M,N] = epilogue( A[M,K] . W[N,K]^T ),  M >= ~400 rows, exact fp32 MFMA.
//
// The fp32 twin of gemm_bf16_tiled.hip (same tile / pipeline / epilogue design, v_mfma_f32_16x16x4_f32 instead of the bf16
// MFMA, LDS rows of fp32): for the shapes the reference builds its engine for -- builder.py:58-64 profiles feat at
// (1..6) x (1..6100) frames with opt 4 x 500, i.e. S = 496 .. 9000 token rows -- the 16-column K-split kernel of gemm.hip
// re-reads every A row N/16 times (conv2 at 4 x 500: 459 us for 21 GFLOP).  Tiles 128 x 128 x 32 when >= 200 of them exist,
// else 64 x 64 x 64; k order inside a 16-deep sub-step: lane (col, kq) feeds its float4 k = 4kq..4kq+3 into four MFMAs,
// the same permutation on both operands (exact).  All epilogues of gemm.hip except the affine LayerNorm prologue / concat
// (router GEMM: stays on the K-split kernel).
#include "common.h"
#include "kernels.h"

namespace m3 {

namespace {
constexpr int f32d_lds_bytes(int BM, int BN, int BK) {
  const int ring = 2 * (BM + BN) * (BK + 4) * 4, image = BM * (BN + 4) * 4;
  return ring > image ? ring : image;
}
}  // namespace

template <int TBM, int TBN, int TBK, bool GLU, bool CONV, bool LN>
__global__ __launch_bounds__(256, 2) void gemm_f32_tiled_kernel(const GemmParams p) {
  constexpr int T_LD = TBK + 4;                     // floats per LDS row (144 / 272 B: conflict-free 16-B reads)
  constexpr int C_LD = TBN + 4;                     // fp32 elements per row of the epilogue image
  constexpr int MT = TBM / 32, NT = TBN / 32;       // 16x16 MFMA tiles per wave (wave tile = TBM/2 x TBN/2)
  constexpr int CA = TBK / 4, RA = 256 / CA, JA = TBM / RA, JB = TBN / RA;   // float4 chunks per row, rows per pass, passes (A, W)
  extern __shared__ __attribute__((aligned(16))) unsigned char tiled_lds_f32d[];
  float* As = reinterpret_cast<float*>(tiled_lds_f32d);            // [2][TBM][T_LD]
  float* Bs = As + 2 * TBM * T_LD;                                 // [2][TBN][T_LD]
  float* Cs = reinterpret_cast<float*>(tiled_lds_f32d);                  // [TBM][C_LD] after the k loop
  __shared__ float stats[TBM][2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int Nout = GLU ? (p.N >> 1) : p.N;
  constexpr int OUTW = GLU ? TBN / 2 : TBN;         // output columns per workgroup
  // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs (id % 8), each with its own L2.  All column
  // tiles of one row tile run on the SAME XCD, back to back, so an A tile (fp32, the dominant traffic) is fetched from
  // HBM / Infinity Cache once instead of once per XCD; only W (small) is replicated over the L2s.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_tile = slot % p.n_tiles, m_tile = (slot / p.n_tiles) * 8 + xcd;
  if (m_tile >= p.m_tiles) return;                  // padding of the last group of 8 row tiles (whole workgroup exits)
  const int m0 = m_tile * TBM, m_end = p.M;
  if (p.m_dev != nullptr && m0 > *p.m_dev) return;   // packed ragged batch: no live row in this tile
  if (CONV && p.conv_len != nullptr) {
    const int per_utt = p.conv_T2 * p.conv_F2;
    const int b0 = m0 / per_utt, b1 = min(m0 + TBM - 1, p.M - 1) / per_utt;
    if (b0 == b1 && (m0 - b0 * per_utt) / p.conv_F2 >= p.conv_len[b0]) return;   // every row is a padded frame
  }
  const int n0 = n_tile * OUTW;

  // first W-tile row of accumulator tile nt of this wave; GLU: tile rows [0, TBN/2) value, [TBN/2, TBN) gate columns
  auto btile = [&](int nt) {
    return GLU ? (nt / (NT / 2)) * (TBN / 2) + wn * (TBN / 4) + 16 * (nt % (NT / 2)) : wn * (TBN / 2) + 16 * nt;
  };

  // ---- global -> register staging, fully coalesced ----
  // A: thread t brings float4 chunk (t % CA) of rows (t / CA) + RA j  (a wave instruction covers whole rows of the tile)
  const int ac = tid % CA, ar0 = tid / CA;
  const float* aptr[JA];
  bool a_zero[JA];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int m = min(m0 + ar0 + RA * j, m_end - 1);
    if (CONV) {
      const int f2 = m % p.conv_F2;
      const int t2 = (m / p.conv_F2) % p.conv_T2;
      const int b = m / (p.conv_F2 * p.conv_T2);
      aptr[j] = p.A + ((size_t)(b * p.conv_T1 + 2 * t2) * p.conv_F1 + 2 * f2) * p.conv_C + 4 * ac;
    } else {
      aptr[j] = p.A + (size_t)m * p.lda + 4 * ac;
    }
    a_zero[j] = p.mask_in ? ((m % p.rows_per_batch) >= p.row_len[m / p.rows_per_batch]) : false;
  }
  // W: same map (float4 chunk ac of tile rows ar0 + RA j)
  const float* bptr[JB];
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int tr = ar0 + RA * j;
    const int n = GLU ? (tr / (TBN / 2)) * Nout + min(n0 + (tr % (TBN / 2)), Nout - 1) : min(n0 + tr, p.N - 1);
    bptr[j] = p.W + (size_t)n * p.K + 4 * ac;
  }
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
  float s1[JA], s2[JA];
#pragma unroll
  for (int j = 0; j < JA; ++j) s1[j] = s2[j] = 0.f;

  const int nsteps = p.K / TBK;
  f32x4 areg[JA], breg[JB];
  auto load_tiles = [&](int s) {
    const int ko = a_offset(s * TBK);
#pragma unroll
    for (int j = 0; j < JA; ++j) areg[j] = ldg4(aptr[j] + ko);
#pragma unroll
    for (int j = 0; j < JB; ++j) breg[j] = ldg4(bptr[j] + s * TBK);
  };
  // `on` = 0 for the redundant store after the last k-step (the store stays unconditional, see the header)
  auto store_tiles = [&](int buf, float on) {
    float* a_dst = As + buf * (TBM * T_LD) + ar0 * T_LD + 4 * ac;
    float* b_dst = Bs + buf * (TBN * T_LD) + ar0 * T_LD + 4 * ac;
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const f32x4 v = areg[j];
      if (LN) {
        s1[j] += on * ((v[0] + v[1]) + (v[2] + v[3]));
        s2[j] += on * ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
      }
      *reinterpret_cast<f32x4*>(a_dst + RA * j * T_LD) = a_zero[j] ? f32x4{0.f, 0.f, 0.f, 0.f} : v;
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) *reinterpret_cast<f32x4*>(b_dst + RA * j * T_LD) = breg[j];
  };

  load_tiles(0);
  store_tiles(0, 1.f);
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    load_tiles(min(s + 1, nsteps - 1));             // clamped: the loads are never behind a branch
    __builtin_amdgcn_sched_barrier(0);              // all staging loads are in flight before the MFMA phase starts
    const float* a_lds = As + (s & 1) * (TBM * T_LD) + ((TBM / 2) * wm + col) * T_LD + 4 * kq;
    const float* b_lds = Bs + (s & 1) * (TBN * T_LD) + col * T_LD + 4 * kq;
#pragma unroll
    for (int ks = 0; ks < TBK / 16; ++ks) {
      f32x4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(b_lds + btile(nt) * T_LD + 16 * ks);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(a_lds + 16 * mt * T_LD + 16 * ks);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[mt][nt] = mfma16(a[j], b[nt][j], acc[mt][nt]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    store_tiles((s + 1) & 1, s + 1 < nsteps ? 1.f : 0.f);
    __syncthreads();
  }

  if (LN) {
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      // the CA (8 / 16 / 32) consecutive lanes that staged row ar0 + RA j: DPP butterfly of the right width
      static_assert(CA == 8 || CA == 16 || CA == 32, "row staged by 8, 16 or 32 lanes");
      float t1 = s1[j], t2 = s2[j];
      t1 += dpp_mov<0xB1>(t1); t2 += dpp_mov<0xB1>(t2);       // xor 1
      t1 += dpp_mov<0x4E>(t1); t2 += dpp_mov<0x4E>(t2);       // xor 2
      t1 += dpp_mov<0x141>(t1); t2 += dpp_mov<0x141>(t2);     // row_half_mirror: 8 lanes
      if (CA >= 16) {
        t1 += dpp_mov<0x140>(t1); t2 += dpp_mov<0x140>(t2);   // row_mirror: 16 lanes
      }
      if (CA == 32) {
        t1 += __shfl_xor(t1, 16, 64);
        t2 += __shfl_xor(t2, 16, 64);
      }
      if (ac == 0) {
        stats[ar0 + RA * j][0] = t1;
        stats[ar0 + RA * j][1] = t2;
      }
    }
  }
  // ---- accumulators -> LDS image (the ring is dead: every wave passed the loop's last barrier) ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[((TBM / 2) * wm + 16 * mt + 4 * kq + r) * C_LD + btile(nt) + col] = acc[mt][nt][r];
  __syncthreads();

  // ---- fp32 epilogue, row-wise: a lane owns 4 consecutive output columns, a wave sweeps rows ----
  constexpr int LPR = OUTW / 4;                     // lanes per output row (32, GLU 16)
  constexpr int RPI = 64 / LPR;                     // rows per wave iteration (2, GLU 4)
  const int c4 = 4 * (lane % LPR);                  // first of this lane's 4 columns inside the tile
  const int n = n0 + c4;
  float bias0[4], bias1[4], wsum0[4], wsum1[4], wbeta0[4], wbeta1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ne = min(n + e, Nout - 1);
    bias0[e] = p.bias ? p.bias[ne] : 0.f;
    bias1[e] = (GLU && p.bias) ? p.bias[ne + Nout] : 0.f;
    wsum0[e] = LN ? p.ln_wsum[ne] : 0.f;
    wsum1[e] = (LN && GLU) ? p.ln_wsum[ne + Nout] : 0.f;
    wbeta0[e] = (LN && p.mask_in) ? p.ln_wbeta[ne] : 0.f;
    wbeta1[e] = (LN && GLU && p.mask_in) ? p.ln_wbeta[ne + Nout] : 0.f;
  }
  const bool vec_ok = ((p.ldy & 3) == 0) && (!p.resid || (p.ldr & 3) == 0) && (n + 3 < Nout);
  // Every load of the sweep is issued BEFORE it (residual rows into registers -- the accumulators are dead --, the row
  // masks resolved here): a load inside the sweep makes hipcc wait vmcnt(0) in every iteration, and vmcnt counts the
  // previous iteration's stores too, so each iteration paid a full store round trip (in-kernel stamps of the LDS-DMA
  // kernel, tools/diag_gemm_dma.py: the sweep was the longest phase of the work-group).
  constexpr int IT = TBM / (4 * RPI);
  f32x4 res_all[IT];
  bool pad_all[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int m = min(m0 + (4 * it + wave) * RPI + lane / LPR, m_end - 1);      // clamped, never branched around
    res_all[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.resid) {
      if (vec_ok) {
        res_all[it] = ldg4(p.resid + (size_t)m * p.ldr + n);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) res_all[it][e] = p.resid[(size_t)m * p.ldr + min(n + e, Nout - 1)];
      }
    }
    pad_all[it] = (p.mask_in || p.mask_out) ? ((m % p.rows_per_batch) >= p.row_len[m / p.rows_per_batch]) : false;
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int row = (4 * it + wave) * RPI + lane / LPR;
    const int m = m0 + row;
    if (m >= m_end || n >= Nout) continue;
    const bool pad = pad_all[it];
    float mean = 0.f, rstd = 1.f;
    if (LN) {
      mean = stats[row][0] / (float)p.K;
      const float var = fmaxf(stats[row][1] / (float)p.K - mean * mean, 0.f);
      rstd = rsqrtf(var + p.ln_eps);
    }
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(Cs + row * C_LD + c4);
    f32x4 v1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (GLU) v1 = *reinterpret_cast<const f32x4*>(Cs + row * C_LD + TBN / 2 + c4);
    const f32x4 res = res_all[it];
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float y0 = v0[e], y1 = v1[e];
      if (LN) {
        if (p.mask_in && pad) {
          y0 = -wbeta0[e];
          y1 = -wbeta1[e];
        } else {
          y0 = rstd * (y0 - mean * wsum0[e]);
          y1 = rstd * (y1 - mean * wsum1[e]);
        }
      }
      float t = y0 + bias0[e];
      if (GLU) t = t * sigmoidf(y1 + bias1[e]);
      if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
      if (p.act == ACT_SILU) t = silu(t);
      if (p.mask_out && pad) t = 0.f;
      t *= p.alpha;
      if (p.resid) t += res[e];
      y[e] = t;
    }
    if (vec_ok) {
      stg4(p.Y + (size_t)m * p.ldy + n, y);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < Nout) p.Y[(size_t)m * p.ldy + n + e] = y[e];
    }
  }
}

// > 64 KB of dynamic LDS must be opted into once per kernel (not a stream operation: outside graph capture)
#define M3_F32D_FOR_ALL(X) \
  X(false, true, false) X(true, false, true) X(true, false, false) X(false, false, true) X(false, false, false)
int init_gemm_f32_tiled_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
#define M3_F32D_ATTR(G_, C_, L_)                                                                              \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f32_tiled_kernel<128, 128, 32, G_, C_, L_>,              \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, f32d_lds_bytes(128, 128, 32))); \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f32_tiled_kernel<64, 64, 64, G_, C_, L_>,                \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, f32d_lds_bytes(64, 64, 64)));
  M3_F32D_FOR_ALL(M3_F32D_ATTR)
#undef M3_F32D_ATTR
  once.mark();
  return 0;
}

bool gemm_f32_tiled_supports(const GemmParams& p) {
  return (p.K & 63) == 0 && (p.lda & 3) == 0 && p.mode != GEMM_A_CONCAT2 && p.ln_gamma == nullptr &&
         (p.mode != GEMM_A_CONV3X3S2 || (p.conv_C & 63) == 0);
}

// caller (launch_gemm_f32) has validated the operands; returns 0 / error
int launch_gemm_f32_tiled(const GemmParams& pin, hipStream_t stream) {
  GemmParams p = pin;
  if (int rc = init_gemm_f32_tiled_kernels()) return rc;
  const bool glu = p.act == ACT_GLU;
  const bool conv = p.mode == GEMM_A_CONV3X3S2;
  const bool ln = p.ln_wsum != nullptr;
  const int Nout = glu ? p.N / 2 : p.N;
  M3_REQUIRE(!(conv && (glu || ln)), "gemm: conv mode supports neither GLU nor LayerNorm");
  const bool big = (long)cdiv(p.M, 128) * cdiv(p.N, 128) >= 200;
  const int bm = big ? 128 : 64, bn = big ? 128 : 64;
  p.m_tiles = cdiv(p.M, bm);
  p.n_tiles = glu ? cdiv(Nout, bn / 2) : cdiv(p.N, bn);
  dim3 grid(cdiv(p.m_tiles, 8) * 8 * p.n_tiles);   // row tiles in groups of 8 (one per XCD)
#define M3_F32D_LAUNCH(G_, C_, L_)                                                                              \
  if (glu == G_ && conv == C_ && ln == L_) {                                                                    \
    if (big)                                                                                                    \
      hipLaunchKernelGGL((gemm_f32_tiled_kernel<128, 128, 32, G_, C_, L_>), grid, dim3(256),                    \
                         f32d_lds_bytes(128, 128, 32), stream, p);                                              \
    else                                                                                                        \
      hipLaunchKernelGGL((gemm_f32_tiled_kernel<64, 64, 64, G_, C_, L_>), grid, dim3(256),                      \
                         f32d_lds_bytes(64, 64, 64), stream, p);                                                \
  }
  M3_F32D_FOR_ALL(M3_F32D_LAUNCH)
#undef M3_F32D_LAUNCH
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
