// LDS-tiled bf16 GEMM for long batches:  Y[M,N] = epilogue( A[M,K] . W[N,K]^T ),  M >= ~400 rows.
//
// Same contract as gemm_bf16.hip (fp32 A rounded to bf16 at the MFMA input, bf16 weights, fp32 accumulate and
// epilogue; folded LayerNorm, mask, bias, ReLU / SiLU / GLU, scale, residual; implicit 3x3-stride-2 conv), for the
// regime the K-split kernel is wrong for: at S = B*T' ~ 2000 tokens (BASELINE.json configs[2], batch 16 x <=500
// frames) a 16-column workgroup re-reads every A row N/16 times and saturates L2.  Here:
//   * one workgroup = 128 x 128 output tile (GLU: 128 rows x 64 value + 64 gate columns), 4 waves in a 2 x 2 grid,
//     64 x 64 per wave = 4 x 4 MFMA tiles of v_mfma_f32_16x16x32_bf16 (64 accumulator registers);
//   * k-steps of 64: A (fp32) and W (bf16) tiles are fetched with fully coalesced 16-byte-per-lane loads (a wave
//     instruction covers whole 128-B lines), A is rounded to bf16 on the way, both go to a 2-stage LDS ring with
//     144-byte rows (conflict-free ds_read_b128 of the MFMA fragments); one barrier per k-step;
//   * a k-step's loads are issued before the previous step's MFMAs (unconditionally: a load whose use sits under a
//     branch is sunk to the branch by the compiler and its latency is exposed);
//   * 74 KB LDS and < 256 VGPRs: two workgroups share a CU, one computes while the other waits on memory
//     (a CU with 4 resident waves pulls only ~30 GB/s from L2 at ~2 us loaded latency);
//   * the epilogue goes through LDS: accumulators -> 128 x 128 fp32 image -> row-wise coalesced float4 stores in a
//     short loop (the fully unrolled per-register epilogue was ~30 KB of code that ran once: instruction-cache misses
//     cost more than the GEMM);
//   * LayerNorm statistics (sum, sum of squares of the fp32 rows) are accumulated by the threads that stage A.
#include "common.h"
#include "kernels.h"

namespace m3 {

constexpr int kGrpRun = 4;   // row tiles of a grouped GEMM that run back to back on one XCD

namespace {
// two tile shapes: 128 x 128 x 64 when the problem has >= ~200 such tiles (MFMA-heavy: conv2), else 64 x 64 x 128 --
// 4x the workgroups and half the k-steps, because a small GEMM is a chain of k-steps of ~1 us memory latency each
constexpr int tiled_lds_bytes(int BM, int BN, int BK) {
  const int ring = 2 * (BM + BN) * (BK + 8) * 2, image = BM * (BN + 4) * 4;
  return ring > image ? ring : image;
}
}

// GRP = 0: dense GEMM.  GRP = 1 / 2: the two GEMMs of the grouped expert FFN for long batches (moe_expert_tiled):
//   row tiles are cut per expert from acc_hist (p.grp_acc), W / bias are that expert's;
//   1: A rows are gathered through pos (fused local_scatter), fp32 -> H = SiLU(. + b1) written as bf16 in sorted order;
//   2: A = H (bf16, no conversion while staging), W2 in the plan's slice-major layout or [D][F], fp32 rows out.
// W8 (grouped forms only): W holds fp8 e4m3 with a per-output-row scale (p.w_scale); the staging threads dequantise it to
// bf16 on the way into LDS (exact) and the scale is applied to the accumulator in the epilogue.
// A16: the A operand is already bf16 in memory (activation copies written by the producing kernels in the 16-bit modes:
// half the A traffic, which is what bounds these GEMMs); LayerNorm statistics are then taken from the bf16 values, i.e.
// from exactly the numbers the MFMA multiplies.  GRP = 2 implies A16.
template <int TBM, int TBN, int TBK, bool GLU, bool CONV, bool LN, int GRP, bool W8 = false, bool A16IN = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16w_tiled_kernel(const GemmParams p) {
  static_assert(GRP == 0 || (!GLU && !CONV && !LN), "grouped form is a plain GEMM");
  static_assert(!W8 || GRP != 0, "fp8 weights: grouped expert GEMMs only");
  constexpr int T_LD = TBK + 8;                     // bf16 elements per LDS row (144 / 272 B: conflict-free 16-B reads)
  constexpr int C_LD = TBN + 4;                     // fp32 elements per row of the epilogue image
  constexpr int MT = TBM / 32, NT = TBN / 32;       // 16x16 MFMA tiles per wave (wave tile = TBM/2 x TBN/2)
  constexpr bool A16 = GRP == 2 || A16IN;           // A operand already bf16 (16-B chunks of 8 elements)
#ifdef M3_NO_EARLY_EPI                              // (A/B builds only)
  constexpr bool EARLY_EPI = false;
#else
  constexpr bool EARLY_EPI = TBM <= 64 && GRP == 0; // epilogue operands requested before the k loop (see below)
#endif
  constexpr int CA = A16 ? TBK / 8 : TBK / 4, RA = 256 / CA, JA = TBM / RA;   // A staging: chunks per row, rows per pass, passes
  constexpr int WCE = W8 ? 16 : 8, WSZ = W8 ? 1 : 2;          // W elements per 16-B chunk, bytes per element
  constexpr int CB = TBK / WCE, RB = 256 / CB, JB = TBN / RB;  // W staging: 16-B chunks per row, rows per pass, passes
  extern __shared__ __attribute__((aligned(16))) unsigned char tiled_lds[];
  bf16_t* As = reinterpret_cast<bf16_t*>(tiled_lds);                // [2][TBM][T_LD]
  bf16_t* Bs = As + 2 * TBM * T_LD;                                 // [2][TBN][T_LD]
  float* Cs = reinterpret_cast<float*>(tiled_lds);                  // [TBM][C_LD] after the k loop
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
  const int n_tile = slot % p.n_tiles;
  // grouped: kGrpRun consecutive row tiles (~ one expert's) stay on one XCD, so the expert's W tiles are fetched once too
  const int q_ = slot / p.n_tiles;
  const int m_tile = GRP ? ((q_ / kGrpRun) * 8 + xcd) * kGrpRun + (q_ % kGrpRun) : q_ * 8 + xcd;
  int m0 = m_tile * TBM, m_end = p.M, expert = 0;
  if (GRP) {   // m_tile counts the row tiles of all experts in expert order: find its expert
    if (p.grp_E <= 64) {
      // one load per lane + a wave prefix sum instead of E dependent scalar loads (those were ~5 us of a ~15 us kernel)
      const int lo = lane < p.grp_E ? p.grp_acc[lane] : 0, hi = lane < p.grp_E ? p.grp_acc[lane + 1] : 0;
      const int nt_e = (hi - lo + TBM - 1) / TBM;
      int incl = nt_e;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
      }
      const unsigned long long owner = __ballot(m_tile < incl);   // lanes (experts) whose tile range ends after m_tile
      if (owner == 0) return;                        // past the last tile (whole workgroup exits)
      const int e = __ffsll((long long)owner) - 1;
      expert = e;
      const int t = m_tile - (__shfl(incl, e, 64) - __shfl(nt_e, e, 64));
      m0 = __shfl(lo, e, 64) + t * TBM;
      m_end = __shfl(hi, e, 64);
    } else {
      int t = m_tile, e = 0;
      for (; e < p.grp_E; ++e) {
        const int lo = p.grp_acc[e], hi = p.grp_acc[e + 1];
        const int nt_e = (hi - lo + TBM - 1) / TBM;
        if (t < nt_e) {
          m0 = lo + t * TBM;
          m_end = hi;
          break;
        }
        t -= nt_e;
      }
      if (e == p.grp_E) return;
      expert = e;
    }
  } else if (m_tile >= p.m_tiles) {
    return;                                         // padding of the last group of 8 row tiles
  }
  // packed ragged batch: the live-row count is a device value.  Its load is requested here and looked at BEHIND the first tiles'
  // loads (rows up to p.M exist, so those are safe to request): in front of them it was one more dependent round trip per launch
  int m_live = 0x7fffffff;
  if (GRP == 0 && p.m_dev != nullptr) m_live = *p.m_dev;
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
    if (GRP == 1) {
      aptr[j] = p.A + (size_t)p.grp_pos[m] * p.lda + 4 * ac;
    } else if (GRP == 2) {
      aptr[j] = reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(p.A) + (size_t)m * p.lda + 8 * ac);
    } else if (CONV) {
      const int f2 = m % p.conv_F2;
      const int t2 = (m / p.conv_F2) % p.conv_T2;
      const int b = m / (p.conv_F2 * p.conv_T2);
      const size_t e0 = ((size_t)(b * p.conv_T1 + 2 * t2) * p.conv_F1 + 2 * f2) * p.conv_C;
      aptr[j] = A16 ? reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(p.A) + e0 + 8 * ac) : p.A + e0 + 4 * ac;
    } else if (A16) {
      aptr[j] = reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(p.A) + (size_t)m * p.lda + 8 * ac);
    } else {
      aptr[j] = p.A + (size_t)m * p.lda + 4 * ac;
    }
    a_zero[j] = p.mask_in ? ((m % p.rows_per_batch) >= p.row_len[m / p.rows_per_batch]) : false;
  }
  // W: thread t brings 16-B chunk (t % CB) of tile rows (t / CB) + RB j
  const int bc = tid % CB, br0 = tid / CB;
  const unsigned char* W = reinterpret_cast<const unsigned char*>(p.W) + (GRP ? (size_t)expert * p.N * p.K * WSZ : 0);
  const unsigned char* bptr[JB];
  int brow_n[JB];
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int tr = br0 + RB * j;
    const int n = GLU ? (tr / (TBN / 2)) * Nout + min(n0 + (tr % (TBN / 2)), Nout - 1) : min(n0 + tr, p.N - 1);
    brow_n[j] = n;
    bptr[j] = W + ((size_t)n * p.K + WCE * bc) * WSZ;
  }
  // slice-major W (plan's expert w_2: [K/64][N][64]): element k of row n sits at ((k>>6)*N + n)*64 + (k&63)
  auto w_ptr = [&](int j, int s) -> const unsigned char* {
    if (GRP == 2 && p.w_sliced) {
      const int k0 = (s * CB + bc) * WCE;
      return W + (((size_t)(k0 >> 6) * p.N + brow_n[j]) * 64 + (k0 & 63)) * WSZ;
    }
    return bptr[j] + (size_t)s * TBK * WSZ;
  };
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

  // ---- fp32 epilogue, row-wise: a lane owns 4 consecutive output columns, a wave sweeps rows ----
  constexpr int LPR = OUTW / 4;                     // lanes per output row (32, GLU 16)
  constexpr int RPI = 64 / LPR;                     // rows per wave iteration (2, GLU 4)
  const int c4 = 4 * (lane % LPR);                  // first of this lane's 4 columns inside the tile
  const int n = n0 + c4;
  float bias0[4], bias1[4], wsum0[4], wsum1[4], wbeta0[4], wbeta1[4], wsc[4];
  const bool vec_ok = ((p.ldy & 3) == 0) && (!p.resid || (p.ldr & 3) == 0) && (n + 3 < Nout);
  // Every load of the sweep is issued BEFORE it (residual rows into registers -- the accumulators are dead --, the row
  // masks resolved here): a load inside the sweep makes hipcc wait vmcnt(0) in every iteration, and vmcnt counts the
  // previous iteration's stores too, so each iteration paid a full store round trip (in-kernel stamps of the LDS-DMA
  // kernel, tools/diag_gemm_dma.py: the sweep was the longest phase of the work-group).
  // EARLY_EPI (64-row tiles: 8 residual float4 per lane): the same operands are requested BEFORE the k loop -- after it they
  // were one more exposed round trip (~1.5 us of an ~11 us launch at configs[2]'s 1984 rows)
  constexpr int IT = TBM / (4 * RPI);
  f32x4 res_all[IT];
  bool pad_all[IT];
  int mo_all[IT];                                   // output row (grouped GEMM-2 with p.y_rows: the row's place in another order)
  auto fetch_epilogue_operands = [&]() {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ne = min(n + e, Nout - 1);
    wsc[e] = W8 ? p.w_scale[(size_t)expert * p.N + ne] : 1.f;
    bias0[e] = p.bias ? p.bias[(GRP ? (size_t)expert * p.N : 0) + ne] : 0.f;
    bias1[e] = (GLU && p.bias) ? p.bias[ne + Nout] : 0.f;
    wsum0[e] = LN ? p.ln_wsum[ne] : 0.f;
    wsum1[e] = (LN && GLU) ? p.ln_wsum[ne + Nout] : 0.f;
    wbeta0[e] = (LN && p.mask_in) ? p.ln_wbeta[ne] : 0.f;
    wbeta1[e] = (LN && GLU && p.mask_in) ? p.ln_wbeta[ne + Nout] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int m = min(m0 + (4 * it + wave) * RPI + lane / LPR, m_end - 1);      // clamped, never branched around
    mo_all[it] = (GRP == 2 && p.y_rows != nullptr) ? p.y_rows[m] : m0 + (4 * it + wave) * RPI + lane / LPR;
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
  };

  const int nsteps = p.K / TBK;
  f32x4 areg[JA];                                   // A16: the 16 bytes are 8 bf16, carried as they are
  u32x4 breg[JB];                                   // 16 bytes of W: 8 bf16, or 16 fp8 (W8)
  auto load_tiles = [&](int s) {
    const int ko = A16 ? a_offset(s * TBK) / 2 : a_offset(s * TBK);       // in floats (A16: two bf16 per float slot)
#pragma unroll
    for (int j = 0; j < JA; ++j) areg[j] = ldg4(aptr[j] + ko);
#pragma unroll
    for (int j = 0; j < JB; ++j) breg[j] = ldg16b(w_ptr(j, s));
  };
  // `on` = 0 for the redundant store after the last k-step (the store stays unconditional, see the header)
  auto store_tiles = [&](int buf, float on) {
    bf16_t* a_dst = As + buf * (TBM * T_LD) + ar0 * T_LD + (A16 ? 8 : 4) * ac;
    bf16_t* b_dst = Bs + buf * (TBN * T_LD) + br0 * T_LD + WCE * bc;
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const f32x4 v = areg[j];
      if (A16) {
        if (LN) {                                    // statistics of the bf16 values themselves
          const bf16x8 h8 = __builtin_bit_cast(bf16x8, v);
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)h8[e];
            t1 += f;
            t2 += f * f;
          }
          s1[j] += on * t1;
          s2[j] += on * t2;
        }
        *reinterpret_cast<f32x4*>(a_dst + RA * j * T_LD) = a_zero[j] ? f32x4{0.f, 0.f, 0.f, 0.f} : v;
        continue;
      }
      if (LN) {
        s1[j] += on * ((v[0] + v[1]) + (v[2] + v[3]));
        s2[j] += on * ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
      }
      bf16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (bf16_t)(a_zero[j] ? 0.f : v[e]);
      *reinterpret_cast<bf16x4*>(a_dst + RA * j * T_LD) = h;
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      if (W8) {
        *reinterpret_cast<bf16x8*>(b_dst + RB * j * T_LD) = fp8x8_to_bf16(breg[j][0], breg[j][1]);
        *reinterpret_cast<bf16x8*>(b_dst + RB * j * T_LD + 8) = fp8x8_to_bf16(breg[j][2], breg[j][3]);
      } else {
        *reinterpret_cast<u32x4*>(b_dst + RB * j * T_LD) = breg[j];
      }
    }
  };

  auto mfma_step = [&](int s) {
    const bf16_t* a_lds = As + (s & 1) * (TBM * T_LD) + ((TBM / 2) * wm + col) * T_LD + 8 * kq;
    const bf16_t* b_lds = Bs + (s & 1) * (TBN * T_LD) + col * T_LD + 8 * kq;
#pragma unroll
    for (int ks = 0; ks < TBK / 32; ++ks) {
      bf16x8 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const bf16x8*>(b_lds + btile(nt) * T_LD + 32 * ks);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_lds + 16 * mt * T_LD + 32 * ks);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16h(a, b[nt], acc[mt][nt]);
      }
    }
  };
  load_tiles(0);
  if (GRP == 0 && m0 > m_live) return;             // no live row in this tile (whole work-group; the requested loads are dropped)
  if constexpr (EARLY_EPI) fetch_epilogue_operands();   // behind the first tiles in the (in-order) return queue
  store_tiles(0, 1.f);
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    load_tiles(min(s + 1, nsteps - 1));             // clamped: the loads are never behind a branch
    __builtin_amdgcn_sched_barrier(0);              // all 12 loads are in flight before the MFMA phase starts
    mfma_step(s);
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

  if constexpr (!EARLY_EPI) fetch_epilogue_operands();
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
      if (W8) y0 *= wsc[e];
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
    if (p.Yb != nullptr && n + 3 < Nout) {          // bf16 copy for the next GEMM's A operand (besides the fp32 output)
      bf16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (bf16_t)y[e];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.Yb) + (size_t)m * p.ldyb + n) = h;
    }
    if (GRP == 1 || p.y_bf16) {   // output itself in bf16 (N % 4 == 0 checked by the launcher)
      bf16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (bf16_t)y[e];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.Y) + (size_t)m * p.ldy + n) = h;
    } else if (vec_ok) {
      stg4(p.Y + (size_t)mo_all[it] * p.ldy + n, y);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < Nout) p.Y[(size_t)mo_all[it] * p.ldy + n + e] = y[e];
    }
  }
}

// > 64 KB of dynamic LDS must be opted into once per kernel (not a stream operation: outside graph capture)
#define M3_TILED_FOR_ALL(X) \
  X(false, true, false) X(true, false, true) X(true, false, false) X(false, false, true) X(false, false, false)
int init_gemm_bf16_tiled_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
#define M3_TILED_ATTR(G_, C_, L_)                                                                                 \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16w_tiled_kernel<128, 128, 64, G_, C_, L_, 0>,            \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, tiled_lds_bytes(128, 128, 64)));   \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16w_tiled_kernel<64, 64, 128, G_, C_, L_, 0>,             \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, tiled_lds_bytes(64, 64, 128)));    \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16w_tiled_kernel<128, 128, 64, G_, C_, L_, 0, false, true>,  \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, tiled_lds_bytes(128, 128, 64)));   \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16w_tiled_kernel<64, 64, 128, G_, C_, L_, 0, false, true>,   \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, tiled_lds_bytes(64, 64, 128)));
  M3_TILED_FOR_ALL(M3_TILED_ATTR)
#undef M3_TILED_ATTR
#define M3_THIN_ATTR(G_, L_)                                                                                       \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16w_tiled_kernel<32, 64, 128, G_, false, L_, 0>,             \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, tiled_lds_bytes(32, 64, 128)));      \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16w_tiled_kernel<32, 64, 128, G_, false, L_, 0, false, true>, \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, tiled_lds_bytes(32, 64, 128)));
  M3_THIN_ATTR(true, true) M3_THIN_ATTR(true, false) M3_THIN_ATTR(false, true) M3_THIN_ATTR(false, false)
#undef M3_THIN_ATTR
#define M3_GRP_ATTR(BM_, BN_, BK_, P_)                                                                            \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16w_tiled_kernel<BM_, BN_, BK_, false, false, false, P_>,  \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, tiled_lds_bytes(BM_, BN_, BK_)))
  M3_GRP_ATTR(128, 128, 64, 1); M3_GRP_ATTR(128, 128, 64, 2); M3_GRP_ATTR(64, 64, 128, 1); M3_GRP_ATTR(64, 64, 128, 2);
#undef M3_GRP_ATTR
#define M3_GRP8_ATTR(BM_, BN_, BK_, P_)                                                                                 \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16w_tiled_kernel<BM_, BN_, BK_, false, false, false, P_, true>,  \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, tiled_lds_bytes(BM_, BN_, BK_)))
  M3_GRP8_ATTR(128, 128, 64, 1); M3_GRP8_ATTR(128, 128, 64, 2); M3_GRP8_ATTR(64, 64, 128, 1); M3_GRP8_ATTR(64, 64, 128, 2);
#undef M3_GRP8_ATTR
  once.mark();
  return 0;
}

bool gemm_bf16w_tiled_supports(const GemmParams& p) {
  return (p.K & 127) == 0 && (p.mode != GEMM_A_CONV3X3S2 || (p.conv_C & 127) == 0);
}

// caller (launch_gemm_bf16w) has validated the operands; returns 0 / error
int launch_gemm_bf16w_tiled(const GemmParams& pin, hipStream_t stream) {
  GemmParams p = pin;
  if (int rc = init_gemm_bf16_tiled_kernels()) return rc;
  const bool glu = p.act == ACT_GLU;
  const bool conv = p.mode == GEMM_A_CONV3X3S2;
  const bool ln = p.ln_wsum != nullptr;
  const int Nout = glu ? p.N / 2 : p.N;
  M3_REQUIRE(!(conv && (glu || ln)), "gemm_bf16w: conv mode supports neither GLU nor LayerNorm");
  const bool big = (long)cdiv(p.M, 128) * cdiv(p.N, 128) >= 200;
  // few 64 x 64 tiles (a ragged batch of ~1000 live rows x a 512- or 1024-wide output: 136-272 live tiles on 256 CUs, each
  // a chain of memory round trips): 32-row tiles double the work-groups.  Measured at configs[2] (M3_TILED_THIN_BELOW=600):
  // one context alone 3.94 -> 3.72 ms, four contexts 2.42 -> 2.31 M frames/s (W tiles are fetched twice as often) -- a
  // latency / throughput trade, off by default
  static const int thin_below = [] { const char* e = getenv("M3_TILED_THIN_BELOW"); return e ? atoi(e) : 0; }();
  const bool thin = !big && !conv && (long)cdiv(p.M, 64) * cdiv(p.N, 64) < thin_below;
  // (k-steps of 256 with one work-group per CU were tried for launches of <= 256 tiles: half the round trips, but 13.2 vs
  //  9.0 us at 1090 x 1024 x 512 and 4.39 vs 3.95 ms per configs[2] forward -- two resident work-groups that overlap each
  //  other's waits are worth more than fewer, longer steps)
  const int bm = big ? 128 : (thin ? 32 : 64), bn = big ? 128 : 64;
  p.m_tiles = cdiv(p.M, bm);
  p.n_tiles = glu ? cdiv(Nout, bn / 2) : cdiv(p.N, bn);
  dim3 grid(cdiv(p.m_tiles, 8) * 8 * p.n_tiles);   // row tiles in groups of 8 (one per XCD), see the kernel
  if (p.a_bf16 && !conv) M3_REQUIRE((p.lda & 7) == 0, "gemm_bf16w: bf16 A needs lda %% 8 == 0");
  if (p.y_bf16 || p.Yb) M3_REQUIRE((Nout & 3) == 0 && (p.ldy & 3) == 0 && (p.ldyb & 3) == 0, "gemm_bf16w: bf16 output needs N %% 4 == 0");
#define M3_TILED_LAUNCH(G_, C_, L_)                                                                                  \
  if (glu == G_ && conv == C_ && ln == L_) {                                                                         \
    if (big && p.a_bf16)                                                                                             \
      hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<128, 128, 64, G_, C_, L_, 0, false, true>), grid, dim3(256),       \
                         tiled_lds_bytes(128, 128, 64), stream, p);                                                  \
    else if (big)                                                                                                    \
      hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<128, 128, 64, G_, C_, L_, 0>), grid, dim3(256),                    \
                         tiled_lds_bytes(128, 128, 64), stream, p);                                                  \
    else if (thin && p.a_bf16) {                                                                                   \
      if constexpr (!C_)                                                                                             \
        hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<32, 64, 128, G_, false, L_, 0, false, true>), grid, dim3(256),   \
                           tiled_lds_bytes(32, 64, 128), stream, p);                                                 \
    } else if (thin) {                                                                                               \
      if constexpr (!C_)                                                                                             \
        hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<32, 64, 128, G_, false, L_, 0>), grid, dim3(256),                \
                           tiled_lds_bytes(32, 64, 128), stream, p);                                                 \
    } else if (p.a_bf16)                                                                                             \
      hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<64, 64, 128, G_, C_, L_, 0, false, true>), grid, dim3(256),        \
                         tiled_lds_bytes(64, 64, 128), stream, p);                                                   \
    else                                                                                                             \
      hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<64, 64, 128, G_, C_, L_, 0>), grid, dim3(256),                     \
                         tiled_lds_bytes(64, 64, 128), stream, p);                                                   \
  }
  M3_TILED_FOR_ALL(M3_TILED_LAUNCH)
#undef M3_TILED_LAUNCH
  M3_LAUNCH_CHECK();
  return 0;
}

// ---- grouped expert FFN on the tiled core (long batches) ----
// H = SiLU(X[pos] W1[e]^T + b1[e]) (bf16, sorted rows), Y = H W2[e]^T (fp32, sorted rows; b2, gate, residual and
// LayerNorm are applied by moe_combine_kernel with one "slab").  hbuf: S*F bf16, ybuf: S*D fp32.
int launch_expert_ffn_bf16w_tiled(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D,
                                  int F, const void* w1, const float* b1, const void* w2, int w2_sliced, void* hbuf,
                                  float* ybuf, hipStream_t stream, const float* b2, float* y_scatter) {
  M3_REQUIRE((D & 127) == 0 && (F & 127) == 0, "expert_ffn tiled: idim=%d / hidden=%d must be multiples of 128", D, F);
  if (y_scatter == nullptr && expert_ffn_bf16_g256(S, E, D, F)) {
    // saturating row counts: 256 x 256 x 64 LDS-DMA tiles (expert_gemm_g256.hip).  Its A operand is bf16 in memory: the rows are
    // converted once (behind ybuf in the slab region: S * D * 2 bytes of its F / 64 * S * D * 4) and gathered by the fills
    void* xb = (char*)ybuf + align_up((size_t)S * D * 4, 256);
    if (int rc = launch_rows_to_bf16(x, ldx, S, D, xb, stream)) return rc;
    return launch_expert_ffn_bf16_g256(xb, D, pos, acc_hist, S, E, D, F, w1, b1, w2, w2_sliced, hbuf, ybuf, stream);
  }
  if (int rc = init_gemm_bf16_tiled_kernels()) return rc;
  // rows per expert ~ S/E: small tiles (4x the workgroups, half the k-steps) until an expert fills 128-row tiles
  const bool big = S / E >= 192;
  const int bm = big ? 128 : 64, bn = big ? 128 : 64;
  const int m_slots = cdiv(cdiv(S, bm) + E, 8 * kGrpRun) * 8 * kGrpRun;   // >= sum_e ceil(cnt_e / bm), padded to 8 XCDs x kGrpRun
  GemmParams g1;
  g1.A = x; g1.lda = ldx; g1.W = (const float*)w1; g1.bias = b1; g1.Y = (float*)hbuf; g1.ldy = F;
  g1.M = S; g1.N = F; g1.K = D; g1.act = ACT_SILU;
  g1.grp_acc = acc_hist; g1.grp_E = E; g1.grp_pos = pos;
  g1.n_tiles = cdiv(F, bn); g1.m_tiles = m_slots;
  GemmParams g2;
  g2.A = (const float*)hbuf; g2.lda = F; g2.W = (const float*)w2; g2.Y = ybuf; g2.ldy = D;
  g2.M = S; g2.N = D; g2.K = F; g2.w_sliced = w2_sliced;
  g2.grp_acc = acc_hist; g2.grp_E = E;
  g2.n_tiles = cdiv(D, bn); g2.m_tiles = m_slots;
  if (y_scatter != nullptr) {   // the expert-parallel receive side: + b2, and every sorted row straight back to the wire row it came from
    g2.Y = y_scatter; g2.y_rows = pos; g2.bias = b2;
  }
#define M3_GRP_LAUNCH(BM_, BN_, BK_)                                                                                 \
  do {                                                                                                               \
    hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<BM_, BN_, BK_, false, false, false, 1>), dim3(m_slots * g1.n_tiles), \
                       dim3(256), tiled_lds_bytes(BM_, BN_, BK_), stream, g1);                                       \
    hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<BM_, BN_, BK_, false, false, false, 2>), dim3(m_slots * g2.n_tiles), \
                       dim3(256), tiled_lds_bytes(BM_, BN_, BK_), stream, g2);                                       \
  } while (0)
  if (big) M3_GRP_LAUNCH(128, 128, 64); else M3_GRP_LAUNCH(64, 64, 128);
#undef M3_GRP_LAUNCH
  M3_LAUNCH_CHECK();
  return 0;
}

// the same with fp8 expert weights + per-row scales (W8A16, see moe_expert_fp8.hip)
int launch_expert_ffn_w8_tiled(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D,
                               int F, const void* w1, const float* s1, const float* b1, const void* w2, const float* s2,
                               int w2_sliced, void* hbuf, float* ybuf, hipStream_t stream) {
  M3_REQUIRE((D & 127) == 0 && (F & 127) == 0, "expert_ffn tiled: idim=%d / hidden=%d must be multiples of 128", D, F);
  if (int rc = init_gemm_bf16_tiled_kernels()) return rc;
  const bool big = S / E >= 192;
  const int bm = big ? 128 : 64, bn = big ? 128 : 64;
  const int m_slots = cdiv(cdiv(S, bm) + E, 8 * kGrpRun) * 8 * kGrpRun;
  GemmParams g1;
  g1.A = x; g1.lda = ldx; g1.W = (const float*)w1; g1.w_scale = s1; g1.bias = b1; g1.Y = (float*)hbuf; g1.ldy = F;
  g1.M = S; g1.N = F; g1.K = D; g1.act = ACT_SILU;
  g1.grp_acc = acc_hist; g1.grp_E = E; g1.grp_pos = pos;
  g1.n_tiles = cdiv(F, bn); g1.m_tiles = m_slots;
  GemmParams g2;
  g2.A = (const float*)hbuf; g2.lda = F; g2.W = (const float*)w2; g2.w_scale = s2; g2.Y = ybuf; g2.ldy = D;
  g2.M = S; g2.N = D; g2.K = F; g2.w_sliced = w2_sliced;
  g2.grp_acc = acc_hist; g2.grp_E = E;
  g2.n_tiles = cdiv(D, bn); g2.m_tiles = m_slots;
#define M3_GRP8_LAUNCH(BM_, BN_, BK_)                                                                                      \
  do {                                                                                                                     \
    hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<BM_, BN_, BK_, false, false, false, 1, true>), dim3(m_slots * g1.n_tiles), \
                       dim3(256), tiled_lds_bytes(BM_, BN_, BK_), stream, g1);                                             \
    hipLaunchKernelGGL((gemm_bf16w_tiled_kernel<BM_, BN_, BK_, false, false, false, 2, true>), dim3(m_slots * g2.n_tiles), \
                       dim3(256), tiled_lds_bytes(BM_, BN_, BK_), stream, g2);                                             \
  } while (0)
  if (big) M3_GRP8_LAUNCH(128, 128, 64); else M3_GRP8_LAUNCH(64, 64, 128);
#undef M3_GRP8_LAUNCH
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
