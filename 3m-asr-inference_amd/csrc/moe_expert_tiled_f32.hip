// Grouped expert FFN for long batches, exact fp32 (v_mfma_f32_16x16x4_f32): two LDS-tiled grouped GEMMs.
//
// Same operator as moe_expert.hip (reference compute_fmoe_expert, fmoe_expert_plugin.cpp:36-142: per expert
// cublasSgemm -> BiasSilu -> cublasSgemm -> Bias on 8 streams with two host syncs), for the regime where experts have
// tens to hundreds of rows each (S >= 1024).  The slab form writes F/64 partial outputs per row (32 KB/row) and stays
// near 39 TFLOP/s however many rows an expert has; here
//   GEMM-1: H[r][:]  = SiLU(X[pos[r]][:] . W1[e]^T + b1[e])   rows gathered through pos (fused local_scatter), H fp32
//   GEMM-2: Ys[r][:] = H[r][:] . W2[e]^T                       W2 in the plan's slice-major layout or [D][F]
// run on row tiles cut per expert from acc_histogram (wave prefix sum on the device, no host round trip), and
// moe_combine_kernel (one "slab") adds b2, gate, residual, LayerNorm and un-permutes.  H and Ys live in the workspace
// region the slabs would use.  Tile / pipeline design = gemm_bf16_tiled.hip (coalesced 16-B staging loads issued one
// k-step ahead, 2-stage LDS ring with 16-B padded rows, 2 x 2 waves of (TBM/2) x (TBN/2), two workgroups per CU,
// LDS-staged row-wise epilogue, row tiles spread XCD-aware); fp32 MFMA is 16x slower than bf16, so this kernel is
// MFMA-bound rather than L2-bound: a 128x128x32 step is 4096 MFMA cycles against 32 KB of loads.
// k order inside a 16-deep sub-step: lane (col, kq) feeds its float4 k = 4kq..4kq+3 into four MFMAs (j = 0..3), the
// same permutation on both operands, so the product is exact.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace m3 {

namespace {
constexpr int f32_tiled_lds_bytes(int BM, int BN, int BK) {
  const int ring = 2 * (BM + BN) * (BK + 4) * 4, image = BM * (BN + 4) * 4;
  return ring > image ? ring : image;
}

struct GroupedParams {
  const float* A; int lda;                 // PHASE 1: token rows (gathered through pos); PHASE 2: H, sorted rows
  const int32_t* pos;                      // PHASE 1 only
  const int32_t* acc; int E;               // acc_histogram [E+1]
  const float* W; int w_sliced;            // [E][N][K], or slice-major [E][K/64][N][64] (PHASE 2, plan layout)
  const float* bias;                       // [E][N] or null
  float* Y; int ldy;                       // sorted rows out
  int S, N, K, n_tiles;
};
}  // namespace

template <int TBM, int TBN, int TBK, int PHASE>
__global__ __launch_bounds__(256, 2) void expert_gemm_f32_tiled_kernel(const GroupedParams p) {
  constexpr int T_LD = TBK + 4;                     // floats per LDS row (144 / 272 B: conflict-free 16-B reads)
  constexpr int C_LD = TBN + 4;
  constexpr int MT = TBM / 32, NT = TBN / 32;
  constexpr int CA = TBK / 4, RA = 256 / CA, JA = TBM / RA, JB = TBN / RA;   // float4 chunks per row, rows per pass, passes
  extern __shared__ __attribute__((aligned(16))) unsigned char tiled_lds_f32[];
  float* As = reinterpret_cast<float*>(tiled_lds_f32);             // [2][TBM][T_LD]
  float* Bs = As + 2 * TBM * T_LD;                                  // [2][TBN][T_LD]
  float* Cs = reinterpret_cast<float*>(tiled_lds_f32);             // [TBM][C_LD] after the k loop

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_tile = slot % p.n_tiles, m_tile = (slot / p.n_tiles) * 8 + xcd;

  // ---- which expert owns row tile m_tile (tiles counted over experts in order) ----
  int m0, m_end, expert;
  if (p.E <= 64) {
    const int lo = lane < p.E ? p.acc[lane] : 0, hi = lane < p.E ? p.acc[lane + 1] : 0;
    const int nt_e = (hi - lo + TBM - 1) / TBM;
    int incl = nt_e;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    const unsigned long long owner = __ballot(m_tile < incl);
    if (owner == 0) return;                          // past the last tile (whole workgroup exits)
    expert = __ffsll((long long)owner) - 1;
    const int t = m_tile - (__shfl(incl, expert, 64) - __shfl(nt_e, expert, 64));
    m0 = __shfl(lo, expert, 64) + t * TBM;
    m_end = __shfl(hi, expert, 64);
  } else {
    int t = m_tile, e = 0;
    m0 = m_end = 0;
    for (; e < p.E; ++e) {
      const int lo = p.acc[e], hi = p.acc[e + 1];
      const int nt_e = (hi - lo + TBM - 1) / TBM;
      if (t < nt_e) {
        m0 = lo + t * TBM;
        m_end = hi;
        break;
      }
      t -= nt_e;
    }
    if (e == p.E) return;
    expert = e;
  }
  const int n0 = n_tile * TBN;

  // ---- staging maps: thread t brings float4 chunk (t % CA) of tile rows (t / CA) + RA j ----
  const int ac = tid % CA, ar0 = tid / CA;
  const float* aptr[JA];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int m = min(m0 + ar0 + RA * j, m_end - 1);
    aptr[j] = p.A + (size_t)(PHASE == 1 ? p.pos[m] : m) * p.lda + 4 * ac;
  }
  const float* W = p.W + (size_t)expert * p.N * p.K;
  int brow_n[JB];
#pragma unroll
  for (int j = 0; j < JB; ++j) brow_n[j] = min(n0 + ar0 + RA * j, p.N - 1);
  auto w_ptr = [&](int j, int s) -> const float* {
    const int k = s * TBK + 4 * ac;                 // first of this thread's 4 k
    if (PHASE == 2 && p.w_sliced) return W + ((size_t)(k >> 6) * p.N + brow_n[j]) * 64 + (k & 63);
    return W + (size_t)brow_n[j] * p.K + k;
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = p.K / TBK;
  f32x4 areg[JA], breg[JB];
  auto load_tiles = [&](int s) {
#pragma unroll
    for (int j = 0; j < JA; ++j) areg[j] = ldg4(aptr[j] + s * TBK);
#pragma unroll
    for (int j = 0; j < JB; ++j) breg[j] = ldg4(w_ptr(j, s));
  };
  auto store_tiles = [&](int buf) {                  // unconditional (a store under a branch sinks the loads feeding it)
    float* a_dst = As + buf * (TBM * T_LD) + ar0 * T_LD + 4 * ac;
    float* b_dst = Bs + buf * (TBN * T_LD) + ar0 * T_LD + 4 * ac;
#pragma unroll
    for (int j = 0; j < JA; ++j) *reinterpret_cast<f32x4*>(a_dst + RA * j * T_LD) = areg[j];
#pragma unroll
    for (int j = 0; j < JB; ++j) *reinterpret_cast<f32x4*>(b_dst + RA * j * T_LD) = breg[j];
  };

  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    load_tiles(min(s + 1, nsteps - 1));
    __builtin_amdgcn_sched_barrier(0);
    const float* a_lds = As + (s & 1) * (TBM * T_LD) + ((TBM / 2) * wm + col) * T_LD + 4 * kq;
    const float* b_lds = Bs + (s & 1) * (TBN * T_LD) + ((TBN / 2) * wn + col) * T_LD + 4 * kq;
#pragma unroll
    for (int ks = 0; ks < TBK / 16; ++ks) {
      f32x4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(b_lds + 16 * nt * T_LD + 16 * ks);
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
    store_tiles((s + 1) & 1);
    __syncthreads();
  }

  // ---- accumulators -> LDS image -> row-wise coalesced epilogue ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[((TBM / 2) * wm + 16 * mt + 4 * kq + r) * C_LD + (TBN / 2) * wn + 16 * nt + col] = acc[mt][nt][r];
  __syncthreads();

  constexpr int LPR = TBN / 4;                      // lanes per output row
  constexpr int RPI = 64 / LPR;                     // rows per wave iteration
  const int c4 = 4 * (lane % LPR);
  const int n = n0 + c4;
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr && n < p.N) bias = ldg4(p.bias + (size_t)expert * p.N + n);   // N % 4 == 0 (launcher)
  for (int it = 0; it < TBM / (4 * RPI); ++it) {
    const int row = (4 * it + wave) * RPI + lane / LPR;
    const int m = m0 + row;
    if (m >= m_end || n >= p.N) continue;
    f32x4 y = *reinterpret_cast<const f32x4*>(Cs + row * C_LD + c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[e] += bias[e];
      if (PHASE == 1) y[e] = silu(y[e]);
    }
    stg4(p.Y + (size_t)m * p.ldy + n, y);
  }
}

bool expert_ffn_f32_tiled(int S, int E, int D, int F) {
  static const int min_rows = [] {
    const char* e = getenv("M3_EXPERT_TILED_MIN_ROWS");
    return e ? atoi(e) : 1024;
  }();
  const size_t slab = expert_ffn_slab_bytes(S, D, F);
  const size_t need = align_up((size_t)S * F * 4, 256) + (size_t)S * D * 4;
  return S >= min_rows && (D & 63) == 0 && (F & 63) == 0 && need <= slab;
}
float* expert_ffn_f32_rows(float* slab, int S, int E, int D, int F) {
  return expert_ffn_f32_tiled(S, E, D, F) ? (float*)((char*)slab + align_up((size_t)S * F * 4, 256)) : slab;
}
int expert_ffn_f32_slices(int S, int E, int D, int F) { return expert_ffn_f32_tiled(S, E, D, F) ? 1 : F / kExpertSlice; }

int init_expert_ffn_f32_tiled_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
#define M3_F32T_ATTR(BM_, BN_, BK_, P_)                                                                      \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_gemm_f32_tiled_kernel<BM_, BN_, BK_, P_>,              \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, f32_tiled_lds_bytes(BM_, BN_, BK_)))
  M3_F32T_ATTR(128, 128, 32, 1); M3_F32T_ATTR(128, 128, 32, 2); M3_F32T_ATTR(64, 64, 64, 1); M3_F32T_ATTR(64, 64, 64, 2);
#undef M3_F32T_ATTR
  once.mark();
  return 0;
}

int launch_expert_ffn_f32_tiled(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D,
                                int F, const float* w1, const float* b1, const float* w2, int w2_sliced, float* hbuf,
                                float* ybuf, hipStream_t stream) {
  M3_REQUIRE((D & 63) == 0 && (F & 63) == 0 && (ldx & 3) == 0, "expert_ffn tiled: idim=%d / hidden=%d must be multiples of 64", D, F);
  if (int rc = init_expert_ffn_f32_tiled_kernels()) return rc;
  const bool big = S / E >= 192;                     // rows per expert fill 128-row tiles
  const int bm = big ? 128 : 64, bn = big ? 128 : 64;
  const int m_slots = cdiv(cdiv(S, bm) + E, 8) * 8;  // upper bound of sum_e ceil(cnt_e / bm), padded to the 8 XCDs
  GroupedParams g1{x, ldx, pos, acc_hist, E, w1, 0, b1, hbuf, F, S, F, D, cdiv(F, bn)};
  GroupedParams g2{hbuf, F, nullptr, acc_hist, E, w2, w2_sliced, nullptr, ybuf, D, S, D, F, cdiv(D, bn)};
#define M3_F32T_LAUNCH(BM_, BN_, BK_)                                                                            \
  do {                                                                                                           \
    hipLaunchKernelGGL((expert_gemm_f32_tiled_kernel<BM_, BN_, BK_, 1>), dim3(m_slots * g1.n_tiles), dim3(256),  \
                       f32_tiled_lds_bytes(BM_, BN_, BK_), stream, g1);                                          \
    hipLaunchKernelGGL((expert_gemm_f32_tiled_kernel<BM_, BN_, BK_, 2>), dim3(m_slots * g2.n_tiles), dim3(256),  \
                       f32_tiled_lds_bytes(BM_, BN_, BK_), stream, g2);                                          \
  } while (0)
  if (big) M3_F32T_LAUNCH(128, 128, 32); else M3_F32T_LAUNCH(64, 64, 64);
#undef M3_F32T_LAUNCH
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
