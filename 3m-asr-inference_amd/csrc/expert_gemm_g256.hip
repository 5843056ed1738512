// Grouped (per-expert) bf16 GEMMs of the expert FFN at saturating row counts, 256 x 256 x 64 tiles fed by LDS-DMA.
//
//   GEMM-1   H[rows of e, F]  = SiLU(X[rows of e, D] . W1[e]^T + b1[e])      bf16 out (sorted rows)
//   GEMM-2   Y[rows of e, D]  = H[rows of e, F] . W2[e]^T                     fp32 out (sorted rows; b2 / gate / residual /
//                                                                             LayerNorm: moe_combine_kernel with one slab)
// Reference operator: compute_fmoe_expert (TRTAPI++/plugin/fmoe_expert_plugin/fmoe_expert_plugin.cpp:82-128: per expert two
// cublasGemm + BiasSilu / Bias kernels) = Expert.forward (trainer_3m_fix/layer/positionwise_feed_forward.py:105-112).
//
// Why another kernel: the 128 x 128 tiles of gemm_bf16_tiled.hip / gemm_bf16_dma.hip pull 64 operand bytes into LDS per
// 4.2 MFLOP; a CU lands ~21 B/cycle there whatever the path, so those kernels sit at ~16-31 % of the bf16
// MFMA peak however they are pipelined.  A 256 x 256 tile does 8.4 MFLOP per 64 KB: twice the FLOPs per byte pulled -- the
// geometry of cdna_hip_programming.md's "256^2" template: 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave = 8 x 4
// accumulator tiles of v_mfma_f32_16x16x32_bf16 (128 registers), one work-group per CU, 2-stage ring of 64 KB stages.
//   * operands: both K-contiguous bf16 rows; a wave instruction (buffer_load_dwordx4 ... lds) lands 8 rows x 128 B; rows are
//     kept as plain 128-B rows whose 16-B chunks are XOR-swizzled by (row >> 1) & 7 on the SOURCE side (LDS-DMA writes
//     lane-linearly), which makes every ds_read_b128 of a fragment conflict-free (as gemm_bf16_dma.hip);
//   * A rows of GEMM-1 are GATHERED through `pos` (the fused local_scatter): a per-lane source address costs LDS-DMA nothing;
//   * the product is formed TRANSPOSED (W fragment as the MFMA's A operand): a lane then holds 4 CONSECUTIVE output columns of
//     one row, i.e. 8 B of bf16 / 16 B of fp32 per accumulator tile for the epilogue's LDS image instead of 2-B pieces;
//   * epilogue through an LDS image of the tile in the (dead) ring: bias + SiLU in registers, swizzled ds_write_b64 / b128,
//     then row-wise 16-B global stores (whole 512-B / 1-KB row segments).
// Row tiles are cut per expert from acc_histogram on the device (wave prefix sum), all column tiles of a row tile run on one XCD.
#include <type_traits>

#include "common.h"
#include "kernels.h"

// -DM3_G256_DIAG: in-kernel cycle accounting (s_memtime) of waves 0 and 4 of every work-group into a debug buffer read back with
// m3_debug_g256_read (tools/diag_g256.py).  Not part of the product build.
#ifdef M3_G256_DIAG
#define G_DIAG(...) __VA_ARGS__
#else
#define G_DIAG(...)
#endif

namespace m3 {

G_DIAG(__device__ unsigned long long g_g256_dbg[4096 * 16];)

namespace {
constexpr int GBM = 256, GBN = 256, GBK = 32;        // K per stage
constexpr int kGStages = 4;
constexpr int kGOp = GBM * GBK * 2;                  // one operand tile of one stage: 16 KB
constexpr int kGStage = 2 * kGOp;                    // A tile, then W tile: 32 KB
constexpr int kGLds = kGStages * kGStage + 32 * 1024; // 4-stage ring (128 KB) + 32 KB: ring buffer 3 and those 32 KB are the epilogue image
#ifndef G256_RUN
#define G256_RUN 8
#endif
#ifndef G256_LATE_READS
#define G256_LATE_READS 3
#endif
constexpr int kGRun = G256_RUN;

__device__ __forceinline__ void g_dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  // (hidden from hipcc's waitcnt pass, M0 reserved: see gemm_bf16_dma.hip)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 g_rsrc(const void* base, size_t bytes) {
  const unsigned long long a = (unsigned long long)base;
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
               (unsigned)__builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)),
               (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bytes > 0xffffffffull ? 0xffffffffull : bytes)), 0x00020000u};
}
// SiLU for a bf16 result: hardware reciprocal instead of the IEEE division sequence (the output keeps 8 mantissa bits)
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
}  // namespace

struct G256Params {
  const bf16_t* A; int lda;                 // bf16 rows; GEMM-1: gathered through pos, GEMM-2: sorted rows
  const int32_t* pos;                       // GEMM-1: sorted row -> source row (NULL: rows are already sorted)
  const bf16_t* W; long w_expert_stride;    // elements between experts
  int w_row_stride, w_kstep_bytes;          // elements between output rows n; bytes between k SLABS of 64 (plain [N][K]: K, 128;
                                            // the plan's slice-major w_2 [K/64][N][64]: 64, N * 128)
  const float* bias; int bias_stride;       // [E][N] or NULL
  void* Y; int ldy;                         // MODE 1: bf16, MODE 2: fp32 (sorted rows)
  const int32_t* acc; int E;                // acc_histogram [E + 1]
  int S, N, K, n_tiles, m_slots;
};

// MODE 1: Y = bf16(SiLU(acc + bias));  MODE 2: Y = fp32 acc
//
// PERSISTENT work-groups (one per CU): tile ids blockIdx.x, + gridDim.x, ...  The fixed part of a tile -- expert lookup, the
// gather indices, the first fills' round trip, bias + SiLU, the LDS image, 128-256 KB of stores -- was 27 k of the 61 k / 91 k
// cycles of a GEMM-1 / GEMM-2 tile in the one-tile-per-work-group form (in-kernel stamps, tools/diag_g256.py).  Here the next
// tile's descriptor and gather indices are fetched while the current tile multiplies, its first three stages are requested
// BEFORE the current tile's epilogue starts (ring buffers 0-2), and the epilogue works through a 64-KB image in ring buffer 3
// + the 32 KB of LDS above the ring: the fills' round trip, and the stores' drain, hide behind the SiLU / image / sweep work.
template <int MODE>
__global__ __launch_bounds__(512, 1) void expert_gemm_g256_kernel(const G256Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char g_lds[];
  constexpr int MT = 8, NT = 4;
  G_DIAG(unsigned long long dg[16]; for (int i_ = 0; i_ < 16; ++i_) dg[i_] = 0; dg[0] = __builtin_amdgcn_s_memtime(); unsigned long long tt_;)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  // (waves w and w + 4 of a work-group share a SIMD: with the rows taken as wave & 1 instead -- the same-row waves paired on a
  //  SIMD -- the kernel is 7-9 % slower, 124 / 94 us against 116 / 86)
  const int wm = wave >> 2, wn = wave & 3;
  const int nsteps = p.K / GBK;

  // ---- tile id -> (expert, rows, column tile): all column tiles of a row tile on one XCD, back to back; kGRun consecutive row
  //      tiles (about one expert's at 2 k rows per expert) on one XCD too: the expert's W tiles then come out of that XCD's L2 ----
  const int lo_ = lane < p.E ? p.acc[lane] : 0, hi_ = lane < p.E ? p.acc[lane + 1] : 0;
  const int nt_e_ = (hi_ - lo_ + GBM - 1) / GBM;
  int incl_ = nt_e_;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl_, d, 64);
    if (lane >= d) incl_ += v;
  }
  struct Tile { int expert, m0, m_end, n0; };
  auto lookup = [&](int id, Tile* t) -> bool {
    const int xcd = id & 7, slot = id >> 3;
    const int n_tile = slot % p.n_tiles;
    const int q_ = slot / p.n_tiles;
    const int m_tile = ((q_ / kGRun) * 8 + xcd) * kGRun + (q_ % kGRun);
    const unsigned long long owner = __ballot(m_tile < incl_);
    if (owner == 0 || id >= p.m_slots * p.n_tiles) return false;
    const int e = __ffsll((long long)owner) - 1;
    t->expert = e;
    t->m0 = __shfl(lo_, e, 64) + (m_tile - (__shfl(incl_, e, 64) - __shfl(nt_e_, e, 64))) * GBM;
    t->m_end = __shfl(hi_, e, 64);
    t->n0 = n_tile * GBN;
    return true;
  };
  // gather sources of this wave's two fill instructions: tile rows 32 wave + 16 j + (lane >> 2)
  auto gather_rows = [&](const Tile& t, int* src) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = min(t.m0 + 32 * wave + 16 * j + (lane >> 2), t.m_end - 1);   // rows past the expert's last one re-read it (never stored)
      src[j] = p.pos ? p.pos[m] : m;
    }
  };

  // ---- fill addressing.  A stage is K = 32: 64-B rows, [256 A rows | 256 W rows] x 64 B = 32 KB, four stages in the 128-KB ring.
  //      Instruction j of a wave carries tile rows 32 wave + 16 j + (lane >> 2); the lane's LDS chunk position lane & 3 holds source
  //      chunk (lane & 3) ^ (((row >> 3) & 1) << 1): with that XOR every ds_read_b128 lane group of a fragment read
  //      (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, ...) covers 16 distinct 16-B slots ----
  const u32x4 rs_a = g_rsrc(p.A, ((size_t)(p.pos ? 0x7fffffff / 2 : p.S - 1) * p.lda + p.K) * 2);
  u32x4 rs_w;
  unsigned voff_a[2], voff_w[2];
  auto setup_fills = [&](const Tile& t, const int* src) {
    rs_w = g_rsrc(p.W + (size_t)t.expert * p.w_expert_stride, (size_t)p.N * p.K * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 32 * wave + 16 * j + (lane >> 2);
      const int c = (lane & 3) ^ (((r >> 3) & 1) << 1);
      voff_a[j] = (unsigned)src[j] * (unsigned)p.lda * 2u + 16u * c;
      voff_w[j] = (unsigned)min(t.n0 + r, p.N - 1) * (unsigned)p.w_row_stride * 2u + 16u * c;
    }
  };
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)g_lds);
  auto issue = [&](int s, int buf) {
    const unsigned dst = lds0 + (unsigned)buf * kGStage + (unsigned)wave * 2048u;
    const unsigned soff_a = (unsigned)s * (GBK * 2);
    const unsigned soff_w = (unsigned)(s >> 1) * (unsigned)p.w_kstep_bytes + (unsigned)(s & 1) * (GBK * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) g_dma16(rs_a, voff_a[j], soff_a, dst + 1024u * j);
#pragma unroll
    for (int j = 0; j < 2; ++j) g_dma16(rs_w, voff_w[j], soff_w, dst + kGOp + 1024u * j);
  };

  // ---- fragments: lane reads row (lane & 15) of a 16-row block, 16-B chunk (lane >> 4), swizzled; two register sets ----
  const int frag_off = col * 64 + 16 * (kq ^ (((col >> 3) & 1) << 1));
  f32x4 acc[MT][NT];
  bf16x8 fa[2][MT], fb[2][NT];
  // The 12 fragment reads of a stage are split over the two phases so that PREP (4 fill issues ~ 366 cycles + reads) and MATH
  // (32 MFMAs ~ 519 cycles + reads) take about the same time: kLateReads of the A fragments (the last ones the MFMAs need) are
  // read in MATH, between the MFMAs of the previous stage; the rest in PREP.
  constexpr int kLateReads = G256_LATE_READS;
  auto read_frags = [&](int buf, int set) {        // PREP part (or everything, for a tile's first stage: late = false)
    const unsigned char* a_lds = g_lds + buf * kGStage + (128 * wm) * 64 + frag_off;
    const unsigned char* b_lds = g_lds + buf * kGStage + kGOp + (64 * wn) * 64 + frag_off;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) fb[set][nt] = *reinterpret_cast<const bf16x8*>(b_lds + 16 * nt * 64);
#pragma unroll
    for (int mt = 0; mt < MT - kLateReads; ++mt) fa[set][mt] = *reinterpret_cast<const bf16x8*>(a_lds + 16 * mt * 64);
  };
  auto read_frags_late = [&](int buf, int set) {
    const unsigned char* a_lds = g_lds + buf * kGStage + (128 * wm) * 64 + frag_off;
#pragma unroll
    for (int mt = MT - kLateReads; mt < MT; ++mt) fa[set][mt] = *reinterpret_cast<const bf16x8*>(a_lds + 16 * mt * 64);
  };
  // MFMAs of stage `set`; late_buf >= 0: the late fragment reads of the NEXT stage (buffer late_buf, the other set) go out after
  // the first MFMAs (their registers were last used by the MFMAs of the stage before this one: free)
  auto multiply = [&](int set, int late_buf) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt == 1 && late_buf >= 0) read_frags_late(late_buf, set ^ 1);
      // TRANSPOSED product: the W fragment is the MFMA's A operand -> this lane holds out[row 16 mt + col][cols 16 nt + 4 kq ..+3]
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16h(fb[set][nt], fa[set][mt], acc[mt][nt]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  Tile cur;
  int id = blockIdx.x;
  if (!lookup(id, &cur)) return;
  {
    int src[2];
    gather_rows(cur, src);
    setup_fills(cur, src);
  }
  G_DIAG(dg[1] = __builtin_amdgcn_s_memtime();)
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);

  for (;;) {
    // ---- the next tile's descriptor and gather indices are requested now and used after the k-loop ----
    Tile nxt;
    const bool more = lookup(id + (int)gridDim.x, &nxt);
    int nsrc[2] = {0, 0};
    if (more) gather_rows(nxt, nsrc);
    // the accumulators START at the bias (GEMM-1): no bias registers live through the k-loop, no add in the epilogue
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (MODE == 1 && p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          b4[r] = p.bias[(size_t)cur.expert * p.bias_stride + min(cur.n0 + 64 * wn + 16 * nt + 4 * kq + r, p.N - 1)];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = b4;
    }

    // ---- k-loop: every K = 32 stage is two PHASES per wave, each closed by a work-group barrier:
    //        PREP(s): issue the fills of stage s+3, wait for the own fills of stage s+2, ds_read the fragments of stage s+1
    //        MATH(s): the 32 MFMAs of stage s (fragments read one PREP earlier)
    //      and the two wave rows (wm = 0 / 1: the two waves of every SIMD) run ONE PHASE APART (the row-1 waves pass one extra
    //      barrier first, the row-0 waves one at the end): while one wave of a SIMD sits in its fill issues and fragment reads
    //      (an LDS-DMA instruction holds its wave for ~90 cycles, a ds_read_b128 for ~24: stamps) the other feeds the matrix pipe.
    //      Visibility: stage s+1 is read in PREP(s); its fills were waited for in PREP(s-1) by row 0 (phase 2s-2) and row 1
    //      (phase 2s-1), each followed by a barrier.  Reuse: stage s+3 lands in the buffer of stage s-1, last read by row 1 in its
    //      PREP(s-2) (phase 2s-3), two barriers earlier.  (Stores of the previous tile's epilogue are older than every fill
    //      waited for here, so the counted waits only ever wait for more, never for less.) ----
    G_DIAG(dg[4] += __builtin_amdgcn_s_memtime() - dg[1]; dg[1] = __builtin_amdgcn_s_memtime();)
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");       // stages 0 and 1 of this tile landed (all waves)
    read_frags(0, 0);
    read_frags_late(0, 0);
    G_DIAG(dg[2] += __builtin_amdgcn_s_memtime() - dg[1]; dg[1] = __builtin_amdgcn_s_memtime();)
    if (wm == 1) asm volatile("s_barrier" ::: "memory");                  // row 1 runs one phase behind
    // (the fragment set is a COMPILE-TIME constant at every use -- a run-time index would put the fragment arrays in scratch --
    //  hence the hand-unrolled pair of steps)
    auto step = [&](int s, auto set_c) {
      constexpr int SET = decltype(set_c)::value;
      G_DIAG(tt_ = __builtin_amdgcn_s_memtime();)
      if (s + 3 < nsteps) {
        issue(s + 3, (s + 3) % kGStages);
        G_DIAG(dg[8] += __builtin_amdgcn_s_memtime() - tt_; tt_ = __builtin_amdgcn_s_memtime();)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // own fills of stage s+2 landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      G_DIAG(dg[9] += __builtin_amdgcn_s_memtime() - tt_; tt_ = __builtin_amdgcn_s_memtime();)
      if (s + 1 < nsteps) read_frags((s + 1) % kGStages, SET ^ 1);
      G_DIAG(dg[10] += __builtin_amdgcn_s_memtime() - tt_; tt_ = __builtin_amdgcn_s_memtime();)
      asm volatile("s_barrier" ::: "memory");
      G_DIAG(dg[11] += __builtin_amdgcn_s_memtime() - tt_; tt_ = __builtin_amdgcn_s_memtime();)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      G_DIAG(dg[12] += __builtin_amdgcn_s_memtime() - tt_; tt_ = __builtin_amdgcn_s_memtime();)
      __builtin_amdgcn_sched_barrier(0);
      multiply(SET, s + 1 < nsteps ? (s + 1) % kGStages : -1);
      G_DIAG(asm volatile("s_nop 0" ::: "memory"); dg[13] += __builtin_amdgcn_s_memtime() - tt_; tt_ = __builtin_amdgcn_s_memtime();)
      // (the late fragment reads were issued ~30 MFMAs ago: returned long since; the wait makes "every read of this phase is done
      //  before the barrier" formal, which is what the ring's reuse argument rests on)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      G_DIAG(dg[14] += __builtin_amdgcn_s_memtime() - tt_;)
    };
    for (int s = 0; s < nsteps; s += 2) {
      step(s, std::integral_constant<int, 0>());
      if (s + 1 < nsteps) step(s + 1, std::integral_constant<int, 1>());
    }
    if (wm == 0) asm volatile("s_barrier" ::: "memory");                  // (pairs with row 1's extra one: the rows are in step again)
    G_DIAG(tt_ = __builtin_amdgcn_s_memtime(); dg[5] += tt_ - dg[1];)

    // ---- the next tile's first three stages go out now (ring buffers 0-2 are free: every wave passed the last barrier with its
    //      fragment reads returned).  What this wave loaded during the k-loop is pinned here, so that hipcc's wait for those loads
    //      sits HERE (nothing else is in flight) and not behind the fills below, deep inside the epilogue ----
    asm volatile("" : "+v"(nsrc[0]), "+v"(nsrc[1]));
    const Tile done = cur;
    if (more) {
      setup_fills(nxt, nsrc);
      issue(0, 0);
      issue(1, 1);
      issue(2, 2);
    }

    // ---- epilogue: registers -> swizzled 64-KB LDS image (ring buffer 3 + the 32 KB above the ring) -> row-wise 16-B stores ----
    unsigned char* img = g_lds + 3 * kGStage;
    const int rows_live = min(GBM, done.m_end - done.m0);
    if (MODE == 1) {
      // two rounds of 128 rows (accumulator rows 16 mt + col with mt in [4 r, 4 r + 4) of BOTH wave rows): image row = 64 wm +
      // 16 (mt - 4 r) + col, 512 B each; the 8-B slot index of a row is XORed with (row & 15) << 2: the 16 rows a wave instruction
      // writes land in 16 different 32-B groups, its 4 column groups (lane >> 4) in the 8-B slots of a group
      bf16_t* Y = reinterpret_cast<bf16_t*>(p.Y);
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        if (rd) __syncthreads();                     // the first round's image has been read
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
          const int mt = 4 * rd + m4;
          const int row = 64 * wm + 16 * m4 + col;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            bf16x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (bf16_t)silu_fast(acc[mt][nt][r]);
            const int g8 = 16 * wn + 4 * nt + kq;                  // 8-B slot of the row (0..63)
            *reinterpret_cast<bf16x4*>(img + row * 512 + 8 * (g8 ^ (col << 2))) = h;
          }
        }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {             // a wave instruction = 2 image rows of 512 B
          const int irow = 16 * it + 2 * wave + (lane >> 5), c16 = lane & 31;
          const f32x4 v = *reinterpret_cast<const f32x4*>(img + irow * 512 + 16 * (c16 ^ ((irow & 15) << 1)));
          const int row = 128 * (irow >> 6) + 64 * rd + (irow & 63);
          if (row < rows_live && done.n0 + 8 * c16 < p.N)
            *reinterpret_cast<f32x4*>(Y + (size_t)(done.m0 + row) * p.ldy + done.n0 + 8 * c16) = v;
        }
      }
    } else {
      // four rounds of 64 rows (mt in [2 r, 2 r + 2) of both wave rows): image row = 32 wm + 16 (mt - 2 r) + col, 1 KB each
      float* Y = reinterpret_cast<float*>(p.Y);
#pragma unroll
      for (int rd = 0; rd < 4; ++rd) {
        if (rd) __syncthreads();
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          const int mt = 2 * rd + m2;
          const int row = 32 * wm + 16 * m2 + col;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int c16 = 16 * wn + 4 * nt + kq;   // 16-B chunk of the row (0..63)
            *reinterpret_cast<f32x4*>(img + row * 1024 + 16 * (c16 ^ col)) = acc[mt][nt];
          }
        }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {             // a wave instruction = 1 image row of 1 KB
          const int irow = 8 * it + wave, c16 = lane;
          const f32x4 v = *reinterpret_cast<const f32x4*>(img + irow * 1024 + 16 * (c16 ^ (irow & 15)));
          const int row = 128 * (irow >> 5) + 32 * rd + (irow & 31);
          if (row < rows_live && done.n0 + 4 * c16 < p.N) stg4(Y + (size_t)(done.m0 + row) * p.ldy + done.n0 + 4 * c16, v);
        }
      }
    }
    G_DIAG(dg[6] += __builtin_amdgcn_s_memtime() - tt_; dg[3] += 1;)
    if (!more) break;
    __syncthreads();                                 // the image is read: ring buffer 3 may take stage 3 of the next tile
    cur = nxt;
    id += (int)gridDim.x;
    G_DIAG(dg[1] = __builtin_amdgcn_s_memtime();)
  }
  G_DIAG(dg[7] = __builtin_amdgcn_s_memtime();
         if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 2048) {
           unsigned long long* o = g_g256_dbg + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 16;
           for (int i_ = 0; i_ < 16; ++i_) o[i_] = dg[i_];
           o[15] = MODE;
         })
}

int init_expert_gemm_g256_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_gemm_g256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kGLds));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_gemm_g256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kGLds));
  once.mark();
  return 0;
}

// rows per expert from which the 256-row tiles pay (M3_G256_MIN_ROWS_PER_EXPERT overrides, read once)
bool expert_ffn_bf16_g256(int S, int E, int D, int F) {
  static const int min_rpe = [] { const char* e = getenv("M3_G256_MIN_ROWS_PER_EXPERT"); return e ? atoi(e) : 512; }();
  return E <= 64 && S / E >= min_rpe && (D % 256) == 0 && (F % 256) == 0 && (size_t)S * F * 2 < ((size_t)1 << 32);
}

// xb: bf16 copy of the MoE input rows [S][D] (row stride ldxb elements), gathered through pos by the LDS-DMA fills.
int launch_expert_ffn_bf16_g256(const void* xb, int ldxb, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D, int F,
                                const void* w1, const float* b1, const void* w2, int w2_sliced, void* hbuf, float* ybuf,
                                hipStream_t stream) {
  M3_REQUIRE(expert_ffn_bf16_g256(S, E, D, F), "expert_ffn g256: needs E <= 64 and D, F multiples of 256 (S=%d E=%d D=%d F=%d)", S, E, D, F);
  M3_REQUIRE((ldxb & 7) == 0, "expert_ffn g256: row stride of the bf16 rows must be a multiple of 8");
  if (int rc = init_expert_gemm_g256_kernels()) return rc;
  const int m_slots = cdiv(cdiv(S, GBM) + E, 8 * kGRun) * 8 * kGRun;   // >= sum_e ceil(cnt_e / 256), padded to 8 XCDs x kGRun
  G256Params g1;
  g1.A = (const bf16_t*)xb; g1.lda = ldxb; g1.pos = pos;
  g1.W = (const bf16_t*)w1; g1.w_expert_stride = (long)F * D; g1.w_row_stride = D; g1.w_kstep_bytes = 128;
  g1.bias = b1; g1.bias_stride = F; g1.Y = hbuf; g1.ldy = F; g1.acc = acc_hist; g1.E = E;
  g1.S = S; g1.N = F; g1.K = D; g1.n_tiles = F / GBN; g1.m_slots = m_slots;
  G256Params g2;
  g2.A = (const bf16_t*)hbuf; g2.lda = F; g2.pos = nullptr;
  g2.W = (const bf16_t*)w2; g2.w_expert_stride = (long)D * F;
  g2.w_row_stride = w2_sliced ? 64 : F; g2.w_kstep_bytes = w2_sliced ? D * 128 : 128;     // (per 64-wide k slab)
  g2.bias = nullptr; g2.bias_stride = 0; g2.Y = ybuf; g2.ldy = D; g2.acc = acc_hist; g2.E = E;
  g2.S = S; g2.N = D; g2.K = F; g2.n_tiles = D / GBN; g2.m_slots = m_slots;
  M3_REQUIRE(D / GBK >= 4 && F / GBK >= 4, "expert_ffn g256: K must cover at least four stages of %d", GBK);
  const int cus = device_cu_count();                // persistent: one work-group per CU walks tile ids id, id + grid, ...
  const int t1 = m_slots * g1.n_tiles, t2 = m_slots * g2.n_tiles;
  hipLaunchKernelGGL((expert_gemm_g256_kernel<1>), dim3(t1 < cus ? t1 : cus), dim3(512), kGLds, stream, g1);
  hipLaunchKernelGGL((expert_gemm_g256_kernel<2>), dim3(t2 < cus ? t2 : cus), dim3(512), kGLds, stream, g2);
  M3_LAUNCH_CHECK();
  return 0;
}

// x fp32 [S][ldx] -> xb bf16 [S][D] (the engine's long-batch modes get the bf16 rows from the kernel that produces xn; the
// stand-alone operator converts here)
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ x, int ldx, int D, bf16_t* __restrict__ xb, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / (D / 4);
    const int c = (int)(i - row * (D / 4)) * 4;
    const f32x4 v = ldg4(x + row * ldx + c);
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x4*>(xb + row * D + c) = h;
  }
}
int launch_rows_to_bf16(const float* x, int ldx, int S, int D, void* xb, hipStream_t stream) {
  M3_REQUIRE((D & 3) == 0 && (ldx & 3) == 0, "rows_to_bf16: D and ldx must be multiples of 4");
  const size_t n4 = (size_t)S * (D / 4);
  if (n4 == 0) return 0;
  hipLaunchKernelGGL(rows_to_bf16_kernel, dim3(grid1d(n4, 8192)), dim3(256), 0, stream, x, ldx, D, (bf16_t*)xb, n4);
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3

#ifdef M3_G256_DIAG
extern "C" int m3_debug_g256_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(m3::g_g256_dbg), bytes < sizeof(m3::g_g256_dbg) ? bytes : sizeof(m3::g_g256_dbg));
}
#endif
