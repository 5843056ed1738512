// bf16 x bf16 GEMM for long batches, fed by LDS-DMA:  Y[M,N] = epilogue( A[M,K] . W[N,K]^T ),  A and W both bf16 in memory.
//
// Replaces the TensorRT-native matmuls of the reference (TRTAPI++/python/trt_helper/torch_network_helper.py:573-605) for the
// dense GEMMs of a block once the engine keeps bf16 activation copies (BASELINE.json configs[2] / [4]: 1 k - 5 k live rows).
// The register-staged kernel of gemm_bf16_tiled.hip spends its time in the staging itself (global load -> VGPR -> convert ->
// ds_write, one k-step of prefetch: a chain of ~1.3 us loaded round trips); here no operand byte passes through a register
// on its way to LDS:
//   * tile 128 x 128 x 64, 4 waves as 2 x 2, 64 x 64 per wave = 4 x 4 tiles of v_mfma_f32_16x16x32_bf16;
//   * both operands are K-contiguous rows, so ONE staging scheme serves both: a wave instruction
//     (buffer_load_dwordx4 ... lds, 64 lanes x 16 B = 1 KB) brings 8 rows x 128 B = 8 full lines; LDS keeps plain 128-B rows
//     whose 16-B chunks are XOR-swizzled by (row >> 1) & 7 -- applied to the per-lane SOURCE address, because LDS-DMA writes
//     lane-linearly -- which makes every ds_read_b128 of an MFMA fragment conflict-free (checked against the bank rule of
//     MI355X_MICROARCH.md: 16-lane groups {0-3,12-15,20-27}, ... each cover 16 distinct 16-B slots);
//   * 2-stage LDS ring (64 KB), the fills of k-step s+1 are issued before the MFMAs of k-step s and waited for (vmcnt(0))
//     just before the one barrier of the step; two work-groups share a CU, so one computes while the other waits;
//   * the fills are inline assembly: a buffer_load ... lds that hipcc can see makes it put s_waitcnt vmcnt(0) in front of
//     every later LDS read, which would serialise fill and compute inside a work-group;
//   * epilogue through an LDS image of the tile (row-wise 16-B stores), identical in arithmetic to gemm_bf16_tiled.hip:
//     folded LayerNorm, bias, ReLU / SiLU / GLU, mask, scale, residual, fp32 and / or bf16 outputs.
// Folded LayerNorm (ln_wsum): the row statistics cannot be taken while staging (nothing passes through registers); they come
// from p.ln_stats -- per row, `ln_stat_parts` partial (sum, sum of squares) pairs written by whichever kernel produced the
// bf16 operand (this kernel's own epilogue with p.Yb_stats, moe_combine, layernorm) from exactly the bf16 values it stored.
#include "common.h"
#include "kernels.h"

// -DM3_DMA_DIAG: in-kernel phase stamps (s_memtime) of every work-group into a debug buffer, read back with
// m3_debug_dma_read (tools/diag_gemm_dma.py); the stamps go nowhere else.  Not part of the product build.
#ifdef M3_DMA_DIAG
#define M3_DIAG(...) __VA_ARGS__
#else
#define M3_DIAG(...)
#endif

namespace m3 {

M3_DIAG(__device__ unsigned long long g_dma_dbg[4096 * 8];)

namespace {
constexpr int DBM = 128, DBN = 128, DBK = 64;
constexpr int kOpBytes = DBM * DBK * 2;             // one operand tile of one stage: 16 KB
constexpr int kStageBytes = 2 * kOpBytes;           // A tile, then W tile
constexpr int kCLd = DBN + 4;                       // fp32 elements per row of the epilogue image
constexpr int kImageBytes = DBM * kCLd * 4;
constexpr int dma_lds_bytes(int stages) { return stages * kStageBytes > kImageBytes ? stages * kStageBytes : kImageBytes; }

// LDS-DMA fill, hidden from hipcc's waitcnt pass (see the header): 64 lanes x 16 B land at lds_addr + 16 lane
// (M0 is not on the clobber list: it is a RESERVED register for hipcc -- naming it draws "clobber list contains reserved
//  registers: m0 ... may lead to undefined behaviour"; the compiler never keeps a value live in M0 across statements, it
//  re-materialises M0 in front of each of its own uses (LDS-DMA / s_movrel lowering), which is what makes this safe.)
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 make_rsrc(const void* base, size_t bytes) {
  const unsigned long long a = (unsigned long long)base;
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
               (unsigned)__builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)),
               (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bytes > 0xffffffffull ? 0xffffffffull : bytes)), 0x00020000u};
}
}  // namespace

// STAGES = 2: 64 KB of LDS, two work-groups per CU (one computes while the other waits for its fills);
// STAGES = 4: 128 KB, one work-group per CU with three k-steps of fills in flight -- for launches whose work-groups do not
// outnumber the CUs (nothing else on the CU could cover a fill's ~1300-cycle round trip: measured 1313 cycles per k-step
// against 512 of MFMA with one step in flight)
template <bool GLU, bool LN, int STAGES>
__global__ __launch_bounds__(256, STAGES == 2 ? 2 : 1) void gemm_bf16_dma_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dma_lds[];
  __shared__ float stats[DBM][2];
  __shared__ float stat_out[DBM][2];
  __shared__ int rowpad[DBM];
  constexpr int MT = 4, NT = 4;
  M3_DIAG(unsigned long long dg[8]; dg[0] = __builtin_amdgcn_s_memtime();)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int Nout = GLU ? (p.N >> 1) : p.N;
  constexpr int OUTW = GLU ? DBN / 2 : DBN;
  // XCD-aware tile order (as gemm_bf16_tiled.hip): all column tiles of a row tile on one XCD, back to back
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_tile = slot % p.n_tiles;
  const int m_tile = (slot / p.n_tiles) * 8 + xcd;
  const int m0 = m_tile * DBM;
  if (m_tile >= p.m_tiles) return;                  // padding of the grid
  // packed ragged batch: the live-row count is a device value; its load is in flight while the fill addresses are formed
  const int m_live = p.m_dev != nullptr ? *p.m_dev : p.M;
  const int n0 = n_tile * OUTW;
  auto btile = [&](int nt) { return GLU ? (nt / 2) * (DBN / 2) + wn * (DBN / 4) + 16 * (nt % 2) : wn * (DBN / 2) + 16 * nt; };

  // ---- fill addressing: instruction j (0..3) of this wave carries tile rows 32 wave + 8 j + (lane >> 3); the lane's LDS
  //      chunk position lane & 7 holds source chunk (lane & 7) ^ ((row >> 1) & 7) ----
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const u32x4 rs_a = make_rsrc(A, ((size_t)(p.M - 1) * p.lda + p.K) * 2);
  const u32x4 rs_w = make_rsrc(p.W, (size_t)p.N * p.K * 2);
  unsigned voff_a[4], voff_w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 32 * wave + 8 * j + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int m = min(m0 + r, p.M - 1);
    voff_a[j] = (unsigned)m * (unsigned)p.lda * 2u + 16u * c;
    const int n = GLU ? (r / (DBN / 2)) * Nout + min(n0 + (r % (DBN / 2)), Nout - 1) : min(n0 + r, p.N - 1);
    voff_w[j] = (unsigned)n * (unsigned)p.K * 2u + 16u * c;
  }
  if (m0 > m_live) return;                          // no live row in this tile (the row AT the count is still computed)
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dma_lds);
  auto issue = [&](int s, int buf) {
    const unsigned soff = (unsigned)s * (DBK * 2);
    const unsigned dst = lds0 + (unsigned)buf * kStageBytes + (unsigned)wave * 4096u;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(rs_a, voff_a[j], soff, dst + 1024u * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(rs_w, voff_w[j], soff, dst + kOpBytes + 1024u * j);
  };

  // ---- fragment addressing: lane reads row (lane & 15) of a 16-row block, chunk 4 kb + (lane >> 4), swizzled ----
  const int frag_off0 = col * 128 + 16 * ((0 + kq) ^ (col >> 1));
  const int frag_off1 = col * 128 + 16 * ((4 + kq) ^ (col >> 1));

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const unsigned char* a_lds = dma_lds + buf * kStageBytes + (64 * wm) * 128;
    const unsigned char* b_lds = dma_lds + buf * kStageBytes + kOpBytes;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int fo = kb ? frag_off1 : frag_off0;
      bf16x8 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const bf16x8*>(b_lds + btile(nt) * 128 + fo);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_lds + 16 * mt * 128 + fo);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16h(a, b[nt], acc[mt][nt]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  const int nsteps = p.K / DBK;
  M3_DIAG(dg[1] = __builtin_amdgcn_s_memtime();)
  // prologue: STAGES - 1 k-steps of fills in flight, the first one waited for
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t)
    if (t < nsteps) issue(t, t);
  if (STAGES == 2 || nsteps == 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  else if (STAGES == 4 && nsteps >= 3) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  M3_DIAG(dg[2] = __builtin_amdgcn_s_memtime();)
  for (int s = 0; s < nsteps; ++s) {
    // the stage freed by step s-1 (every wave passed that step's barrier with its fragment reads returned) takes step s+STAGES-1
    if (s + STAGES - 1 < nsteps) issue(s + STAGES - 1, (s + STAGES - 1) % STAGES);
    compute(s % STAGES);
    // own fills of step s+1 landed (8 fills per step and wave stay counted for each later step in flight), own fragment reads
    // returned, then the step's barrier
    const int ahead = min(STAGES - 2, nsteps - 2 - s);   // k-steps that may stay in flight beyond step s+1
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  M3_DIAG(dg[3] = __builtin_amdgcn_s_memtime();)
  // ---- epilogue.  Measured with in-kernel stamps (tools/diag_gemm_dma.py): the row sweep, not the k-loop, was the longest
  //      phase (13-19 k cycles of 28-45 k): a load inside the sweep (residual, row length) makes hipcc wait vmcnt(0) in every
  //      iteration, and vmcnt counts the previous iteration's STORES too, so each iteration paid a full store round trip.
  //      Now every load of the sweep is issued up front -- residual rows into registers (the accumulators are dead by then),
  //      row masks into LDS -- and the sweep itself is loads-free: stores stream back to back. ----
  constexpr int LPR = OUTW / 4;                     // lanes per output row (32, GLU 16)
  constexpr int RPI = 64 / LPR;                     // rows per wave iteration (2, GLU 4)
  constexpr int IT = DBM / (4 * RPI);               // iterations of the sweep (16, GLU 8)
  const int c4 = 4 * (lane % LPR);
  const int n = n0 + c4;
  // 16-byte row accesses everywhere (every shape of the model but the logits, N = 1434): the unrolled, loads-free sweep;
  // otherwise a rolled sweep with element-wise accesses (slow path, correctness only)
  const bool fast = ((p.ldy & 3) == 0) && (!p.resid || (p.ldr & 3) == 0) && ((Nout & 3) == 0);
  f32x4 res[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    res[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.resid && fast) {
      const int m = min(m0 + (4 * it + wave) * RPI + lane / LPR, p.M - 1);       // clamped, never branched around
      res[it] = ldg4(p.resid + (size_t)m * p.ldr + min(n, Nout - 4));
    }
  }
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
  // accumulators -> LDS image (the ring is dead: every wave passed the loop's last barrier)
  float* Cs = reinterpret_cast<float*>(dma_lds);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(64 * wm + 16 * mt + 4 * kq + r) * kCLd + btile(nt) + col] = acc[mt][nt][r];
  if (tid < DBM) {
    const int m = min(m0 + tid, p.M - 1);
    if (LN) {   // row statistics of the bf16 operand, summed over the partials its producer left
      float t1 = 0.f, t2 = 0.f;
      for (int q = 0; q < p.ln_stat_parts; ++q) {
        t1 += p.ln_stats[((size_t)m * p.ln_stat_parts + q) * 2];
        t2 += p.ln_stats[((size_t)m * p.ln_stat_parts + q) * 2 + 1];
      }
      const float mean = t1 / (float)p.K;
      stats[tid][0] = mean;
      stats[tid][1] = rsqrtf(fmaxf(t2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
    }
    rowpad[tid] = (p.mask_in || p.mask_out) ? ((m % p.rows_per_batch) >= p.row_len[m / p.rows_per_batch] ? 1 : 0) : 0;
  }
  __syncthreads();
  M3_DIAG(dg[4] = __builtin_amdgcn_s_memtime();)

  // one row of the sweep; rv = the residual values of this lane's 4 columns, (v0, v1, pad, mean, rstd) = what the LDS image and
  // the row tables hold for it.  In the fast path these were all read AHEAD of the sweep: with the ds_read inside, every
  // iteration was the serial chain ds_read -> wait -> arithmetic -> store (~600 cycles per iteration and wave in the stamps, 9.7 k
  // cycles per tile: twice the k-loop's MFMA time); read up front the sweep is arithmetic and stores only.
  // Stores are ISSUE-bound here, not byte-bound (MI355X_MICROARCH.md, store tail: a CU retires a store wave-instruction every
  // ~50-70 cycles whatever its width -- the stamps showed 9 k cycles for the 192 instructions of a tile with fp32 + bf16 + stats
  // outputs).  So the sweep issues as few as the bytes allow: fp32 rows as dwordx4 (64 per tile); the bf16 rows as dwordx4 too --
  // a lane holds 4 columns = 8 bytes, so lanes 2i / 2i+1 trade halves over TWO row iterations (even lane: 8 columns of row
  // `it`, odd lane: 8 columns of row `it + 1`; 32 per tile instead of 64); the row statistics go to LDS and leave as two
  // instructions per tile after the sweep instead of one per row.
  auto sweep_row = [&](int it, const f32x4& rv, bool vec, const f32x4& v0, const f32x4& v1, bool pad, float mean, float rstd) -> bf16x4 {
    const int row = (4 * it + wave) * RPI + lane / LPR;
    const int m = m0 + row;
    const bool live = m < p.M && n < Nout;
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
      if (p.resid) t += rv[e];
      y[e] = t;
    }
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (bf16_t)y[e];
    if (p.Yb_stats != nullptr) {
      // (sum, sum of squares) of the bf16 values of this row's OUTW columns: one partial per column tile, read back by the
      // folded-LayerNorm GEMM that consumes Yb.  All lanes take part in the reduction (rows past M contribute nothing).
      float t1 = 0.f, t2 = 0.f;
      if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f = (n + e < Nout) ? (float)h[e] : 0.f;
          t1 += f;
          t2 += f * f;
        }
      }
      t1 += dpp_mov<0xB1>(t1); t2 += dpp_mov<0xB1>(t2);
      t1 += dpp_mov<0x4E>(t1); t2 += dpp_mov<0x4E>(t2);
      t1 += dpp_mov<0x141>(t1); t2 += dpp_mov<0x141>(t2);
      t1 += dpp_mov<0x140>(t1); t2 += dpp_mov<0x140>(t2);
      if (LPR == 32) {
        t1 += __shfl_xor(t1, 16, 64);
        t2 += __shfl_xor(t2, 16, 64);
      }
      if ((lane % LPR) == 0) {
        stat_out[row][0] = t1;
        stat_out[row][1] = t2;
      }
    }
    if (live && !p.y_bf16) {
      if (vec) {
        stg4(p.Y + (size_t)m * p.ldy + n, y);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < Nout) p.Y[(size_t)m * p.ldy + n + e] = y[e];
      }
    }
    return h;
  };
  // bf16 rows of iterations `it` (ha) and `it + 1` (hb): see above.  wide = the tile's column count and the row pitches allow
  // 16-byte pieces (N a multiple of 8: every shape of the model); else 8 bytes per lane and row as before.
  const bool wide = ((Nout & 7) == 0) && ((p.ldyb & 7) == 0) && (!p.y_bf16 || (p.ldy & 7) == 0);
  auto store_bf16_pair = [&](int it, const bf16x4& ha, const bf16x4& hb) {
    if (p.Yb == nullptr && !p.y_bf16) return;
    const int row_a = (4 * it + wave) * RPI + lane / LPR, row_b = (4 * (it + 1) + wave) * RPI + lane / LPR;
    bf16_t* yb = reinterpret_cast<bf16_t*>(p.Yb);
    bf16_t* y16 = reinterpret_cast<bf16_t*>(p.Y);
    if (wide) {
      const bool odd = (lane & 1) != 0;
      const f32x2 fa = __builtin_bit_cast(f32x2, ha), fb = __builtin_bit_cast(f32x2, hb);
      const f32x2 give = odd ? fa : fb;
      const f32x2 recv = f32x2{dpp_mov<0xB1>(give[0]), dpp_mov<0xB1>(give[1])};      // from the neighbour lane (quad_perm [1,0,3,2])
      const f32x4 out = odd ? f32x4{recv[0], recv[1], fb[0], fb[1]} : f32x4{fa[0], fa[1], recv[0], recv[1]};
      const int m = m0 + (odd ? row_b : row_a), nn = odd ? n - 4 : n;
      if (m < p.M && nn < Nout) {
        if (yb != nullptr) *reinterpret_cast<f32x4*>(yb + (size_t)m * p.ldyb + nn) = out;
        if (p.y_bf16) *reinterpret_cast<f32x4*>(y16 + (size_t)m * p.ldy + nn) = out;
      }
    } else {
      if (n < Nout) {
        if (m0 + row_a < p.M) {
          if (yb != nullptr) *reinterpret_cast<bf16x4*>(yb + (size_t)(m0 + row_a) * p.ldyb + n) = ha;
          if (p.y_bf16) *reinterpret_cast<bf16x4*>(y16 + (size_t)(m0 + row_a) * p.ldy + n) = ha;
        }
        if (m0 + row_b < p.M) {
          if (yb != nullptr) *reinterpret_cast<bf16x4*>(yb + (size_t)(m0 + row_b) * p.ldyb + n) = hb;
          if (p.y_bf16) *reinterpret_cast<bf16x4*>(y16 + (size_t)(m0 + row_b) * p.ldy + n) = hb;
        }
      }
    }
  };
  auto gather = [&](int it, f32x4& v0, f32x4& v1, bool& pad, float& mean, float& rstd) {
    const int row = (4 * it + wave) * RPI + lane / LPR;
    pad = rowpad[row] != 0;
    mean = 0.f;
    rstd = 1.f;
    if (LN) {
      mean = stats[row][0];
      rstd = stats[row][1];
    }
    v0 = *reinterpret_cast<const f32x4*>(Cs + row * kCLd + c4);
    v1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (GLU) v1 = *reinterpret_cast<const f32x4*>(Cs + row * kCLd + DBN / 2 + c4);
  };
  if (fast) {
    f32x4 v0s[IT], v1s[IT];
    float means[IT], rstds[IT];
    bool pads[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) gather(it, v0s[it], v1s[it], pads[it], means[it], rstds[it]);
#pragma unroll
    for (int it = 0; it < IT; it += 2) {
      const bf16x4 ha = sweep_row(it, res[it], true, v0s[it], v1s[it], pads[it], means[it], rstds[it]);
      const bf16x4 hb = sweep_row(it + 1, res[it + 1], true, v0s[it + 1], v1s[it + 1], pads[it + 1], means[it + 1], rstds[it + 1]);
      store_bf16_pair(it, ha, hb);
    }
  } else {
    for (int it = 0; it < IT; ++it) {               // fp32 output only (the launcher rejects bf16 outputs with N % 4 != 0)
      f32x4 rv = f32x4{0.f, 0.f, 0.f, 0.f};
      const int m = min(m0 + (4 * it + wave) * RPI + lane / LPR, p.M - 1);
      if (p.resid)
#pragma unroll
        for (int e = 0; e < 4; ++e) rv[e] = p.resid[(size_t)m * p.ldr + min(n + e, Nout - 1)];
      f32x4 v0, v1;
      float mean, rstd;
      bool pad;
      gather(it, v0, v1, pad, mean, rstd);
      sweep_row(it, rv, false, v0, v1, pad, mean, rstd);
    }
  }
  if (p.Yb_stats != nullptr) {                      // the tile's row statistics: 8 bytes per row, two wave-instructions
    __syncthreads();
    if (tid < DBM && m0 + tid < p.M)
      *reinterpret_cast<f32x2*>(p.Yb_stats + ((size_t)(m0 + tid) * p.n_tiles + n_tile) * 2) = f32x2{stat_out[tid][0], stat_out[tid][1]};
  }
  M3_DIAG(dg[5] = __builtin_amdgcn_s_memtime();
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          dg[6] = __builtin_amdgcn_s_memtime();
          if (lane == 0 && wave == 0 && blockIdx.x < 4096) {
            unsigned long long* o = g_dma_dbg + (size_t)blockIdx.x * 8;
            for (int i = 0; i < 7; ++i) o[i] = dg[i];
            o[7] = __builtin_amdgcn_s_memrealtime();
          })
}

int init_gemm_bf16_dma_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
#define M3_DMA_ATTR(G_, L_)                                                                                                     \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_dma_kernel<G_, L_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, dma_lds_bytes(2))); \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_dma_kernel<G_, L_, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, dma_lds_bytes(4)))
  M3_DMA_ATTR(false, false); M3_DMA_ATTR(false, true); M3_DMA_ATTR(true, false); M3_DMA_ATTR(true, true);
#undef M3_DMA_ATTR
  once.mark();
  return 0;
}

// plain row-major bf16 A (no implicit conv, no concat, no grouping), K a multiple of the k-step, folded LayerNorm only with
// the producer's row statistics at hand
bool gemm_bf16_dma_supports(const GemmParams& p) {
  if (!p.w_bf16 || !p.a_bf16 || p.mode != GEMM_A_PLAIN || p.grp_acc != nullptr || p.w_scale != nullptr) return false;
  if ((p.K % DBK) != 0 || (p.lda & 7) != 0 || p.ln_gamma != nullptr) return false;
  if (p.ln_wsum != nullptr && p.ln_stats == nullptr) return false;
  if (((size_t)(p.M - 1) * p.lda + p.K) * 2 >= ((size_t)1 << 32) || (size_t)p.N * p.K * 2 >= ((size_t)1 << 32)) return false;
  return true;
}

int gemm_bf16_dma_col_tiles(const GemmParams& p) { return cdiv(p.act == ACT_GLU ? p.N / 2 : p.N, p.act == ACT_GLU ? DBN / 2 : DBN); }

int launch_gemm_bf16_dma(const GemmParams& pin, hipStream_t stream) {
  GemmParams p = pin;
  M3_REQUIRE(gemm_bf16_dma_supports(p), "gemm_bf16_dma: unsupported problem (bf16 A and W, plain mode, K %% 64 == 0)");
  if (int rc = init_gemm_bf16_dma_kernels()) return rc;
  const bool glu = p.act == ACT_GLU, ln = p.ln_wsum != nullptr;
  const int Nout = glu ? p.N / 2 : p.N;
  M3_REQUIRE(!(ln && p.mask_in) || p.ln_wbeta, "gemm_bf16_dma: folded LayerNorm + input mask needs ln_wbeta");
  if (p.mask_in || p.mask_out) M3_REQUIRE(p.row_len && p.rows_per_batch > 0, "gemm_bf16_dma: mask needs row_len");
  if (p.y_bf16 || p.Yb) M3_REQUIRE((Nout & 3) == 0 && (p.ldy & 3) == 0 && (p.ldyb & 3) == 0, "gemm_bf16_dma: bf16 output needs N %% 4 == 0");
  p.m_tiles = cdiv(p.M, DBM);
  p.n_tiles = gemm_bf16_dma_col_tiles(p);
  dim3 grid(cdiv(p.m_tiles, 8) * 8 * p.n_tiles);
  // work-groups that do not outnumber the CUs: one per CU with a 4-stage ring; else two per CU with 2 stages each
  const int cus = device_cu_count();
  const bool deep = (long)p.m_tiles * p.n_tiles <= cus && p.K / DBK >= 4;
#define M3_DMA_LAUNCH(G_, L_)                                                                                              \
  do {                                                                                                                     \
    if (deep) hipLaunchKernelGGL((gemm_bf16_dma_kernel<G_, L_, 4>), grid, dim3(256), dma_lds_bytes(4), stream, p);         \
    else hipLaunchKernelGGL((gemm_bf16_dma_kernel<G_, L_, 2>), grid, dim3(256), dma_lds_bytes(2), stream, p);              \
  } while (0)
  if (glu && ln) M3_DMA_LAUNCH(true, true);
  else if (glu) M3_DMA_LAUNCH(true, false);
  else if (ln) M3_DMA_LAUNCH(false, true);
  else M3_DMA_LAUNCH(false, false);
#undef M3_DMA_LAUNCH
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3

#ifdef M3_DMA_DIAG
extern "C" int m3_debug_dma_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(m3::g_dma_dbg), bytes < sizeof(m3::g_dma_dbg) ? bytes : sizeof(m3::g_dma_dbg));
}
#endif
