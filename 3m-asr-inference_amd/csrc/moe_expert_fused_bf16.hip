// Grouped expert FFN, bf16, ONE kernel, H never leaves the registers.
//
// Replaces, for long batches, the reference's per-expert host loop of cuBLAS GEMM + BiasSilu + GEMM + Bias launches
// (TRTAPI++/plugin/fmoe_expert_plugin/fmoe_expert_plugin.cpp:82-128, fmoe_expert_kernel.cu:130-189) and this repo's own
// two-GEMM grouped form (gemm_bf16_tiled.hip), which writes H (S x F bf16) to memory and reads it back: with D = 512,
// F = 1024 that form sits below the bf16 ridge even at infinite tokens.
//
// Formulation: everything is computed TRANSPOSED, tokens on the MFMA column (lane) axis, weights as the A operand:
//     Ht[f, tok] = SiLU( W1[e][f, :] . Xt[:, tok] + b1[e][f] )          v_mfma_f32_32x32x16_bf16, A = W1 rows, B = Xt
//     Yt[d, tok] = W2[e][d, :] . Ht[:, tok]                              A = W2 rows, B = Ht
// A 32 x 32 accumulator tile has its column on the lane and its rows in the registers, which is exactly the B-operand
// layout of the next product once its rows are paired into bf16 (cdna_hip_programming.md 3, "An accumulator tile as the
// next MFMA's operand"): Ht goes from GEMM-1's accumulator to GEMM-2's operand through v_cvt_pk_bf16_f32 alone -- no
// LDS, no memory.  The k order of that operand is a fixed permutation of the accumulator's row order; it is undone for
// free by fetching W1's rows in the inverse order (pi below), so W2 is consumed in its natural order.
//
// Work-group = 4 waves, one per SIMD, each owning 32 tokens of one expert (128 tokens per work-group) for the whole
// kernel: its X fragments (32 k-steps x 4 VGPRs) and its 512 x 32 output tile (16 accumulators of 16 registers, AGPRs)
// stay in registers; only the WEIGHTS move: each 64-wide slice of F is 64 KB of W1 rows + 64 KB of W2 (the plan's
// slice-major w_2), streamed through an 8-slot x 16 KB LDS ring by LDS-DMA (global_load_lds, 16 B per lane, source
// addresses pre-swizzled so that the ds_read_b128 fragment reads are bank-conflict-free), seven pieces ahead of the
// MFMAs, one s_barrier + one counted s_waitcnt vmcnt per 16 MFMAs.  Per token tile every weight byte is read once from L2
// (128 FLOP per byte), four work-groups of an expert sit on the same XCD.
//
// Output: sorted rows Y[row][d] (fp32) in `ybuf`, FSPLIT partial slabs when the F range is split over FSPLIT work-groups
// (more work-groups for short batches); b2 / gate / residual / LayerNorm / un-permute are moe_combine_kernel's.
// Numerics: operands rounded to bf16 (RNE) at the MFMA inputs, fp32 accumulation over k in MFMA order, SiLU in fp32 --
// the arithmetic tests/test_bf16_gpu.py::test_fmoe_expert_bf16 states.  A row's result does not depend on the other rows'
// values; its fp32 summation order over the F slices depends on its tile index (see "phase0" below).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace m3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kTok = 128;            // tokens per work-group
constexpr int kPiece = 32768;        // bytes per ring slot
constexpr int kRing = 4;             // slots = pieces per 64-wide F slice (2 of W1: 32 full rows each; 2 of W2: 256 rows x 64 f)
constexpr int kD = 512;              // model width this kernel is built for (32 k-steps, 16 output blocks)

// MFMA wrappers.  M3_FUSED_ASM_MFMA = 1 pins the two accumulator sets to register classes by hand (GEMM-1's tile in arch
// VGPRs, the 512 x 32 output tile in a0..a255) with inline-asm MFMAs; the default is the builtin, scheduled by hipcc.
#ifndef M3_FUSED_ASM_MFMA
#define M3_FUSED_ASM_MFMA 0
#endif
#if M3_FUSED_ASM_MFMA
__device__ __forceinline__ void mfma32_v(f32x16& c, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_a(f32x16& c, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_v0(f32x16& c, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain_v(f32x16& c) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c)); }
__device__ __forceinline__ void mfma_drain_a(f32x16& c) { asm volatile("s_nop 15\n\ts_nop 3" : "+a"(c)); }
#elif defined(M3_FUSED_NO_MFMA)
__device__ __forceinline__ void mfma32_v(f32x16& c, bf16x8 a, bf16x8 b) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma32_a(f32x16& c, bf16x8 a, bf16x8 b) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma32_v0(f32x16& c, bf16x8 a, bf16x8 b) {
#pragma unroll
  for (int j = 0; j < 16; ++j) c[j] = 0.f;
  asm volatile("" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain_v(f32x16&) {}
__device__ __forceinline__ void mfma_drain_a(f32x16&) {}
#else
__device__ __forceinline__ void mfma32_v(f32x16& c, bf16x8 a, bf16x8 b) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void mfma32_a(f32x16& c, bf16x8 a, bf16x8 b) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void mfma32_v0(f32x16& c, bf16x8 a, bf16x8 b) {
  f32x16 z;
#pragma unroll
  for (int j = 0; j < 16; ++j) z[j] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ void mfma_drain_v(f32x16&) {}
__device__ __forceinline__ void mfma_drain_a(f32x16&) {}
#endif

// diagnostic build (-DM3_FUSED_DIAG): per wave, shader-clock cycles spent in the counted vmcnt wait, in the barrier and in
// total; read back with m3_debug_fused_read.  No stamp executes in the product build.
// ablation switches for timing-only diagnostic builds (results are wrong by construction):
//   M3_FUSED_NO_READS  fragments are not read from LDS (one read per step keeps the address math alive)
//   M3_FUSED_NO_MFMA   no matrix instructions (fragments kept live)
//   M3_FUSED_NO_FILL   no LDS-DMA inside the loop
#ifdef M3_FUSED_DIAG
__device__ unsigned long long g_fused_dbg[4096 * 4];
#define M3_DIAG(x) x
#else
#define M3_DIAG(x)
#endif

// LDS-DMA: 64 lanes x 16 B land at lds + lane * 16 (wave-uniform lds); lane source = buffer base + voff + soff
// (buffer_load_dwordx4 ... offen lds: one VGPR byte offset per lane, the piece's base in an SGPR, no address arithmetic)
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// row order in which W1's rows are fetched into a 32-row LDS block: swaps the two middle quads of every 16 rows, the
// inverse of the k permutation a 32x32 accumulator carries when it is re-used as a 32x32x16 B operand
__device__ __forceinline__ int pi_row(int r) {
  const int q = (r >> 2) & 3;
  const int qp = (q == 1) ? 2 : (q == 2 ? 1 : q);
  return (r & ~12) | (qp << 2);
}

}  // namespace

template <int FSPLIT>
__global__ __launch_bounds__(256) void expert_ffn_fused_bf16_kernel(
    const float* __restrict__ x, int ldx, const int32_t* __restrict__ pos, const int32_t* __restrict__ acc_hist, int S, int E,
    int F, const bf16_t* __restrict__ w1, const float* __restrict__ b1, const bf16_t* __restrict__ w2, int w2_row_stride,
    int w2_slice_stride, float* __restrict__ ybuf, int nblk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // ring [8][16 KB] | b1 of this expert's F range (fp32)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;

  // ---- work-group -> (token tile, F part): consecutive logical ids share an XCD, tiles of one expert are consecutive ----
  const int per = nblk >> 3;
  const int logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int tile = logical / FSPLIT, fs = logical - tile * FSPLIT;
  int e = -1, tt = 0;
  {
    int base = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int ee = e0 + lane;
      const int cnt = ee < E ? acc_hist[ee + 1] - acc_hist[ee] : 0;
      const int nt = (cnt + kTok - 1) / kTok;
      int incl = nt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
      }
      const int excl = base + incl - nt;
      const unsigned long long m = __ballot(tile >= excl && tile < excl + nt);
      if (m) {
        const int src = __ffsll((long long)m) - 1;
        e = e0 + src;
        tt = tile - __shfl(excl, src, 64);
        break;
      }
      base += __shfl(incl, 63, 64);
    }
  }
  if (e < 0) return;                                   // beyond the last tile (uniform over the work-group)
  // make the uniformity of (expert, tile) provable: everything derived from them (buffer descriptors, piece offsets) must
  // live in SGPRs, or hipcc wraps every buffer load in a waterfall loop (cdna_hip_programming.md T20)
  e = __builtin_amdgcn_readfirstlane(e);
  tt = __builtin_amdgcn_readfirstlane(tt);
  const int row_begin = acc_hist[e], row_end = acc_hist[e + 1];
  const int my_row = row_begin + tt * kTok + wv * 32 + r;         // sorted row of this lane's token
  const bool live = my_row < row_end;
  const int nsl = F / (64 * FSPLIT);                   // 64-wide slices of F this work-group contracts
  const int sl0 = fs * nsl;
  // The token tiles of one expert run side by side on one XCD and would stream the SAME weight bytes in lockstep: every
  // line is then wanted by all of them at the same moment and they queue on one L2 channel (measured: 1.35-1.75x the
  // time).  So tile tt starts its walk over the F slices at slice 5 tt (mod nsl) and wraps around.  Consequence for the
  // numerics: the fp32 accumulation order over the slices depends on the token's tile index within its expert -- results
  // are reproducible run to run and do not depend on other tokens' values, but are equal across different batches only
  // to fp32 summation-order rounding, not bit for bit (the slab and two-GEMM forms are).
#ifdef M3_FUSED_NO_DEPHASE
  const int phase0 = 0;
#else
  const int phase0 = (tt * 5) % nsl;
#endif
  auto abs_slice = [&](int rel) { const int v = rel + phase0; return sl0 + (v >= nsl ? v - nsl : v); };

  float* bias_lds = reinterpret_cast<float*>(smem + kRing * kPiece);
  for (int i = threadIdx.x * 4; i < nsl * 64; i += 1024)
    *reinterpret_cast<f32x4*>(bias_lds + i) = ldg4(b1 + (size_t)e * F + sl0 * 64 + i);

  // ---- X fragments of this lane's token: Xt[k = 16 s + 8 h + j][tok] = x[pos[row]][16 s + 8 h + j] ----
  bf16x8 xf[kD / 16];
  {
    const float* xr = x + (size_t)pos[live ? my_row : row_end - 1] * ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < kD / 16; ++s) xf[s] = cvt8(ldg4(xr + 16 * s), ldg4(xr + 16 * s + 4));
  }

  // ---- per-lane constants of the LDS-DMA fills (wave wv issues instructions 8 wv .. 8 wv + 7 of every piece) ----
  // Pieces are CONTIGUOUS 32 KB of memory (all L2 channels busy): a W1 piece = 32 full rows of 1 KB (block fb of a slice),
  // a W2 piece = 256 rows x 128 B of the slice-major layout.
  // W1: instruction i carries row i; lane l -> physical 16-B chunk p = l holds logical chunk c = (p & 48) | ((p ^ i) & 15)
  // W2: instruction i carries rows 8 i .. 8 i + 7; lane l -> row 8 i + (l >> 3), physical chunk p = l & 7 holds logical
  //     chunk c = p ^ ((row >> 1) & 7)
  // (byte offsets, unsigned: a uniform 64-bit base + a zero-extended 32-bit lane offset is one SGPR pair + one VGPR)
  unsigned w1_off[8], w2_off[8];
#pragma unroll
  for (int ii = 0; ii < 8; ++ii) {
    const int i = 8 * wv + ii;
    const int c1 = (lane & 48) | ((lane ^ i) & 15);
    w1_off[ii] = (unsigned)(pi_row(i) * kD + 8 * c1) * 2u;
    const int row2 = 8 * i + (lane >> 3), p2 = lane & 7;
    const int c2 = p2 ^ ((row2 >> 1) & 7);
    w2_off[ii] = (unsigned)(row2 * w2_row_stride + 8 * c2) * 2u;
  }
  const unsigned wbytes = (unsigned)F * kD * 2u;      // both layouts hold D * F elements per expert
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(w1 + (size_t)e * F * kD), 0, (int)wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(w2 + (size_t)e * F * kD), 0, (int)wbytes, 0x00020000);

  // piece q (0..3) of slice sl (absolute 64-wide slice index) -> ring slot q
  auto issue = [&](int sl, int q) {
    char* dst = smem + q * kPiece + wv * 8192;
    if (q < 2) {
      const unsigned base = (unsigned)((sl * 64 + q * 32) * kD) * 2u;
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) blds16(rs1, w1_off[ii], base, dst + ii * 1024);
    } else {
      const unsigned base = (unsigned)(sl * w2_slice_stride + (q - 2) * 256 * w2_row_stride) * 2u;
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) blds16(rs2, w2_off[ii], base, dst + ii * 1024);
    }
  };

  // ---- fragment read addresses (bank-conflict-free with the swizzles above) ----
  const int rd1 = (r << 10) | ((h ^ (r & 15)) << 4);           // W1 piece: k-step s (0..31) reads byte (rd1 ^ (s << 5))
  const int rd2 = (r << 7) | ((h ^ ((r >> 1) & 7)) << 4);      // W2 piece: block dbl, k-step s4 reads (rd2 ^ (s4 << 5)) + dbl * 4096

  f32x16 accy[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) accy[i][j] = 0.f;

  __syncthreads();                                     // bias in LDS (nothing of the ring is in flight yet)
  // prologue: pieces 0..2 of the first slice
#pragma unroll
  for (int q = 0; q < kRing - 1; ++q) issue(abs_slice(0), q);

#ifdef M3_FUSED_NO_READS
  bf16x8 a_const = xf[0];
  auto RD = [&](const char* p) -> bf16x8 { asm volatile("" : "+v"(a_const)); return a_const; };
#else
  auto RD = [&](const char* p) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(p); };
#endif
  f32x16 acc1;
  bf16x8 hf[4];                                        // Ht fragments of the current slice: k-steps of 16 f

  auto silu_pack = [&](int fb, int sl_rel) {
    // acc1 rows (i & 3) + 8 (i >> 2) + 4 h of block fb hold f = 64 sl + 32 fb + pi(row); bias of 4 consecutive f per quad
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row0 = 8 * m + 4 * h;                  // rows row0 .. row0 + 3 <-> registers 4 m .. 4 m + 3
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias_lds + sl_rel * 64 + fb * 32 + pi_row(row0));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = acc1[4 * m + j] + b[j];
        acc1[4 * m + j] = v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 t;
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = (bf16_t)acc1[8 * s + j];
      hf[2 * fb + s] = t;
    }
  };

  M3_DIAG(unsigned long long t_wait = 0; unsigned long long t_bar = 0; const unsigned long long t_begin = __builtin_amdgcn_s_memtime();)
  for (int sl = 0; sl < nsl; ++sl) {
    const bool last = sl == nsl - 1;
#pragma unroll
    for (int q = 0; q < kRing; ++q) {
      // piece q of this slice has landed: this wave's 8 fills of it (counted: the fills issued after it stay in flight),
      // then the barrier for the other waves' fills
      M3_DIAG(const unsigned long long d0 = __builtin_amdgcn_s_memtime();)
      if (!last || q < 2) {
        wait_vmcnt<16>();
      } else {
        if (q == 2) wait_vmcnt<8>();
        if (q == 3) wait_vmcnt<0>();
      }
      M3_DIAG(const unsigned long long d1 = __builtin_amdgcn_s_memtime();)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      M3_DIAG(const unsigned long long d2 = __builtin_amdgcn_s_memtime(); t_wait += d1 - d0; t_bar += d2 - d1;)
      // refill the slot consumed one step ago with the piece 3 ahead (every wave is past that step's reads)
      {
        const int qn = (q + kRing - 1) & (kRing - 1);
        const int sln = q == 0 ? sl : sl + 1;
#ifndef M3_FUSED_NO_FILL
        if (sln < nsl) issue(abs_slice(sln), qn);
#endif
      }
      const char* slot = smem + q * kPiece;
      if (q < 2) {                                     // GEMM-1: block fb = q, all 32 k-steps
        int rb = rd1;
        asm volatile("" : "+v"(rb));                   // the fragment addresses are formed here, not kept across steps
        // fragments in batches of 4 (16 VGPRs), the next batch's reads issued ahead of this batch's MFMAs
        bf16x8 a[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[0][j] = RD(slot + (rb ^ (j << 5)));
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          if (b < 7) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[(b + 1) & 1][j] = RD(slot + (rb ^ ((4 * b + 4 + j) << 5)));
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (b == 0 && j == 0) mfma32_v0(acc1, a[0][0], xf[0]);
            else mfma32_v(acc1, a[b & 1][j], xf[4 * b + j]);
          }
        }
        mfma_drain_v(acc1);
        silu_pack(q, abs_slice(sl) - sl0);
      } else {                                         // GEMM-2: output blocks 8 (q - 2) .. + 7, 4 k-steps of 16 f
        int rb = rd2;
        asm volatile("" : "+v"(rb));
        bf16x8 a[2][4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) a[0][s4] = RD(slot + (rb ^ (s4 << 5)));
#pragma unroll
        for (int dbl = 0; dbl < 8; ++dbl) {
          if (dbl < 7) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
              a[(dbl + 1) & 1][s4] = RD(slot + (dbl + 1) * 4096 + (rb ^ (s4 << 5)));
          }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) mfma32_a(accy[(q - 2) * 8 + dbl], a[dbl & 1][s4], hf[s4]);
        }
      }
    }
  }

  M3_DIAG(if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long* o = g_fused_dbg + (blockIdx.x * 4 + wv) * 4;
    o[0] = t_wait; o[1] = t_bar; o[2] = __builtin_amdgcn_s_memtime() - t_begin; o[3] = (unsigned long long)tile;
  })
#pragma unroll
  for (int db = 12; db < 16; ++db) mfma_drain_a(accy[db]);   // the last step's MFMAs
  if (live) {
    float* yr = ybuf + ((size_t)fs * S + my_row) * kD + 4 * h;
#pragma unroll
    for (int db = 0; db < 16; ++db)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        stg4(yr + 32 * db + 8 * m, f32x4{accy[db][4 * m], accy[db][4 * m + 1], accy[db][4 * m + 2], accy[db][4 * m + 3]});
  }
}

// ---- host side ----
static int fused_min_rows() {   // read on every call (a getenv per launch is noise): tests lower it per module
  const char* e = getenv("M3_EXPERT_FUSED_MIN_ROWS");
  return e ? atoi(e) : 32768;     // measured cross-over against the two-GEMM form: 16 k rows lose, 64 k win
}

// the F range is split over 2 work-groups while the token tiles alone would leave CUs idle
int expert_ffn_fused_bf16_fsplit(int S, int E, int D, int F) {
  const int tiles = cdiv(S, kTok) + E / 2;
  return (tiles < 224 && F % 128 == 0) ? 2 : 1;
}

bool expert_ffn_fused_bf16_applies(int S, int E, int D, int F) {
  // 128-token tiles per expert: below ~64 rows per expert they are mostly padding and the 64-row two-GEMM form wins
  return D == kD && F % 64 == 0 && F <= 4096 && S >= fused_min_rows() && S >= 64 * E && E <= 1024;
}

int init_expert_ffn_fused_bf16_kernels() {
  static bool done = false;
  if (done) return 0;
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  done = true;
  return 0;
}

int launch_expert_ffn_fused_bf16(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D,
                                 int F, const void* w1, const float* b1, const void* w2, int w2_sliced, float* ybuf,
                                 hipStream_t stream) {
  M3_REQUIRE(expert_ffn_fused_bf16_applies(S, E, D, F), "expert_ffn_fused_bf16: shape S=%d E=%d D=%d F=%d not supported", S, E, D, F);
  M3_REQUIRE((ldx & 3) == 0, "expert_ffn_fused_bf16: ldx=%d must be a multiple of 4", ldx);
  if (int rc = init_expert_ffn_fused_bf16_kernels()) return rc;
  const int fsplit = expert_ffn_fused_bf16_fsplit(S, E, D, F);
  const int tiles = cdiv(S, kTok) + E;                       // >= sum_e ceil(cnt_e / 128)
  const int nblk = cdiv(tiles * fsplit, 8) * 8;
  const int row_stride = w2_sliced ? 64 : F;                 // elements between consecutive d rows of W2
  const int slice_stride = w2_sliced ? D * 64 : 64;          // elements between consecutive 64-wide f slices
  const size_t lds = (size_t)kRing * kPiece + (size_t)(F / fsplit) * sizeof(float);
#define M3_FUSED_LAUNCH(FS_)                                                                                          \
  hipLaunchKernelGGL((expert_ffn_fused_bf16_kernel<FS_>), dim3(nblk), dim3(256), lds, stream, x, ldx, pos, acc_hist, \
                     S, E, F, (const bf16_t*)w1, b1, (const bf16_t*)w2, row_stride, slice_stride, ybuf, nblk)
  if (fsplit == 2) M3_FUSED_LAUNCH(2); else M3_FUSED_LAUNCH(1);
#undef M3_FUSED_LAUNCH
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3

#ifdef M3_FUSED_DIAG
extern "C" int m3_debug_fused_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(m3::g_fused_dbg), bytes < sizeof(m3::g_fused_dbg) ? bytes : sizeof(m3::g_fused_dbg));
}
#endif
