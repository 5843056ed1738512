// Grouped expert FFN in fp8 arithmetic (A8W8): e4m3 weights x e4m3 activations on v_mfma_f32_32x32x16_fp8_fp8, fp32
// accumulation, ONE kernel, H never leaves the registers.  This is the path the reference's `--int8` flag names and never
// finished (builder.py:39-49 `assert 0`; fmoe_expert_plugin.cpp:264-266 asserts on anything but fp32): there is no
// reference output for it, the arithmetic is defined here and tested against an fp64 evaluation of exactly this
// quantised computation (tests/test_fp8_gpu.py).
//
// Quantisation (OCP e4m3fn, |max| = 448, conversions round to nearest even and SATURATE to +-448: the kernel sets
// MODE.FP16_OVFL, without which v_cvt_pk_fp8_f32 turns an out-of-range input into NaN on gfx950 -- tools/ubench/fp8_cvt_sat.hip):
//   W1[e][f][:], W2[e][d][:]   e4m3 with one fp32 scale per output row (s1[e][f], s2[e][d]) -- the plan's fp8 format
//   X[tok][:]                  e4m3 with a per-token DYNAMIC scale sx[tok] = amax(row) / 448, computed while the row is loaded
//   H[tok][:]                  e4m3 with ONE static scale per layer, h_scale (from the calibrator: amax of H over the
//                              calibration batches x 1.25 / 448; e4m3 is a floating-point format, so the scale only has to
//                              keep H inside [2^-9, 448] h_scale -- it does not set the precision)
//   z = (W1q . Xq) s1[f] sx[tok] + b1[f];  H = SiLU(z);  Y = (W2q . Hq) s2[d] h_scale      (b2 etc.: moe_combine_kernel)
//
// Structure: the transposed, register-resident formulation first built for bf16 in round 2 (tokens on the MFMA column axis,
// weights as the A operand streamed through LDS, GEMM-1's accumulator re-used as GEMM-2's B operand).
// What fp8 changes: X fragments take 64 VGPRs instead of 128 and weight fragments 2 instead of 4; a 64-wide slice of F is
// 32 KB of W1 + 32 KB of W2, i.e. ONE 32-KB piece each: 64 MFMAs per barrier, half the L2 -> LDS bytes per FLOP.
// Weight staging is REGISTER staging (global_load_dwordx4 -> ds_write_b128, one piece in flight in 32 VGPRs per wave, two
// 32-KB LDS slots), not LDS-DMA: measured on this chip, a CU keeps only ~16 KB of LDS-DMA in flight (throughput = 16 KB /
// latency: 24 GB/s from HBM, 80-108 GB/s from a quiet L2, 13-27 GB/s inside this kernel) and a wave that issues a fill
// beyond that stalls AT THE ISSUE -- with one wave per SIMD that stalled the MFMAs for 55 % of the kernel (in-kernel
// stamps: 434 cycles per fill; tools/ubench/ldsdma_fill.hip).  Ordinary loads are not subject to that limit.
// One ds_read_b128 brings the A operands of TWO k-steps: the k order inside every 32-byte group is (h, step, j) instead of
// (step, h, j) -- applied to both operands, a permutation of the summation index changes nothing but the fp32 summation
// order.  W1's rows are fetched in the order pi8 that makes GEMM-1's accumulator rows land in GEMM-2's k order.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace m3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// diagnostic build (-DM3_FUSED_DIAG): shader-clock cycles per wave spent in the counted wait, the barrier, issuing the
// LDS-DMA fills, GEMM-1 (+ SiLU + quantisation) and GEMM-2; read back with m3_debug_fused8_read.  Not in the product build.
#ifdef M3_FUSED_DIAG
__device__ unsigned long long g_fused8_dbg[4096 * 8];
#define M3_DIAG(...) __VA_ARGS__
#else
#define M3_DIAG(...)
#endif


namespace {

constexpr int kTok = 128;            // tokens per work-group (4 waves x 32)
constexpr int kPiece = 32768;        // bytes per piece = one 64-wide slice of W1 (64 rows x 512 B) or of W2 (512 rows x 64 B)
constexpr int kW1Row = 528;          // W1 rows are padded to 528 B in LDS: conflict-free ds_read_b128 at base + immediate
constexpr int kSlot = 64 * kW1Row;   // bytes per LDS slot (33792)
constexpr int kRing = 2;
constexpr int kD = 512;

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
}
// The same LDS-DMA fill as inline assembly.  hipcc's waitcnt insertion treats a buffer_load ... lds it can see as a pending
// write to ALL of LDS and puts s_waitcnt vmcnt(0) in front of every later ds_write / ds_read of the loop -- which would also
// drain the register-staged weight loads each time.  Hidden in asm, the fill is invisible to that pass: hipcc's own vmcnt
// waits then under-count the operations in flight (they wait for a little more than needed, never less) and the waits for the
// fills themselves are written by hand (wait_vmcnt below).
__device__ __forceinline__ void dma16_hidden(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");   // (m0 is reserved: hipcc re-loads it before its own uses)
}
// Y leaves through stores hipcc does not see either: with loads AND stores pending in its model (gfx9 counts both in vmcnt
// and they may complete out of order), every later wait for a load becomes s_waitcnt vmcnt(0) until a vmcnt(0) is executed
// on all paths -- in the persistent tile loop that turned each wait for a staged weight KB into a full drain.  Nothing ever
// waits for these stores (the end of the kernel does); an unseen store in flight only makes a counted wait for an older load
// conservative, because loads return in order among themselves.
__device__ __forceinline__ void store16_nt_hidden(float* p, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#ifdef M3_F8_NO_MFMA   // ablation build: everything but the matrix instructions (results are garbage)
__device__ __forceinline__ f32x16 mfma8(long a, long b, f32x16 c) {
  asm volatile("" : "+a"(c) : "v"(a), "v"(b));
  return c;
}
#else
__device__ __forceinline__ f32x16 mfma8(long a, long b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
}
#endif
// GEMM-1's two z blocks must live in ARCH VGPRs (SiLU reads them with VALU instructions); hipcc gives every MFMA result an
// AGPR home and, with all 256 AGPRs taken by the output tile, parks two output blocks in VGPRs around GEMM-1 and copies the z
// blocks out again (~250 v_accvgpr moves per slice).  Inline assembly pins the register class.  Hazards: dependent MFMAs
// on the same accumulator issue back to back (hipcc does the same); hipcc does not know that the asm result comes from
// the matrix pipe and may schedule a VALU read of it right behind the last MFMA, so every chain ends with mfma8_v_settle
// (24 wait states >= the 19 a 16-pass MFMA needs before a VALU read); operands come from ds_read (hipcc waits on
// lgkmcnt for asm inputs) or are long-lived.
#ifndef M3_F8_ASM_MFMA
#define M3_F8_ASM_MFMA 1
#endif
#ifdef M3_F8_NO_MFMA
__device__ __forceinline__ void mfma8_v0(f32x16& c, long a, long b) {
#pragma unroll
  for (int j = 0; j < 16; ++j) c[j] = 0.f;
  asm volatile("" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma8_v(f32x16& c, long a, long b) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); }
#elif M3_F8_ASM_MFMA
__device__ __forceinline__ void mfma8_v0(f32x16& c, long a, long b) {
  asm("v_mfma_f32_32x32x16_fp8_fp8 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma8_v(f32x16& c, long a, long b) {
  asm("v_mfma_f32_32x32x16_fp8_fp8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#else
__device__ __forceinline__ void mfma8_v0(f32x16& c, long a, long b) {
  f32x16 z;
#pragma unroll
  for (int j = 0; j < 16; ++j) z[j] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ void mfma8_v(f32x16& c, long a, long b) { c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0); }
#endif
__device__ __forceinline__ void mfma8_v_settle(f32x16& c) { asm("s_nop 15\n\ts_nop 7" : "+v"(c)); }
// max of non-negative values over the 64 lanes, one instruction per step: v_max_f32 with a DPP source (hipcc's fmaxf costs
// three: v_mov_dpp, a canonicalising v_max, v_max), rows of 16 first, then row_bcast 15 / 31 carry the row maxima to lane 63
// and v_readlane brings the total back as a scalar.  (s_nop 1: a DPP source written by the previous VALU instruction needs
// two wait states, which hipcc does not insert inside asm.)
__device__ __forceinline__ float wave_amax_dpp(float v) {
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float clamp448(float v) { return __builtin_amdgcn_fmed3f(v, -448.f, 448.f); }
// 8 floats -> 8 e4m3 bytes (element j in byte j)
__device__ __forceinline__ long q8(const float* v, float inv) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[0] * inv), clamp448(v[1] * inv), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[2] * inv), clamp448(v[3] * inv), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[4] * inv), clamp448(v[5] * inv), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[6] * inv), clamp448(v[7] * inv), hi, true);
  return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// the same for values that are already in units of the scale
__device__ __forceinline__ long q8s(const float* v) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);       // (saturating: MODE.FP16_OVFL is set)
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// LDS row rho (0..31) of a W1 block holds the block's row pi8(rho): rho = 16 s + 8 a + 4 h + b  ->  16 h + 8 s + 4 a + b
__device__ __forceinline__ int pi8_row(int rho) {
  return (((rho >> 2) & 1) << 4) | (((rho >> 4) & 1) << 3) | (((rho >> 3) & 1) << 2) | (rho & 3);
}

}  // namespace

struct TileRef {
  int e, tt, fs;
};
// item -> (expert, token tile of the expert, F part), from the histogram alone (one wave scan per 64 experts); uniform
__device__ __forceinline__ TileRef find_tile(const int32_t* __restrict__ acc_hist, int E, int item, int fsplit, int lane) {
  const int tile = item / fsplit;
  int e = -1, tt = 0, base = 0;
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
  return TileRef{__builtin_amdgcn_readfirstlane(e), __builtin_amdgcn_readfirstlane(tt), item - tile * fsplit};
}

// LDS map (160 KB, all of it): two 33-KB weight slots (slot 1 doubles as the Y staging area between two tiles) | 8 KB b1, s1 of
// the current tile's F range (<= 1024 floats each) | 4 KB s2 * h_scale of the current / next tile's expert | 66 KB four
// wave-private X images [32 rows][528 B] (e4m3) | 16 KB four wave-private raw rows x 2 (the next tile's X arriving by LDS-DMA)
constexpr int kOffBias = kRing * kSlot;
constexpr int kOffS2 = kOffBias + 8192;
constexpr int kOffImg = kOffS2 + 4096;
constexpr int kImgRow = 528;
constexpr int kOffRaw = kOffImg + 4 * 32 * kImgRow;
constexpr int kLdsBytes = kOffRaw + 4 * 4096;

// XQ: the input rows arrive ALREADY quantised (e4m3 [S][512] + one fp32 scale per row, written by moe_router_kernel with the
// arithmetic of quant_row below): a tile's X is 64 KB instead of 256 KB of fp32 rows and nothing is converted here.  With one
// work item per work-group (configs[4]'s share: ~140 items on 256 CUs) the 256-KB prologue was 20 k of an item's ~90 k cycles.
template <int FSPLIT, bool XQ = false>
__global__ __launch_bounds__(256) void expert_ffn_fused_fp8_kernel(
    const float* __restrict__ x, int ldx, const int32_t* __restrict__ pos, const int32_t* __restrict__ acc_hist, int S, int E,
    int F, const unsigned char* __restrict__ w1, const float* __restrict__ s1, const float* __restrict__ b1,
    const unsigned char* __restrict__ w2, const float* __restrict__ s2, int w2_row_stride, int w2_slice_stride, float h_scale,
    float* __restrict__ ybuf, const unsigned char* __restrict__ xq_in, const float* __restrict__ xq_scale,
    int32_t* __restrict__ fs_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);         // hwreg(HW_REG_MODE, 23, 1): FP16_OVFL -- fp8 conversions saturate
  // The kernel runs at the register limit (256 accumulator + 256 other registers, one wave per SIMD).  Per-lane address
  // constants that live across the whole tile loop get spilled by hipcc -- and a scratch re-load inside the loop costs an
  // s_waitcnt vmcnt(0), i.e. a drain of the weight loads in flight.  So the hot loop RECOMPUTES them from the lane id at
  // every use (a handful of VALU operations in the MFMAs' shadow); the opaque copy keeps hipcc from hoisting them back out.
  auto olane = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };

  // ---- PERSISTENT work-groups, one per CU.  The number of real tiles T is only known on the device (it depends on the
  //      routing): every work-group sums it from the histogram; XCD x (= blockIdx % 8: work-groups b and b + 8 share an XCD
  //      and its L2) owns the consecutive items [x * per, (x + 1) * per), per = ceil(T * FSPLIT / 8) -- the tiles of one
  //      expert stay on one XCD and all XCDs get the same number of tiles -- and its work-groups walk them with stride
  //      gridDim / 8, so the tiles in flight on an XCD at any time are neighbours (same experts) ----
  int total_tiles = 0;
  for (int e0 = 0; e0 < E; e0 += 64) {
    const int ee = e0 + lane;
    int nt = ee < E ? (acc_hist[ee + 1] - acc_hist[ee] + kTok - 1) / kTok : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nt += __shfl_xor(nt, d, 64);
    total_tiles += nt;
  }
  // F parts per token tile.  FSPLIT is the host's choice from the PADDED row count; fs_out != null (all experts local): the
  // tile count is known here, and when it leaves CUs without an item the split is made finer (every work-group derives the same
  // value; it is left in *fs_out for the combine, which sums that many slabs): configs[4]'s share has ~50-64 tiles -- 100-128
  // items of 832 KB at FSPLIT = 2 on 256 CUs, 200-256 items of 576 KB at 4
  int fsplit = FSPLIT;
  {
    const int tt_ = __builtin_amdgcn_readfirstlane(total_tiles);
    if (fs_out != nullptr) {
      if (tt_ * 4 <= (int)gridDim.x && F % 512 == 0) fsplit = 4;
      else if (tt_ * 2 <= (int)gridDim.x && F % 256 == 0 && FSPLIT < 2) fsplit = 2;
      if (blockIdx.x == 0 && threadIdx.x == 0) *fs_out = fsplit;
    }
  }
  const int items = __builtin_amdgcn_readfirstlane(total_tiles) * fsplit;
  const int per = (items + 7) >> 3;
  const int stride = gridDim.x >> 3;
  const int item_end = min(((int)(blockIdx.x & 7) + 1) * per, items);
  int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (item >= item_end) return;                        // (uniform over the work-group)
  M3_DIAG(const unsigned long long t_k0 = __builtin_amdgcn_s_memtime();)

  const int nsl = F / (64 * fsplit);                   // 64-wide slices of F a work item contracts
  const int np = 2 * nsl;                              // pieces (= steps) per work item
  float* b1_lds = reinterpret_cast<float*>(smem + kOffBias);
  float* s1_lds = b1_lds + 1024;
  float* s2h_lds = reinterpret_cast<float*>(smem + kOffS2);        // [2][512]: parity of the work-group's tile counter
  char* img = smem + kOffImg + wv * (32 * kImgRow);
  char* rawb = smem + kOffRaw + wv * 4096;

  // ---- the current work item (everything below is wave-uniform and lives in SGPRs) ----
  TileRef cur = find_tile(acc_hist, E, item, fsplit, lane);
  if (cur.e < 0) return;
  int row_end = acc_hist[cur.e + 1];
  int tile_row0 = acc_hist[cur.e] + cur.tt * kTok + wv * 32;
  int sl0 = cur.fs * nsl;
  int phase0 = (cur.tt * 5) % nsl;                     // de-phased walk over the slices (co-resident tiles must not ask L2 for the same lines at the same moment)
  const unsigned char* w1e = w1 + (size_t)cur.e * F * kD;
  const unsigned char* w2e = w2 + (size_t)cur.e * F * kD;
  auto abs_slice = [&](int rel) { const int v = rel + phase0; return sl0 + (v >= nsl ? v - nsl : v); };

  // ---- X: 32 rows per wave, quantised with each row's own scale, through the wave-private LDS image: rows are gathered
  //      through pos and read fully coalesced (two 1-KB instructions per row, lane = 16 bytes of the row), amax by a wave
  //      reduction, e4m3 bytes to LDS (rows padded to 528 B: conflict-free 16-B column reads); then lane (token r, half h)
  //      reads its fragments: k-step 2 m + s' holds x[32 m + 16 h + 8 s' + j], i.e. the 16 bytes at 32 m + 16 h of the row
  //      are the operands of two k-steps ----
  long xq[kD / 16];
  float sx = 1.f, sx_n = 1.f;
  auto quant_row = [&](const f32x4& v0, const f32x4& v1, int i, float& sxv) {
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(v0[j]), fabsf(v1[j])));
    amax = fmaxf(wave_max(amax), 1e-30f);
    const float inv = 448.f * __builtin_amdgcn_rcpf(amax);
    if (r == i) sxv = amax * (1.f / 448.f);
    int q0 = 0, q1 = 0;
    q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v0[0] * inv, v0[1] * inv, q0, false);
    q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v0[2] * inv, v0[3] * inv, q0, true);
    q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v1[0] * inv, v1[1] * inv, q1, false);
    q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v1[2] * inv, v1[3] * inv, q1, true);
    *reinterpret_cast<int*>(img + i * kImgRow + 4 * lane) = q0;
    *reinterpret_cast<int*>(img + i * kImgRow + 256 + 4 * lane) = q1;
  };
  // (the image is private to this wave: LDS operations of one wave complete in order, no barrier needed)
  auto read_xq = [&]() {
    int off = r * kImgRow + 16 * h;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int m = 0; m < kD / 32; ++m) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(img + off + 32 * m);
      xq[2 * m] = (long)(((unsigned long long)t[1] << 32) | t[0]);
      xq[2 * m + 1] = (long)(((unsigned long long)t[3] << 32) | t[2]);
    }
  };
  if constexpr (XQ) {
    const int my_src = pos[min(tile_row0 + r, row_end - 1)];
    sx = xq_scale[my_src];
    // two rows per instruction (lanes 0-31 / 32-63: 16 B each of a 512-B row), all 16 instructions in flight, straight into the image
    u32x4 t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int src = __shfl(my_src, 2 * i + h, 64);
      t[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xq_in + (size_t)src * kD + 16 * r));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4*>(img + (2 * i + h) * kImgRow + 16 * r) = t[i];
    read_xq();
  } else {
    const int my_src = pos[min(tile_row0 + r, row_end - 1)];          // source row of token r (lanes r and r + 32 agree)
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8) {                                    // 8 rows per batch: 16 loads in flight per lane
      f32x4 v[8][2];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float* xr = x + (size_t)__shfl(my_src, 8 * g8 + i, 64) * ldx + 4 * lane;
        v[i][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr));        // read once: keep W in L2
        v[i][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 256));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) quant_row(v[i][0], v[i][1], 8 * g8 + i, sx);
    }
    read_xq();
  }
  if (threadIdx.x * 4 < nsl * 64) {
    *reinterpret_cast<f32x4*>(b1_lds + threadIdx.x * 4) = ldg4(b1 + (size_t)cur.e * F + sl0 * 64 + threadIdx.x * 4);
    *reinterpret_cast<f32x4*>(s1_lds + threadIdx.x * 4) = ldg4(s1 + (size_t)cur.e * F + sl0 * 64 + threadIdx.x * 4);
  }
  s2h_lds[threadIdx.x] = s2[(size_t)cur.e * kD + threadIdx.x] * h_scale;
  s2h_lds[threadIdx.x + 256] = s2[(size_t)cur.e * kD + threadIdx.x + 256] * h_scale;
  int par = 0;

  // ---- the NEXT tile's X arrives under this tile's MFMAs: two rows per step by LDS-DMA into the raw rows (no registers
  //      held across the latency; 8 KB per CU in flight, below the LDS-DMA limit), quantised one step later ----
  const unsigned long long xaddr = XQ ? (unsigned long long)xq_in : (unsigned long long)x;
  const u32x4 rsx = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xaddr),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)((xaddr >> 32) & 0xffffu)),
                     XQ ? (unsigned)S * (unsigned)kD : ((unsigned)(S - 1) * (unsigned)ldx + kD) * 4u, 0x00020000u};
  const unsigned raw_lds = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)rawb);
  int my_src_n = 0;
  auto x_issue = [&](int i) {                           // row i of the next tile -> raw row i & 1
    const unsigned soff = (unsigned)__builtin_amdgcn_readlane(my_src_n, i) * (XQ ? (unsigned)kD : (unsigned)ldx * 4u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of the raw row's previous occupant have returned
    const unsigned dst = raw_lds + (unsigned)(i & 1) * 2048u;
    if constexpr (XQ) {
      // a quantised row is 512 B: both halves of the wave fetch it (lanes l and l + 32 the same 16 B), and the fill is issued
      // TWICE so that the counted waits of the step schedule (two fills per row) hold unchanged; 3 of the 4 copies come out of L2
      const unsigned voff = (unsigned)(olane() & 31) * 16u;
      dma16_hidden(rsx, voff, soff, dst);
      dma16_hidden(rsx, voff, soff, dst + 1024u);
    } else {
    const unsigned voff = (unsigned)olane() * 16u;
    dma16_hidden(rsx, voff, soff, dst);                   // (asm volatile + memory clobber: no load is moved across the
    dma16_hidden(rsx, voff + 1024u, soff, dst + 1024u);   // fills; the counted waits rely on the issue order)
    }
  };
  // quantisation of an arrived row in two parts, each with ONE LDS round trip (the wave stalls on every lgkmcnt wait -- it is
  // the only wave of its SIMD): part 1 reads the row and keeps only its partial amax, part 2 (a service later) reduces it over
  // the wave without the LDS pipe, re-reads the row and converts it
  float x_amax = 0.f;
  auto x_quant1 = [&](int i) {                          // (after the wait for row i's two fills)
    if constexpr (XQ) return;
    const char* src = rawb + (i & 1) * 2048 + olane() * 16;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 1024);
    float m;                                             // four v_max3_f32 with |.| source modifiers
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(v0[0]), "v"(v0[1]), "v"(v0[2]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(v0[3]), "v"(v1[0]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(v1[1]), "v"(v1[2]));
    asm("v_max_f32 %0, %0, |%1|" : "+v"(m) : "v"(v1[3]));
    x_amax = m;
  };
  float x_inv = 0.f;
  auto x_quant2a = [&](int i) {                          // reduce over the wave, scale
    if constexpr (XQ) return;
    const float amax = fmaxf(wave_amax_dpp(x_amax), 1e-30f);
    x_inv = 448.f * __builtin_amdgcn_rcpf(amax);
    if ((olane() & 31) == i) sx_n = amax * (1.f / 448.f);
  };
  auto x_quant2b = [&](int i) {                          // convert (saturating: MODE.FP16_OVFL) and store into the image
    const int l = olane();
    if constexpr (XQ) {                                  // the arrived row is e4m3 already: raw row -> image (lanes l and l + 32 write the same bytes)
      *reinterpret_cast<u32x4*>(img + i * kImgRow + 16 * (l & 31)) = *reinterpret_cast<const u32x4*>(rawb + (i & 1) * 2048 + 16 * (l & 31));
      return;
    }
    const char* src = rawb + (i & 1) * 2048 + l * 16;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 1024);
    const float inv = x_inv;
    int q0 = 0, q1 = 0;
    q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v0[0] * inv, v0[1] * inv, q0, false);
    q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v0[2] * inv, v0[3] * inv, q0, true);
    q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v1[0] * inv, v1[1] * inv, q1, false);
    q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v1[2] * inv, v1[3] * inv, q1, true);
    *reinterpret_cast<int*>(img + i * kImgRow + 4 * l) = q0;
    *reinterpret_cast<int*>(img + i * kImgRow + 256 + 4 * l) = q1;
  };
  auto x_quant = [&](int i) {
    x_quant1(i);
    x_quant2a(i);
    x_quant2b(i);
  };

  // ---- weight staging through registers: wave wv brings KB 8 wv .. 8 wv + 7 of every 32-KB piece, natural (fully
  //      coalesced) source order; the row permutation pi8 and the bank swizzles are applied to the LDS DESTINATION ----
  // W1 piece = 64 rows x 512 B.  LDS row rho of a piece (rows padded to 528 B) holds the row f = 32 (rho >> 5) +
  //   pi8(rho & 31) of the slice.  Instruction ii of wave wv covers four HALF rows (16 lanes x 16 B = 256 B each):
  //   rho = 16 (ii >> 1) + 4 wv + (lane >> 4), half ii & 1 -- the row-dependent part of f depends on the lane only, so source
  //   and destination are ONE per-lane base each plus immediates
  // W2 piece = 512 rows x 64 B: instruction ii covers rows 128 wv + 16 ii + (lane >> 2); chunk c = lane & 3 goes to physical
  //   chunk c ^ ((row >> 2) & 3) (independent of ii)
#ifndef M3_F8_RECOMPUTE_ADDR
  const unsigned src1_lane = (unsigned)((16 * (wv & 1) + 4 * (wv >> 1) + (lane >> 4)) * 512 + (lane & 15) * 16);
  const unsigned src2_lane = (unsigned)((128 * wv + (lane >> 2)) * w2_row_stride + (lane & 3) * 16);   // + ii * 16 rows + slice
  const unsigned dst1_lane = (unsigned)((4 * wv + (lane >> 4)) * kW1Row + (lane & 15) * 16);
  const unsigned dst2_lane = (unsigned)((128 * wv + (lane >> 2)) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
#endif
  u32x4 stg[8];                                        // the piece in flight (this wave's 8 KB)
  // one KB of the stream: ds_write the KB of piece t + 1 that arrived a step ago, then re-use its registers for the same KB
  // of piece t + 2.  A step spreads its 8 services between its MFMA batches: a wave that issues 8 loads back to back waits
  // ~160 cycles per load for the CU's load path (in-kernel stamps), one load per 8 MFMAs issues into an idle path.
  auto load_kb = [&](const unsigned char* wa, const unsigned char* wb, int sl, int q, int ii) {
    if (q == 0) {
      stg[ii] = *reinterpret_cast<const u32x4*>(wa + (size_t)sl * 32768 + src1_lane + ((ii & 1) * 256 + ((ii >> 1) & 1) * 4096 + (ii >> 2) * 16384));
    } else {
      stg[ii] = *reinterpret_cast<const u32x4*>(wb + (size_t)sl * w2_slice_stride + src2_lane + (size_t)ii * 16 * w2_row_stride);
    }
  };
  auto store_kb = [&](int q, int slot_i, int ii) {
    char* dst = smem + slot_i * kSlot;
    if (q == 0) {
      *reinterpret_cast<u32x4*>(dst + dst1_lane + (ii >> 1) * (16 * kW1Row) + (ii & 1) * 256) = stg[ii];
    } else {
      *reinterpret_cast<u32x4*>(dst + dst2_lane + ii * 1024) = stg[ii];
    }
  };

  // fragment read addresses: W1 block fb, step pair m: byte fb * 32 * 528 + r * 528 + 16 h + 32 m (padded rows: no swizzle)
  //                          W2 block db, step pair m: byte db * 2048 + (rd2 ^ (m << 5)), rd2 = (r << 6) | ((h ^ ((r >> 2) & 3)) << 4)

  f32x16 accy[16];

  // pieces are numbered t = 2 * slice + q (q = 0: W1 rows, 1: W2 columns of the slice) and live in LDS slot t & 1; the
  // stream runs on across tiles (np is even): pieces np, np + 1 of a tile are pieces 0, 1 of the work-group's next tile
#pragma unroll
  for (int ii = 0; ii < 8; ++ii) load_kb(w1e, w2e, abs_slice(0), 0, ii);
#pragma unroll
  for (int ii = 0; ii < 8; ++ii) {
    store_kb(0, 0, ii);                                // piece 0 -> slot 0 (waits for its loads)
    load_kb(w1e, w2e, abs_slice(0), 1, ii);            // piece 1 in flight
  }

  int hq32[2][4];                                      // Hq of the current slice: [f block][dword = 4 e4m3 of registers 4 d .. 4 d + 3]

  auto lo64 = [](const u32x4& a) { return (long)(((unsigned long long)a[1] << 32) | a[0]); };
  auto hi64 = [](const u32x4& a) { return (long)(((unsigned long long)a[3] << 32) | a[2]); };
  auto pack64 = [](int lo, int hi) { return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo); };

  // ---- The step as 32 SLOTS of two MFMAs (64 cycles of the matrix pipe) each.  This wave is the only one on its SIMD and
  //      issues in order: while an MFMA occupies the pipe, a second MFMA behind it blocks everything, an independent VALU /
  //      LDS / memory instruction does not.  Left to itself hipcc groups 8-14 MFMAs and the other ~470 instructions of a step
  //      into separate runs (measured: 75 k of the MFMAs' 131 k cycles per work-group not overlapped).  So every slot is
  //      [fragment read for slot + 3][2 MFMAs][<= ~14 other instructions], fenced by sched_barrier:
  //        slots 3, 7, .., 31   one KB of the weight stream (ds_write of piece t + 1, load of piece t + 2)
  //        slots 1, 5, .., 29   the next tile's X (row A: wait + partial amax, reduce, convert, next fill; row B the same)
  //        even slots           SiLU + quantisation of two z registers (GEMM-1: block 0 in the second half; GEMM-2: block 1
  //                             in the first half)
  // SiLU in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two elements per instruction) with the H scale folded into the
  // sigmoid's denominator: H / h_scale = z / ((1 + 2^(-z log2 e)) h_scale); b1 / s1 of a pair are fetched one pair ahead.
  const f32x2 hs2 = {h_scale, h_scale};
  f32x2 bb_c, ss_c;
  auto bias_fetch = [&](int sl_rel, int fb, int p) {
    const int fo = sl_rel * 64 + fb * 32 + 16 * (olane() >> 5) + 2 * p;
    bb_c = *reinterpret_cast<const f32x2*>(b1_lds + fo);
    ss_c = *reinterpret_cast<const f32x2*>(s1_lds + fo);
  };
  auto silu_pair = [&](const f32x16& acc, int fb, int p, int sl_rel_n, int fb_n, int p_n) {
#ifdef M3_F8_ABL_NO_SILU   // ablation builds (diagnostics; results are garbage)
    hq32[fb][p >> 1] = __builtin_bit_cast(int, acc[2 * p]);
    return;
#endif
    const f32x2 bb = bb_c, ss = ss_c;
    bias_fetch(sl_rel_n, fb_n, p_n);
    const f32x2 a2 = {acc[2 * p], acc[2 * p + 1]};
    const f32x2 sc = ss * f32x2{sx, sx};
    const f32x2 z = __builtin_elementwise_fma(a2, sc, bb);
    const f32x2 t = z * f32x2{-1.44269504088896f, -1.44269504088896f};
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    const f32x2 d = __builtin_elementwise_fma(e, hs2, hs2);
    const f32x2 hh = z * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    // (saturating conversion: MODE.FP16_OVFL is set)
    if (p & 1) hq32[fb][p >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(hh[0], hh[1], hq32[fb][p >> 1], true);
    else hq32[fb][p >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(hh[0], hh[1], 0, false);
    asm volatile("" : "+v"(hq32[fb][p >> 1]));          // (pins the pair to its slot: hipcc otherwise sinks the transcendentals
  };                                                    //  of all pairs to the end of the step, where nothing overlaps them)
  f32x16 acc_a, acc_b;                                 // z blocks fb = 0 / 1 of the current slice (arch VGPRs: asm MFMAs)
  // fragment read addresses: W1 block fb, step pair m: byte fb * 32 * 528 + r * 528 + 16 h + 32 m (padded rows: no swizzle)
  //                          W2 block db, step pair m: byte db * 2048 + (rd2 ^ (m << 5)), rd2 = (r << 6) | ((h ^ ((r >> 2) & 3)) << 4)
  // step q = 0 of a slice: slots 0..15 z block 0, slots 16..31 z block 1
  auto gemm1 = [&](const char* slot, auto&& work) {
    const int l = olane();
    const char* rp = slot + (l & 31) * kW1Row + (l >> 5) * 16;
    auto frag = [&](int s) -> u32x4 {
#ifdef M3_F8_ABL_NO_READS
      u32x4 v = {1u, 2u, 3u, 4u};
      asm volatile("" : "+v"(v));
      return v;
#else
      return *reinterpret_cast<const u32x4*>(rp + (s >> 4) * (32 * kW1Row) + (s & 15) * 32);
#endif
    };
    u32x4 fr[4];
#pragma unroll
    for (int s = 0; s < 3; ++s) fr[s] = frag(s);
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      if (s + 3 < 32) fr[(s + 3) & 3] = frag(s + 3);
      const int m = s & 15;
      if (s < 16) {
        if (m == 0) mfma8_v0(acc_a, lo64(fr[s & 3]), xq[0]);
        else mfma8_v(acc_a, lo64(fr[s & 3]), xq[2 * m]);
        mfma8_v(acc_a, hi64(fr[s & 3]), xq[2 * m + 1]);
      } else {
        if (m == 0) mfma8_v0(acc_b, lo64(fr[s & 3]), xq[0]);
        else mfma8_v(acc_b, lo64(fr[s & 3]), xq[2 * m]);
        mfma8_v(acc_b, hi64(fr[s & 3]), xq[2 * m + 1]);
      }
      work(s);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // step q = 1: slots 0..15 multiply the f block 0 half of W2 into the 16 output blocks, slots 16..31 the f block 1 half
  auto gemm2 = [&](const char* slot, auto&& work) {
    const int l = olane();
    const int rb = ((l & 31) << 6) | (((l >> 5) ^ ((l >> 2) & 3)) << 4);
    auto frag = [&](int s) -> u32x4 {
#ifdef M3_F8_ABL_NO_READS
      u32x4 v = {1u, 2u, 3u, 4u};
      asm volatile("" : "+v"(v));
      return v;
#else
      return *reinterpret_cast<const u32x4*>(slot + (s & 15) * 2048 + (rb ^ ((s >> 4) << 5)));
#endif
    };
    u32x4 fr[4];
#pragma unroll
    for (int s = 0; s < 3; ++s) fr[s] = frag(s);
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      if (s + 3 < 32) fr[(s + 3) & 3] = frag(s + 3);
      const int m = s >> 4, db = s & 15;
      accy[db] = mfma8(lo64(fr[s & 3]), pack64(hq32[m][0], hq32[m][1]), accy[db]);
      accy[db] = mfma8(hi64(fr[s & 3]), pack64(hq32[m][2], hq32[m][3]), accy[db]);
      work(s);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  M3_DIAG(unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; dg[6] = __builtin_amdgcn_s_memtime() - t_k0;)
  for (;;) {
    // ---- the work-group's next item (if any): its weights follow this tile's in the piece stream, its X is prefetched ----
    const int item_n = item + stride;
    const bool has_next = item_n < item_end;
    TileRef nxt = cur;
    if (has_next) nxt = find_tile(acc_hist, E, item_n, fsplit, lane);
    const int row_end_n = acc_hist[nxt.e + 1];
    const int tile_row0_n = acc_hist[nxt.e] + nxt.tt * kTok + wv * 32;
    const int sl0_n = nxt.fs * nsl, phase0_n = (nxt.tt * 5) % nsl;
    const unsigned char* w1n = w1 + (size_t)nxt.e * F * kD;
    const unsigned char* w2n = w2 + (size_t)nxt.e * F * kD;
    if (has_next) my_src_n = pos[min(tile_row0_n + r, row_end_n - 1)];
    if constexpr (XQ) { if (has_next) sx_n = xq_scale[my_src_n]; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) accy[i][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+a"(accy[i]));
    // settle the pos load HERE (behind the zeroing): a load still pending in hipcc's model at the loop head would put an
    // s_waitcnt vmcnt(0) in front of every fill of the loop, draining the weight loads in flight each time
    asm volatile("" : "+v"(my_src_n));
    if constexpr (XQ) asm volatile("" : "+v"(sx_n));

    M3_DIAG(const unsigned long long t_begin = __builtin_amdgcn_s_memtime();)
    // step t: barrier (every wave is done reading slot (t + 1) & 1, piece t is visible) -> MFMAs on piece t, and between the
    // MFMA batches, KB by KB: ds_write piece t + 1 (arrived during step t - 1) into the other slot, load piece t + 2
#pragma clang loop unroll(disable)
    for (int sl = 0; sl < nsl; ++sl) {
      const bool last_sl = sl + 1 == nsl;
      const unsigned char* wa = last_sl ? w1n : w1e;   // piece t + 2 of the last slice's steps = piece 0 / 1 of the next tile
      const unsigned char* wb = last_sl ? w2n : w2e;
      int sl_next = abs_slice(sl + 1);
      if (last_sl) sl_next = sl0_n + phase0_n;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 2 * sl + q;
        M3_DIAG(const unsigned long long c0 = __builtin_amdgcn_s_memtime();)
        __syncthreads();
        M3_DIAG(const unsigned long long c1 = __builtin_amdgcn_s_memtime();)
#ifdef M3_F8_ABL_NO_XPF
        const bool xq_step = false, xi_step = false;
#else
        const bool xq_step = has_next && t >= 1 && t <= 16, xi_step = has_next && t < 16;
#endif
        // vmcnt arithmetic of the X prefetch: a step issues its 8 weight loads in slots 3, 7, .., 31, row A's two fills in slot
        // 13 and row B's in slot 29; row A is first read in slot 1 of the next step (younger: 5 loads + 2 fills = 7), row B in
        // slot 17 (1 + 4 loads + row A's 2 new fills = 7, or 5 at t = 16 when no further fills follow)
        const int sl_rel = abs_slice(sl) - sl0, sl_rel_next = last_sl ? 0 : abs_slice(sl + 1) - sl0;
        auto work = [&](int s) {
          if ((s & 3) == 3) {
#ifndef M3_F8_ABL_NO_STAGE
            // (unconditional: past the work-group's last piece the stream re-loads that tile's first pieces into slots
            //  nobody reads any more -- straight-line code keeps hipcc's vmcnt arithmetic exact)
            store_kb(q ^ 1, q ^ 1, s >> 2);            // piece t + 1 has the other q and the other slot
            load_kb(wa, wb, sl_next, q, s >> 2);
#endif
          } else if (s & 1) {
            const int row = 2 * (t - 1) + (s >> 4);    // the row that arrived during the previous step
            switch ((s >> 2) & 3) {
              case 0:
                if (xq_step) {
                  if (s < 16 || t < 16) wait_vmcnt<7>(); else wait_vmcnt<5>();
                  x_quant1(row);
                }
                break;
              case 1: if (xq_step) x_quant2a(row); break;
              case 2: if (xq_step) x_quant2b(row); break;
              default: if (xi_step) x_issue(2 * t + (s >> 4)); break;
            }
          } else if (q == 0 && s >= 16) {
            const int p = (s - 16) >> 1;
            // (the z blocks come from asm MFMAs: hipcc does not know they are matrix-pipe results and is free to place these
            //  VALU reads right behind slot 15's last MFMA -- inside a slot nothing orders them against the slot's own MFMAs.
            //  Without the settle the XQ instantiation's schedule read accumulator registers before they were written back:
            //  0.1 of |y| 2.6 wrong, deterministic; the builtin-MFMA build (-DM3_F8_ASM_MFMA=0) was bit-identical to the other form)
            if (s == 16) mfma8_v_settle(acc_a);
            silu_pair(acc_a, 0, p, sl_rel, p < 7 ? 0 : 1, p < 7 ? p + 1 : 0);
          } else if (q == 1 && s < 16) {
            const int p = s >> 1;
            if (s == 0) mfma8_v_settle(acc_b);
            silu_pair(acc_b, 1, p, p < 7 ? sl_rel : sl_rel_next, p < 7 ? 1 : 0, p < 7 ? p + 1 : 0);
          }
        };
        const char* slot = smem + q * kSlot;
        if (q == 0) {
          if (sl == 0) bias_fetch(sl_rel, 0, 0);       // (behind the barrier that makes this tile's b1 / s1 visible)
          gemm1(slot, work);
        } else {
          gemm2(slot, work);
        }
        M3_DIAG(asm volatile("s_nop 0" ::: "memory"); const unsigned long long c4 = __builtin_amdgcn_s_memtime();
                dg[0] += c1 - c0; if (q == 0) dg[3] += c4 - c1; else dg[4] += c4 - c1;)
      }
    }
    M3_DIAG(const unsigned long long t_loop = __builtin_amdgcn_s_memtime(); dg[5] += t_loop - t_begin;)

    // ---- tile end: the next tile's small operands, this tile's output, then the next tile's X fragments ----
    __syncthreads();                                   // every wave is past its last b1 / s1 read and its last read of slot 1
    if (has_next) {
      // (all loads of this section are consumed before the first Y store is issued: a wait for a load issued after the
      //  stores would wait for the stores as well, and they are meant to drain under the next tile's MFMAs)
      if ((nxt.e != cur.e || nxt.fs != cur.fs) && threadIdx.x * 4 < nsl * 64) {
        *reinterpret_cast<f32x4*>(b1_lds + threadIdx.x * 4) = ldg4(b1 + (size_t)nxt.e * F + sl0_n * 64 + threadIdx.x * 4);
        *reinterpret_cast<f32x4*>(s1_lds + threadIdx.x * 4) = ldg4(s1 + (size_t)nxt.e * F + sl0_n * 64 + threadIdx.x * 4);
      }
      s2h_lds[(par ^ 1) * 512 + threadIdx.x] = s2[(size_t)nxt.e * kD + threadIdx.x] * h_scale;
      s2h_lds[(par ^ 1) * 512 + threadIdx.x + 256] = s2[(size_t)nxt.e * kD + threadIdx.x + 256] * h_scale;
      // X rows the loop did not get to (short loops: np < 17): pending fills first, then the rest directly
      const int issued = 2 * min(np, 16), done = 2 * (min(np, 17) - 1);
      if (done < issued) {
        wait_vmcnt<0>();
        for (int i = done; i < issued; ++i) x_quant(i);
      }
      for (int i = issued; i < 32; i += 2) {
        x_issue(i);
        x_issue(i + 1);
        wait_vmcnt<0>();
        x_quant(i);
        x_quant(i + 1);
      }
    }
    M3_DIAG(const unsigned long long t_mid = __builtin_amdgcn_s_memtime(); dg[2] += t_mid - t_loop;)
    // Y[tok][d] = accy * s2[d] * h_scale, d = 32 db + (i & 3) + 8 (i >> 2) + 4 h.  A lane holds one token's column strip;
    // writing it out directly would touch 64 rows per store instruction in 16-B pieces.  So 64 columns at a time go through
    // this wave's quarter of weight slot 1 (free between the last step of a tile and the first service of the next) as
    // [32 tokens][256 B], 16-B chunks XOR-swizzled by the row, and leave as 256-B row segments.  The stores are not waited
    // for: they drain under the next tile's MFMAs; nothing between the first store and the next tile's loop touches vmcnt.
    {
      int le = lane;                                   // opaque copy: keeps the epilogue's per-lane addresses out of the
      asm volatile("" : "+v"(le));                     // registers that live across the whole tile loop
      const int er = le & 31, eh = le >> 5, rrow = le >> 4, rc = le & 15;
      char* ys = smem + kSlot + wv * 8192;
      const float* sc = s2h_lds + par * 512 + 4 * eh;
      float* yrow = ybuf + ((size_t)cur.fs * S + tile_row0) * kD + 4 * rc;
#pragma unroll
      for (int p8 = 0; p8 < 8; ++p8) {
        __builtin_amdgcn_sched_barrier(0);
        // (the two blocks of this pass stay in AGPRs until here: hipcc otherwise copies most of the output tile to VGPRs
        //  ahead of the tile-end barrier and spills the loop's constants to make room)
        asm volatile("" : "+a"(accy[2 * p8]), "+a"(accy[2 * p8 + 1]));
#pragma unroll
        for (int dbl = 0; dbl < 2; ++dbl) {
          const int db = 2 * p8 + dbl;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + 32 * db + 8 * m);
            const int c = 8 * dbl + 2 * m + eh;
            *reinterpret_cast<f32x4*>(ys + er * 256 + ((c ^ (er & 15)) << 4)) =
                f32x4{accy[db][4 * m] * s4[0], accy[db][4 * m + 1] * s4[1], accy[db][4 * m + 2] * s4[2], accy[db][4 * m + 3] * s4[3]};
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                      // instruction i: tokens 4 i .. 4 i + 3, 256 B each
          const int tok = 4 * i + rrow;
          const f32x4 v = *reinterpret_cast<const f32x4*>(ys + tok * 256 + ((rc ^ (tok & 15)) << 4));
          if (tile_row0 + tok < row_end) store16_nt_hidden(yrow + (size_t)tok * kD + 64 * p8, v);
        }
      }
    }
    M3_DIAG(dg[1] += __builtin_amdgcn_s_memtime() - t_mid;)
    if (!has_next) break;
    read_xq();
    par ^= 1;
    item = item_n;
    cur = nxt;
    row_end = row_end_n;
    tile_row0 = tile_row0_n;
    sl0 = sl0_n;
    phase0 = phase0_n;
    w1e = w1n;
    w2e = w2n;
    sx = sx_n;
  }
  M3_DIAG(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0 && blockIdx.x < 1024) {
            unsigned long long* o = g_fused8_dbg + (blockIdx.x * 4 + wv) * 8;
            for (int i = 0; i < 7; ++i) o[i] = dg[i];
            o[7] = __builtin_amdgcn_s_memtime() - t_k0;
          })
}

// ---- host side ----
static int fused8_min_rows() {   // read once: the engine freezes the form (and the combine's slab layout) when a shape is bound
  static const int v = [] {
    const char* e = getenv("M3_EXPERT_FUSED_FP8_MIN_ROWS");
    return e ? atoi(e) : 4096;
  }();
  return v;
}
// persistent grid: one work-group per CU, a multiple of 8 (XCDs)
static int fused8_grid() {
  const int cus = device_cu_count();
  return cus >= 8 ? cus / 8 * 8 : 8;
}
int expert_ffn_fused_fp8_fsplit(int S, int E, int D, int F) {
  static const int forced = [] { const char* e = getenv("M3_FUSED8_FSPLIT"); return e ? atoi(e) : 0; }();
  if (forced == 1 || forced == 2 || forced == 4) return forced;
  // F parts per token tile.  A work item streams its part of the expert's weights (D F / fs bytes of W1 and of W2) plus the
  // tile's X and Y (128 tokens x D x 4 B each, whatever fs), all at the ~20 B/cycle one CU pulls; the persistent grid is one
  // work-group per CU, so a launch costs rounds = ceil(tiles fs / CUs) items.  Measured at configs[4]'s share (4480 live rows,
  // 64 experts, ~70 tiles): fs = 4 -> 54.7 us (2 rounds), fs = 2 -> 40.2 us (1 round), fs = 1 -> 59.9 us -- the model's
  // 1536 : 1024 : 1536.  (tiles: every expert may end in a partly filled tile.)
  const int tiles = cdiv(S, kTok) + E / 2;
  const int cus = fused8_grid();
  int fs = 1;
  long best = -1;
  for (int f = 1; f <= 4; f *= 2) {
    if (F % (128 * f) != 0) continue;
    const long cost = (long)cdiv(tiles * f, cus) * (2L * D * F / f + 2L * kTok * D * 4);
    if (best < 0 || cost < best) { best = cost; fs = f; }
  }
  while (F / fs > 1024 && fs < 4) fs *= 2;                   // b1 / s1 of a work item's F range: 1024 floats each in LDS
  return fs;
}
bool expert_ffn_fused_fp8_applies(int S, int E, int D, int F) {
  if (!(D == kD && F % 128 == 0 && F <= 4096 && S >= fused8_min_rows() && S >= 64 * E && E <= 1024)) return false;
  const int fs = expert_ffn_fused_fp8_fsplit(S, E, D, F);
  return F % (128 * fs) == 0 && F / fs <= 1024;
}
int init_expert_ffn_fused_fp8_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_fp8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_fp8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_fp8_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_fp8_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_fp8_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_fp8_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  once.mark();
  return 0;
}

// rows -> e4m3 + one fp32 scale per row: the quantisation quant_row applies inside the fused kernel and moe_router_kernel applies to
// the rows it normalises, as an operator of its own (m3_quantize_rows_e4m3; D = 512: one wave per row)
__global__ __launch_bounds__(256) void rows_to_e4m3_kernel(const float* __restrict__ x, int ldx, int S, unsigned char* __restrict__ xq,
                                                           float* __restrict__ sc) {
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= S) return;
  const f32x4 v0 = ldg4(x + (size_t)row * ldx + 4 * lane), v1 = ldg4(x + (size_t)row * ldx + 256 + 4 * lane);
  float amax = 0.f;
  for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(v0[j]), fabsf(v1[j])));
  amax = fmaxf(wave_max(amax), 1e-30f);
  const float inv = 448.f * __builtin_amdgcn_rcpf(amax);
  int q0 = 0, q1 = 0;
  q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v0[0] * inv, v0[1] * inv, q0, false);
  q0 = __builtin_amdgcn_cvt_pk_fp8_f32(v0[2] * inv, v0[3] * inv, q0, true);
  q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v1[0] * inv, v1[1] * inv, q1, false);
  q1 = __builtin_amdgcn_cvt_pk_fp8_f32(v1[2] * inv, v1[3] * inv, q1, true);
  *reinterpret_cast<int*>(xq + (size_t)row * 512 + 4 * lane) = q0;
  *reinterpret_cast<int*>(xq + (size_t)row * 512 + 256 + 4 * lane) = q1;
  if (lane == 0) sc[row] = amax * (1.f / 448.f);
}
int launch_quantize_rows_e4m3(const float* x, int ldx, int S, int D, void* xq, float* scale, hipStream_t stream) {
  M3_REQUIRE(D == kD && (ldx & 3) == 0 && S >= 0, "quantize_rows_e4m3: D=%d must be %d, ldx=%d a multiple of 4", D, kD, ldx);
  M3_REQUIRE(x && xq && scale, "quantize_rows_e4m3: null pointer");
  if (S == 0) return 0;
  hipLaunchKernelGGL(rows_to_e4m3_kernel, dim3((S + 3) / 4), dim3(256), 0, stream, x, ldx, S, (unsigned char*)xq, scale);
  M3_LAUNCH_CHECK();
  return 0;
}

int launch_expert_ffn_fused_fp8(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D, int F,
                                const void* w1, const float* s1, const float* b1, const void* w2, const float* s2, int w2_sliced,
                                float h_scale, float* ybuf, hipStream_t stream, const void* xq, const float* xq_scale, int32_t* fs_dev) {
  M3_REQUIRE(xq == nullptr || xq_scale != nullptr, "expert_ffn_fused_fp8: quantised rows without their scales");
  M3_REQUIRE(expert_ffn_fused_fp8_applies(S, E, D, F), "expert_ffn_fused_fp8: shape S=%d E=%d D=%d F=%d not supported", S, E, D, F);
  M3_REQUIRE((ldx & 3) == 0, "expert_ffn_fused_fp8: ldx=%d must be a multiple of 4", ldx);
  M3_REQUIRE(h_scale > 0.f, "expert_ffn_fused_fp8: h_scale must be positive (got %g)", (double)h_scale);
  if (int rc = init_expert_ffn_fused_fp8_kernels()) return rc;
  const int fsplit = expert_ffn_fused_fp8_fsplit(S, E, D, F);
  M3_REQUIRE((size_t)S * ldx * 4 < ((size_t)1 << 32), "expert_ffn_fused_fp8: input of %d rows x %d floats exceeds a 4-GB buffer", S, ldx);
  const int nblk = fused8_grid();
  const int row_stride = w2_sliced ? 64 : F;                 // bytes between consecutive d rows of W2
  const int slice_stride = w2_sliced ? D * 64 : 64;          // bytes between consecutive 64-wide f slices
  const size_t lds = kLdsBytes;
#define M3_FUSED8_LAUNCH(FS_)                                                                                              \
  do {                                                                                                                     \
    if (xq != nullptr)                                                                                                     \
      hipLaunchKernelGGL((expert_ffn_fused_fp8_kernel<FS_, true>), dim3(nblk), dim3(256), lds, stream, x, ldx, pos, acc_hist, S, \
                         E, F, (const unsigned char*)w1, s1, b1, (const unsigned char*)w2, s2, row_stride, slice_stride,  \
                         h_scale, ybuf, (const unsigned char*)xq, xq_scale, fs_dev);                                      \
    else                                                                                                                   \
      hipLaunchKernelGGL((expert_ffn_fused_fp8_kernel<FS_, false>), dim3(nblk), dim3(256), lds, stream, x, ldx, pos, acc_hist, S, \
                         E, F, (const unsigned char*)w1, s1, b1, (const unsigned char*)w2, s2, row_stride, slice_stride,  \
                         h_scale, ybuf, (const unsigned char*)nullptr, (const float*)nullptr, fs_dev);                    \
  } while (0)
  if (fsplit == 4) M3_FUSED8_LAUNCH(4); else if (fsplit == 2) M3_FUSED8_LAUNCH(2); else M3_FUSED8_LAUNCH(1);
#undef M3_FUSED8_LAUNCH
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3

#ifdef M3_FUSED_DIAG
extern "C" int m3_debug_fused8_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(m3::g_fused8_dbg), bytes < sizeof(m3::g_fused8_dbg) ? bytes : sizeof(m3::g_fused8_dbg));
}
#endif
