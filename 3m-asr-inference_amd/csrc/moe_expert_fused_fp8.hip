// Grouped expert FFN in fp8 arithmetic (A8W8): e4m3 weights x e4m3 activations on v_mfma_f32_32x32x16_fp8_fp8, fp32
// accumulation, ONE kernel, H never leaves the registers.  This is the path the reference's `--int8` flag names and never
// finished (builder.py:39-49 `assert 0`; fmoe_expert_plugin.cpp:264-266 asserts on anything but fp32): there is no
// reference output for it, the arithmetic is defined here and tested against an fp64 evaluation of exactly this
// quantised computation (tests/test_fp8_gpu.py).
//
// Quantisation (OCP e4m3fn, |max| = 448, conversions round to nearest even; values are clamped to +-448 first because
// v_cvt_pk_fp8_f32 turns an out-of-range input into NaN on gfx950, it does not saturate):
//   W1[e][f][:], W2[e][d][:]   e4m3 with one fp32 scale per output row (s1[e][f], s2[e][d]) -- the plan's fp8 format
//   X[tok][:]                  e4m3 with a per-token DYNAMIC scale sx[tok] = amax(row) / 448, computed while the row is loaded
//   H[tok][:]                  e4m3 with ONE static scale per layer, h_scale (from the calibrator: amax of H over the
//                              calibration batches x 1.25 / 448; e4m3 is a floating-point format, so the scale only has to
//                              keep H inside [2^-9, 448] h_scale -- it does not set the precision)
//   z = (W1q . Xq) s1[f] sx[tok] + b1[f];  H = SiLU(z);  Y = (W2q . Hq) s2[d] h_scale      (b2 etc.: moe_combine_kernel)
//
// Structure: the transposed, register-resident formulation of moe_expert_fused_bf16.hip (tokens on the MFMA column axis,
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
constexpr int kPiece = 32768;        // bytes per ring slot = one 64-wide slice of W1 (64 rows x 512 B) or of W2 (512 rows x 64 B)
constexpr int kRing = 2;
constexpr int kD = 512;

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ f32x16 mfma8(long a, long b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
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
// LDS row rho (0..31) of a W1 block holds the block's row pi8(rho): rho = 16 s + 8 a + 4 h + b  ->  16 h + 8 s + 4 a + b
__device__ __forceinline__ int pi8_row(int rho) {
  return (((rho >> 2) & 1) << 4) | (((rho >> 4) & 1) << 3) | (((rho >> 3) & 1) << 2) | (rho & 3);
}

}  // namespace

template <int FSPLIT>
__global__ __launch_bounds__(256) void expert_ffn_fused_fp8_kernel(
    const float* __restrict__ x, int ldx, const int32_t* __restrict__ pos, const int32_t* __restrict__ acc_hist, int S, int E,
    int F, const unsigned char* __restrict__ w1, const float* __restrict__ s1, const float* __restrict__ b1,
    const unsigned char* __restrict__ w2, const float* __restrict__ s2, int w2_row_stride, int w2_slice_stride, float h_scale,
    float* __restrict__ ybuf, int nblk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // two 32-KB slots | b1 | s1 of this work-group's F range
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;

  // ---- work-group -> (token tile, F part).  The number of real tiles T is only known on the device (it depends on the
  //      routing): every work-group sums it from the histogram, then XCD x (= blockIdx % 8: work-groups b and b + 8 share an
  //      XCD and its L2) takes the consecutive items [x * per, (x + 1) * per), per = ceil(T * FSPLIT / 8) -- the tiles of one
  //      expert stay on one XCD and all XCDs get the same number of tiles; the surplus work-groups of the worst-case grid exit
  int total_tiles = 0;
  for (int e0 = 0; e0 < E; e0 += 64) {
    const int ee = e0 + lane;
    int nt = ee < E ? (acc_hist[ee + 1] - acc_hist[ee] + kTok - 1) / kTok : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nt += __shfl_xor(nt, d, 64);
    total_tiles += nt;
  }
  const int items = total_tiles * FSPLIT;
  const int per = (items + 7) >> 3;
  const int slot_in_xcd = blockIdx.x >> 3;
  const int logical = (blockIdx.x & 7) * per + slot_in_xcd;
  if (slot_in_xcd >= per || logical >= items) return;  // (uniform over the work-group)
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
  if (e < 0) return;
  M3_DIAG(const unsigned long long t_k0 = __builtin_amdgcn_s_memtime();)
  e = __builtin_amdgcn_readfirstlane(e);               // provably uniform: descriptors / piece offsets stay in SGPRs
  tt = __builtin_amdgcn_readfirstlane(tt);
  const int row_begin = acc_hist[e], row_end = acc_hist[e + 1];
  const int nsl = F / (64 * FSPLIT);                   // 64-wide slices of F this work-group contracts (even)
  const int sl0 = fs * nsl;
  const int phase0 = (tt * 5) % nsl;                   // de-phased walk over the slices (see moe_expert_fused_bf16.hip)
  auto abs_slice = [&](int rel) { const int v = rel + phase0; return sl0 + (v >= nsl ? v - nsl : v); };

  float* b1_lds = reinterpret_cast<float*>(smem + kRing * kPiece);
  float* s1_lds = b1_lds + nsl * 64;

  // ---- X: the 32 rows of this wave, quantised with each row's own scale, through a wave-private LDS image.
  //      Rows are gathered through pos and read fully coalesced (two 1-KB instructions per row, lane = 16 bytes of the row),
  //      amax by a wave reduction, e4m3 bytes to LDS (rows padded to 528 B: conflict-free 16-B column reads); then lane
  //      (token r, half h) reads its fragments: k-step 2 m + s' holds x[32 m + 16 h + 8 s' + j], i.e. the 16 bytes at
  //      32 m + 16 h of the row are the operands of two k-steps ----
  long xq[kD / 16];
  float sx = 1.f;
  {
    constexpr int kXs = 528;
    char* xs = smem + wv * (32 * kXs);
    const int tile_row0 = row_begin + tt * kTok + wv * 32;
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
      for (int i = 0; i < 8; ++i) {
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(v[i][0][j]), fabsf(v[i][1][j])));
        amax = fmaxf(wave_max(amax), 1e-30f);
        const float inv = 448.f / amax;
        if (r == 8 * g8 + i) sx = amax * (1.f / 448.f);
        int q0 = 0, q1 = 0;
        q0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[i][0][0] * inv), clamp448(v[i][0][1] * inv), q0, false);
        q0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[i][0][2] * inv), clamp448(v[i][0][3] * inv), q0, true);
        q1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[i][1][0] * inv), clamp448(v[i][1][1] * inv), q1, false);
        q1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[i][1][2] * inv), clamp448(v[i][1][3] * inv), q1, true);
        *reinterpret_cast<int*>(xs + (8 * g8 + i) * kXs + 4 * lane) = q0;
        *reinterpret_cast<int*>(xs + (8 * g8 + i) * kXs + 256 + 4 * lane) = q1;
      }
    }
    // (the image is private to this wave: LDS operations of one wave complete in order, no barrier needed)
#pragma unroll
    for (int m = 0; m < kD / 32; ++m) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(xs + r * kXs + 32 * m + 16 * h);
      xq[2 * m] = (long)(((unsigned long long)t[1] << 32) | t[0]);
      xq[2 * m + 1] = (long)(((unsigned long long)t[3] << 32) | t[2]);
    }
  }
  __syncthreads();                                     // every wave is done with its X image: the slots may be filled
  for (int i = threadIdx.x * 4; i < nsl * 64; i += 1024) {
    *reinterpret_cast<f32x4*>(b1_lds + i) = ldg4(b1 + (size_t)e * F + sl0 * 64 + i);
    *reinterpret_cast<f32x4*>(s1_lds + i) = ldg4(s1 + (size_t)e * F + sl0 * 64 + i);
  }

  // ---- weight staging through registers: wave wv brings KB 8 wv .. 8 wv + 7 of every 32-KB piece, natural (fully
  //      coalesced) source order; the row permutation pi8 and the bank swizzles are applied to the LDS DESTINATION ----
  // W1 piece = 64 rows x 512 B: instruction ii covers rows f = 16 wv + 2 ii + (lane >> 5) (32 lanes x 16 B each); row f of
  //   a block goes to LDS row rho = pi8^-1(f): f = [h' s a b1 b0] -> rho = [s a h' b1 b0]; 16-B chunk c = lane & 31 goes to
  //   physical chunk (c & 16) | ((c ^ rho) & 15)
  // W2 piece = 512 rows x 64 B: instruction ii covers rows 128 wv + 16 ii + (lane >> 2); chunk c = lane & 3 goes to physical
  //   chunk c ^ ((row >> 2) & 3) (independent of ii)
  const unsigned char* w1e = w1 + (size_t)e * F * kD;
  const unsigned char* w2e = w2 + (size_t)e * F * kD;
  const unsigned src1_lane = (unsigned)(wv * 8192 + lane * 16);                       // + ii * 1024 + slice * 32768
  const unsigned src2_lane = (unsigned)((128 * wv + (lane >> 2)) * w2_row_stride + (lane & 3) * 16);   // + ii * 16 rows + slice
  const int hb = lane >> 5, c1 = lane & 31;
  const int x1 = c1 ^ (4 * (wv & 1) + hb);                                            // (c ^ rho) low bits, ii part XORed in below
  const unsigned dst2_lane = (unsigned)((128 * wv + (lane >> 2)) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
  u32x4 stg[8];                                        // the piece in flight (this wave's 8 KB)
  // one KB of the stream: ds_write the KB of piece t + 1 that arrived a step ago, then re-use its registers for the same KB
  // of piece t + 2.  A step spreads its 8 services between its MFMA batches: a wave that issues 8 loads back to back waits
  // ~160 cycles per load for the CU's load path (in-kernel stamps), one load per 8 MFMAs issues into an idle path.
  auto load_kb = [&](int sl, int q, int ii) {
    if (q == 0) stg[ii] = *reinterpret_cast<const u32x4*>(w1e + (size_t)sl * 32768 + src1_lane + ii * 1024);
    else stg[ii] = *reinterpret_cast<const u32x4*>(w2e + (size_t)sl * w2_slice_stride + src2_lane + (size_t)ii * 16 * w2_row_stride);
  };
  auto store_kb = [&](int q, int slot_i, int ii) {
    char* dst = smem + slot_i * kPiece;
    if (q == 0) {
      const int rho = 32 * (wv >> 1) + 16 * (ii >> 2) + 8 * ((ii >> 1) & 1) + 4 * (wv & 1) + 2 * (ii & 1) + hb;
      const int p = (c1 & 16) | ((x1 ^ (8 * ((ii >> 1) & 1) + 2 * (ii & 1))) & 15);
      *reinterpret_cast<u32x4*>(dst + rho * 512 + p * 16) = stg[ii];
    } else {
      *reinterpret_cast<u32x4*>(dst + dst2_lane + ii * 1024) = stg[ii];
    }
  };

  const int rd1 = (r << 9) | ((h ^ (r & 15)) << 4);            // W1 block fb, step pair m: byte fb * 16384 + (rd1 ^ (m << 5))
  const int rd2 = (r << 6) | ((h ^ ((r >> 2) & 3)) << 4);      // W2 block db, step pair m: byte db * 2048 + (rd2 ^ (m << 5))

  f32x16 accy[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) accy[i][j] = 0.f;

  // pieces are numbered t = 2 * slice + q (q = 0: W1 rows, 1: W2 columns of the slice) and live in LDS slot t & 1
#pragma unroll
  for (int ii = 0; ii < 8; ++ii) load_kb(abs_slice(0), 0, ii);
#pragma unroll
  for (int ii = 0; ii < 8; ++ii) {
    store_kb(0, 0, ii);                                // piece 0 -> slot 0 (waits for its loads)
    load_kb(abs_slice(0), 1, ii);                      // piece 1 in flight
  }
  __syncthreads();                                     // b1 / s1 and piece 0 in LDS

  const float inv_h = 1.f / h_scale;
  long hq[2][2];                                       // Hq fragments of the current slice: [f block][k-step of the block]

  auto lo64 = [](const u32x4& a) { return (long)(((unsigned long long)a[1] << 32) | a[0]); };
  auto hi64 = [](const u32x4& a) { return (long)(((unsigned long long)a[3] << 32) | a[2]); };
  // SiLU + quantisation of a finished 32 x 32 block of z, a quarter (registers 4 q4 .. 4 q4 + 3 = f 32 fb + 16 h + 4 q4 ..)
  // at a time: the quarters are issued between the MFMA batches of the NEXT block, so the VALU work runs in the MFMAs' shadow
  float hv[16];
  auto silu_quarter = [&](const f32x16& acc, int sl_rel, int fb, int q4) {
    const int fo = sl_rel * 64 + fb * 32 + 16 * h + 4 * q4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(b1_lds + fo);
    const f32x4 ss = *reinterpret_cast<const f32x4*>(s1_lds + fo);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float z = acc[4 * q4 + j] * (ss[j] * sx) + bb[j];
      hv[4 * q4 + j] = z * __builtin_amdgcn_rcpf(1.f + __expf(-z));
    }
  };
  f32x16 acc_a, acc_b;                                 // z blocks fb = 0 / 1 of the current slice
  // fragment reads run one batch of 4 (16 VGPRs, 8 k-steps) ahead of the MFMAs that consume them
  auto chain1 = [&](const char* blk, f32x16& acc, auto&& between) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    int rb = rd1;
    asm volatile("" : "+v"(rb));
    u32x4 a[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[0][j] = *reinterpret_cast<const u32x4*>(blk + (rb ^ (j << 5)));
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[(b + 1) & 1][j] = *reinterpret_cast<const u32x4*>(blk + (rb ^ ((4 * b + 4 + j) << 5)));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = 4 * b + j;
        acc = mfma8(lo64(a[b & 1][j]), xq[2 * m], acc);
        acc = mfma8(hi64(a[b & 1][j]), xq[2 * m + 1], acc);
      }
      between(b);
    }
  };
  // step q = 0 of a slice: z block 0, then z block 1 with SiLU(block 0) in its shadow; SiLU(block 1) runs in GEMM-2's shadow
  auto gemm1 = [&](const char* slot, int sl_rel, auto&& service) {
    chain1(slot, acc_a, [&](int b) { service(b); });
    chain1(slot + 16384, acc_b, [&](int b) {
      silu_quarter(acc_a, sl_rel, 0, b);
      service(4 + b);
    });
    hq[0][0] = q8(hv, inv_h);
    hq[0][1] = q8(hv + 8, inv_h);
  };
  // step q = 1: pass A multiplies the f block 0 half of W2 (k-steps 0, 1) into all 16 output blocks while SiLU(block 1)
  // is computed, pass B the f block 1 half
  auto gemm2 = [&](const char* slot, int sl_rel, auto&& service) {
    int rb = rd2;
    asm volatile("" : "+v"(rb));
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      u32x4 a[2][4];                                    // a batch = four output blocks
#pragma unroll
      for (int k = 0; k < 4; ++k) a[0][k] = *reinterpret_cast<const u32x4*>(slot + k * 2048 + (rb ^ (m << 5)));
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (b < 3) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            a[(b + 1) & 1][k] = *reinterpret_cast<const u32x4*>(slot + (4 * b + 4 + k) * 2048 + (rb ^ (m << 5)));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int db = 4 * b + k;
          accy[db] = mfma8(lo64(a[b & 1][k]), hq[m][0], accy[db]);
          accy[db] = mfma8(hi64(a[b & 1][k]), hq[m][1], accy[db]);
        }
        if (m == 0) silu_quarter(acc_b, sl_rel, 1, b);
        service(4 * m + b);
      }
      if (m == 0) {
        hq[1][0] = q8(hv, inv_h);
        hq[1][1] = q8(hv + 8, inv_h);
      }
    }
  };

  M3_DIAG(unsigned long long dg[5] = {0, 0, 0, 0, 0}; const unsigned long long t_begin = __builtin_amdgcn_s_memtime();)
  // step t: barrier (every wave is done reading slot (t + 1) & 1, piece t is visible) -> MFMAs on piece t, and between the
  // MFMA batches, KB by KB: ds_write piece t + 1 (arrived during step t - 1) into the other slot, load piece t + 2
  const int np = 2 * nsl;
  for (int sl = 0; sl < nsl; ++sl) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = 2 * sl + q;
      M3_DIAG(const unsigned long long c0 = __builtin_amdgcn_s_memtime();)
      if (t > 0) __syncthreads();
      M3_DIAG(const unsigned long long c1 = __builtin_amdgcn_s_memtime();)
      const bool has1 = t + 1 < np, has2 = t + 2 < np;
      const int sl_next = abs_slice(sl + 1 < nsl ? sl + 1 : sl);
      auto service = [&](int ii) {
        if (has1) store_kb(q ^ 1, q ^ 1, ii);          // piece t + 1 has the other q and the other slot
        if (has2) load_kb(sl_next, q, ii);
      };
      const char* slot = smem + q * kPiece;
      if (q == 0) gemm1(slot, abs_slice(sl) - sl0, service);
      else gemm2(slot, abs_slice(sl) - sl0, service);
      M3_DIAG(asm volatile("s_nop 0" ::: "memory"); const unsigned long long c4 = __builtin_amdgcn_s_memtime();
              dg[0] += c1 - c0; if (q == 0) dg[3] += c4 - c1; else dg[4] += c4 - c1;)
    }
  }

  M3_DIAG(if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long* o = g_fused8_dbg + (blockIdx.x * 4 + wv) * 8;
    for (int i = 0; i < 5; ++i) o[i] = dg[i];
    o[5] = __builtin_amdgcn_s_memtime() - t_begin;
    o[6] = t_begin - t_k0;                             // prologue: X load + quantisation, first pieces
  })
  // ---- epilogue: Y[tok][d] = accy * s2[d] * h_scale, d = 32 db + (i & 3) + 8 (i >> 2) + 4 h.  A lane holds one token's
  //      column strip; writing it out directly would touch 64 rows per store instruction in 16-B pieces.  So 128 columns
  //      at a time go through the wave-private LDS image [32 tokens][528 B] and leave as whole 512-B row segments ----
  __syncthreads();                                     // every wave is past its last fragment read
  {
    constexpr int kYs = 528;
    char* ys = smem + wv * (32 * kYs);
    const int tile_row0 = row_begin + tt * kTok + wv * 32;
    const float* s2r = s2 + (size_t)e * kD + 4 * h;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
      for (int dbl = 0; dbl < 4; ++dbl) {
        const int db = 4 * g4 + dbl;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const f32x4 sc = ldg4(s2r + 32 * db + 8 * m);
          *reinterpret_cast<f32x4*>(ys + r * kYs + (32 * dbl + 8 * m + 4 * h) * 4) =
              f32x4{accy[db][4 * m] * (sc[0] * h_scale), accy[db][4 * m + 1] * (sc[1] * h_scale),
                    accy[db][4 * m + 2] * (sc[2] * h_scale), accy[db][4 * m + 3] * (sc[3] * h_scale)};
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {                     // instruction i: tokens 2 i, 2 i + 1, 512 B each
        const int tok = 2 * i + (lane >> 5);
        const f32x4 v = *reinterpret_cast<const f32x4*>(ys + tok * kYs + (lane & 31) * 16);
        if (tile_row0 + tok < row_end)
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ybuf + ((size_t)fs * S + tile_row0 + tok) * kD + 128 * g4 + 4 * (lane & 31)));
      }
    }
  }
  M3_DIAG(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0 && blockIdx.x < 1024) g_fused8_dbg[(blockIdx.x * 4 + wv) * 8 + 7] = __builtin_amdgcn_s_memtime() - t_k0;)
}

// ---- host side ----
static int fused8_min_rows() {
  const char* e = getenv("M3_EXPERT_FUSED_FP8_MIN_ROWS");
  return e ? atoi(e) : 4096;
}
int expert_ffn_fused_fp8_fsplit(int S, int E, int D, int F) {
  const int tiles = cdiv(S, kTok) + E / 2;
  if (F % 512 == 0 && tiles < 112) return 4;
  return (tiles < 224 && F % 256 == 0) ? 2 : 1;
}
bool expert_ffn_fused_fp8_applies(int S, int E, int D, int F) {
  return D == kD && F % 128 == 0 && F <= 4096 && S >= fused8_min_rows() && S >= 64 * E && E <= 1024;
}
int init_expert_ffn_fused_fp8_kernels() {
  static bool done = false;
  if (done) return 0;
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_fp8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_fp8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_fused_fp8_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  done = true;
  return 0;
}

int launch_expert_ffn_fused_fp8(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D, int F,
                                const void* w1, const float* s1, const float* b1, const void* w2, const float* s2, int w2_sliced,
                                float h_scale, float* ybuf, hipStream_t stream) {
  M3_REQUIRE(expert_ffn_fused_fp8_applies(S, E, D, F), "expert_ffn_fused_fp8: shape S=%d E=%d D=%d F=%d not supported", S, E, D, F);
  M3_REQUIRE((ldx & 3) == 0, "expert_ffn_fused_fp8: ldx=%d must be a multiple of 4", ldx);
  M3_REQUIRE(h_scale > 0.f, "expert_ffn_fused_fp8: h_scale must be positive (got %g)", (double)h_scale);
  if (int rc = init_expert_ffn_fused_fp8_kernels()) return rc;
  const int fsplit = expert_ffn_fused_fp8_fsplit(S, E, D, F);
  const int tiles = cdiv(S, kTok) + E;
  const int nblk = cdiv(tiles * fsplit, 8) * 8;
  const int row_stride = w2_sliced ? 64 : F;                 // bytes between consecutive d rows of W2
  const int slice_stride = w2_sliced ? D * 64 : 64;          // bytes between consecutive 64-wide f slices
  const size_t lds = (size_t)kRing * kPiece + (size_t)(F / fsplit) * 2 * sizeof(float);
#define M3_FUSED8_LAUNCH(FS_)                                                                                              \
  hipLaunchKernelGGL((expert_ffn_fused_fp8_kernel<FS_>), dim3(nblk), dim3(256), lds, stream, x, ldx, pos, acc_hist, S, E, \
                     F, (const unsigned char*)w1, s1, b1, (const unsigned char*)w2, s2, row_stride, slice_stride, h_scale, ybuf, nblk)
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
