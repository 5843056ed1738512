// MoE router product:  logits[S, N] = [embed | LayerNorm(x)] . W_r^T (+ bias),  xn = LayerNorm(x) written out, exact fp32.
//
// Reference: the router matmul on cat([embed, x]) of trainer_3m_fix/layer/positionwise_feed_forward.py:169-180,225 (N = number
// of experts, 32 / 64), norm_ff of fmoe_transformer.py:138-141 in front of it (layer_norm_kernel.cu:33-139, with eps), and
// the normalised rows are the expert FFN's input.  The generic skinny GEMM (gemm.hip, affine-LayerNorm + concat form) did
// this with 16-column work-groups: every A row was re-read N / 16 times and normalised by each of them (two passes over the
// row), 45 us at 4.4 k rows and the slowest GEMM of the B = 1 chain (11.6 us).  Here one work-group owns 16 rows and ALL N
// columns:
//   phase 1  the 4 waves normalise 4 rows each (one pass: row in registers, mean / variance by DPP reductions), write xn to
//            memory and to LDS, and copy the embed rows to LDS (full-line loads, every A byte read once);
//   phase 2  the waves split K four ways; A fragments come from LDS (rows padded by 8 floats: conflict-free ds_read_b128),
//            W fragments straight from L2 into registers, two groups of 4 k-steps in flight (W is 128-256 KB and
//            shared by every work-group: it stays in L2);  v_mfma_f32_16x16x4_f32, N / 16 accumulator tiles per wave;
//   phase 3  fixed-order sum of the 4 K-partials through LDS (bitwise reproducible), bias, store.
// fp32 in every engine mode: a flipped top-1 is a discrete error.
#include "common.h"
#include "kernels.h"
#include "moe_gate.h"

namespace m3 {

template <int NT>
__global__ __launch_bounds__(256, 2) void moe_router_kernel(const float* __restrict__ emb, int lde, int De,
                                                            const float* __restrict__ x, int ldx, int D,
                                                            const float* __restrict__ W, const float* __restrict__ bias,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float* __restrict__ xn, int ldxn,
                                                            float* __restrict__ Y, int ldy, int M, int N,
                                                            const int32_t* __restrict__ m_dev,
                                                            int32_t* __restrict__ gate_idx, float* __restrict__ gate_val,
                                                            const int32_t* __restrict__ row_len, int rows_per_batch,
                                                            unsigned char* __restrict__ xq, float* __restrict__ xq_scale) {
  extern __shared__ __attribute__((aligned(16))) float rt_lds[];
  // xq != null (D == 512): the normalised rows also leave as e4m3 with a per-row scale -- exactly the quantisation the fused fp8
  // expert kernel applies to its input rows (moe_expert_fused_fp8.hip: amax / 448, reciprocal, saturating conversion; the
  // conversions need MODE.FP16_OVFL, which touches nothing else in this kernel), so that kernel reads 512 B per row instead of 2 KB
  if (xq != nullptr) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 16;
  if (m_dev != nullptr && m0 > *m_dev) {              // packed ragged batch: no live row in this tile
    if (gate_idx != nullptr && blockIdx.y == 0 && threadIdx.x < 16 && m0 + (int)threadIdx.x < M) {   // (its rows are "padding" for the gate)
      gate_idx[m0 + threadIdx.x] = -1;
      gate_val[m0 + threadIdx.x] = 0.f;
    }
    return;
  }
  // short inputs have few row tiles: the expert columns are then split over blockIdx.y (each work-group normalises its rows
  // itself -- 16 rows, cheap -- and only column group 0 writes xn), so that more CUs pull W
  const int col_group = blockIdx.y, n_base = col_group * 16 * NT;
  const int e_ld = De + 8, x_ld = D + 8;
  float* Es = rt_lds;                                  // [16][De + 8]
  float* Xs = rt_lds + 16 * e_ld;                      // [16][D + 8]
  const int K = De + D;

  // ---- W fragments: lane (col, kq) holds W[16 t + col][k + 4 kq .. + 3]; 4 k-steps of 16 per group, 2 groups in flight ----
  // wave w owns k in [w Kh / 4, (w + 1) Kh / 4) of each half (Kh = De, then D)
  const int steps_e = De >> 6, steps_x = D >> 6;       // k-steps of 16 per wave in each half
  const int total = steps_e + steps_x;
  const float* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wrow[t] = W + (size_t)min(n_base + 16 * t + col, N - 1) * K + 4 * kq;
  auto kbase = [&](int s) -> int {                     // k index (in the concatenated row) of this wave's step s
    return s < steps_e ? wave * (De >> 2) + 16 * s : De + wave * (D >> 2) + 16 * (s - steps_e);
  };
  constexpr int G = 4;
  f32x4 wb[2][G][NT];
  auto load_group = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int s = min(G * g + i, total - 1);         // clamped, not branched
      const int k = kbase(s);
#pragma unroll
      for (int t = 0; t < NT; ++t) wb[buf][i][t] = ldg4(wrow[t] + k);
    }
  };
  load_group(0, 0);

  // ---- phase 1: LayerNorm of 4 rows per wave (rows in registers), embed rows to LDS.  All 16 row loads of the wave are
  //      issued before the first reduction: one memory round trip, not one per row ----
  f32x4 v[4][2], ev[4][2];                             // D, De <= 512 per 64 lanes x 2 float4 (wider rows: second pass below)
  const bool wide = D > 512 || De > 512;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = min(m0 + 4 * wave + i, M - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 4 * lane + 256 * j;
      // loads are never branched around (clamped address, value zeroed afterwards): straight-line code, one batch of loads
      const f32x4 xv = ldg4(x + (size_t)m * ldx + min(c, D - 4));
      const f32x4 ee = ldg4(emb + (size_t)m * lde + min(c, De - 4));
      const float onx = c < D ? 1.f : 0.f, one = c < De ? 1.f : 0.f;
      v[i][j] = f32x4{xv[0] * onx, xv[1] * onx, xv[2] * onx, xv[3] * onx};
      ev[i][j] = f32x4{ee[0] * one, ee[1] * one, ee[2] * one, ee[3] * one};
    }
  }
  f32x4 ga[2], be[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = min(4 * lane + 256 * j, D - 4);
    ga[j] = ldg4(gamma + c);
    be[j] = ldg4(beta + c);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * wave + i;
    const int m = min(m0 + r, M - 1);
    const bool live = (m0 + r) < M && (col_group == 0);
    f32x4 hi[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float s = (v[i][0][0] + v[i][0][1]) + (v[i][0][2] + v[i][0][3]) + (v[i][1][0] + v[i][1][1]) + (v[i][1][2] + v[i][1][3]);
    if (wide) {                                        // columns 512 .. 1023
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = 512 + 4 * lane + 256 * j;
        if (c < D) {
          hi[j] = ldg4(x + (size_t)m * ldx + c);
          s += (hi[j][0] + hi[j][1]) + (hi[j][2] + hi[j][3]);
        }
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d0 = v[i][j][e] - mean;
        q += (4 * lane + 256 * j < D) ? d0 * d0 : 0.f;
        const float d1 = hi[j][e] - mean;
        q += (wide && 512 + 4 * lane + 256 * j < D) ? d1 * d1 : 0.f;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    f32x4 o2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 4 * lane + 256 * j;
      if (c < D) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][j][e] - mean) * rstd * ga[j][e] + be[j][e];
        *reinterpret_cast<f32x4*>(Xs + r * x_ld + c) = o;
        if (live && xn != nullptr) stg4(xn + (size_t)m * ldxn + c, o);
        o2[j] = o;
      }
    }
    if (xq != nullptr) {                               // (D == 512: the lane's 8 values are the whole of its share of the row)
      float amax = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(o2[0][e]), fabsf(o2[1][e])));
      amax = fmaxf(wave_max(amax), 1e-30f);
      const float inv = 448.f * __builtin_amdgcn_rcpf(amax);
      int q0 = 0, q1 = 0;
      q0 = __builtin_amdgcn_cvt_pk_fp8_f32(o2[0][0] * inv, o2[0][1] * inv, q0, false);
      q0 = __builtin_amdgcn_cvt_pk_fp8_f32(o2[0][2] * inv, o2[0][3] * inv, q0, true);
      q1 = __builtin_amdgcn_cvt_pk_fp8_f32(o2[1][0] * inv, o2[1][1] * inv, q1, false);
      q1 = __builtin_amdgcn_cvt_pk_fp8_f32(o2[1][2] * inv, o2[1][3] * inv, q1, true);
      if (live) {
        *reinterpret_cast<int*>(xq + (size_t)m * 512 + 4 * lane) = q0;
        *reinterpret_cast<int*>(xq + (size_t)m * 512 + 256 + 4 * lane) = q1;
        if (lane == 0) xq_scale[m] = amax * (1.f / 448.f);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 4 * lane + 256 * j;
      if (c < De) *reinterpret_cast<f32x4*>(Es + r * e_ld + c) = ev[i][j];
      if (wide) {
        const int c2 = 512 + c;
        if (c2 < D) {
          const f32x4 g2 = ldg4(gamma + c2), b2 = ldg4(beta + c2);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (hi[j][e] - mean) * rstd * g2[e] + b2[e];
          *reinterpret_cast<f32x4*>(Xs + r * x_ld + c2) = o;
          if (live && xn != nullptr) stg4(xn + (size_t)m * ldxn + c2, o);
        }
        if (c2 < De) *reinterpret_cast<f32x4*>(Es + r * e_ld + c2) = ldg4(emb + (size_t)m * lde + c2);
      }
    }
  }
  __syncthreads();

  // ---- phase 2: this wave's quarter of K, both halves ----
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute_group = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int s = G * g + i;
      if (s < total) {
        const bool second = s >= steps_e;
        const float* src = second ? Xs + col * x_ld + wave * (D >> 2) + 16 * (s - steps_e) + 4 * kq
                                  : Es + col * e_ld + wave * (De >> 2) + 16 * s + 4 * kq;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[t] = mfma16(a[j], wb[buf][i][t][j], acc[t]);
      }
    }
  };
  const int ngroups = (total + G - 1) / G;
  for (int g = 0; g < ngroups; g += 2) {
    if (g + 1 < ngroups) load_group(g + 1, 1);
    compute_group(g, 0);
    if (g + 1 < ngroups) {
      if (g + 2 < ngroups) load_group(g + 2, 0);
      compute_group(g + 1, 1);
    }
  }

  // ---- phase 3: fixed-order sum of the four K-partials, bias, store ----
  __syncthreads();                                     // every wave is done with Es / Xs: reuse as the reduction buffer
  float* red = rt_lds;                                 // [4][NT][256]
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * NT + t) * 256 + r * 64 + lane] = acc[t][r];
  __syncthreads();
  for (int t = wave; t < NT; t += 4) {
    const int n = n_base + 16 * t + col;
    const float b = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sum = (red[(0 * NT + t) * 256 + r * 64 + lane] + red[(1 * NT + t) * 256 + r * 64 + lane]) +
                        (red[(2 * NT + t) * 256 + r * 64 + lane] + red[(3 * NT + t) * 256 + r * 64 + lane]);
      const int m = m0 + 4 * kq + r;
      if (m < M && n < N) Y[(size_t)m * ldy + n] = sum + b;
      if (gate_idx != nullptr) red[4 * NT * 256 + (4 * kq + r) * (16 * NT) + 16 * t + col] = sum + b;   // logits tile [16][16 NT]
    }
  }
  // ---- phase 4 (gate_idx != null; this work-group holds ALL N = 16 NT columns): SoftmaxTopK on the 16 rows, one lane each
  //      (softmax_topk_kernel.cu:26-120: the reference's arg-max tree, value = 1 / sum exp(x - max); padded frames idx -1 /
  //      value 0) -- the row-parallel top-1 launch behind the router disappears ----
  if (gate_idx != nullptr) {
    __syncthreads();
    if (threadIdx.x < 16) {
      const int m = m0 + threadIdx.x;
      if (m < M) {
        int gi = -1;
        float gv = 0.f;
        const bool live = row_len == nullptr || (m % rows_per_batch) < row_len[m / rows_per_batch];
        if (live) gate_top1_lane<16 * NT>(red + 4 * NT * 256 + threadIdx.x * (16 * NT), &gi, &gv);
        gate_idx[m] = gi;
        gate_val[m] = gv;
      }
    }
  }
}

bool moe_router_fuses_top1(int N) { return N == 16 || N == 32 || N == 64; }   // whole rows in one work-group, N = 16 NT

bool moe_router_supports(int De, int D, int N) {
  return N >= 1 && N <= 64 && (De & 63) == 0 && (D & 63) == 0 && De >= 64 && D >= 64 && De <= 1024 && D <= 1024;
}

#ifndef M3_ROUTER_LDS_FLOOR
#define M3_ROUTER_LDS_FLOOR 0
#endif
constexpr size_t kRouterLdsFloor = M3_ROUTER_LDS_FLOOR;
static size_t router_lds_bytes(int De, int D, int NT) {
  const size_t tiles = (size_t)16 * (De + 8 + D + 8) * 4, red = (size_t)(4 * NT * 256 + 16 * 16 * NT) * 4;   // (+ the logits tile of phase 4)
  return tiles > red ? tiles : red;
}

int init_moe_router_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
  const int big = 160 * 1024;
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)moe_router_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)moe_router_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)moe_router_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  once.mark();
  return 0;
}

int launch_moe_router(const float* emb, int lde, int De, const float* x, int ldx, int D, const float* W, const float* bias,
                      const float* gamma, const float* beta, float eps, float* xn, int ldxn, float* Y, int ldy, int M, int N,
                      const int32_t* m_dev, hipStream_t stream, int32_t* gate_idx, float* gate_val, const int32_t* row_len,
                      int rows_per_batch, void* xq, float* xq_scale) {
  M3_REQUIRE(xq == nullptr || (D == 512 && xq_scale != nullptr), "moe_router: quantised rows need D == 512 and a scale buffer");
  M3_REQUIRE(M > 0 && moe_router_supports(De, D, N), "moe_router: unsupported problem M=%d De=%d D=%d N=%d", M, De, D, N);
  M3_REQUIRE((lde & 3) == 0 && (ldx & 3) == 0 && (xn == nullptr || (ldxn & 3) == 0), "moe_router: row strides must be multiples of 4");
  M3_REQUIRE(emb && x && W && gamma && beta && Y, "moe_router: null pointer");
  if (int rc = init_moe_router_kernels()) return rc;
  // column tiles per work-group: all of them (A read once) when the row tiles alone fill the chip, fewer for short inputs
  const int tiles = cdiv(N, 16), rows16 = cdiv(M, 16);
  int nt = tiles <= 1 ? 1 : (tiles == 2 ? 2 : 4);
  const bool top1 = gate_idx != nullptr;
  if (top1) {   // SoftmaxTopK in the kernel's tail: the work-group must hold whole logits rows, i.e. all column tiles
    M3_REQUIRE(moe_router_fuses_top1(N) && gate_val != nullptr && (row_len == nullptr || rows_per_batch > 0),
               "moe_router: fused top-1 needs 16 / 32 / 64 experts (got %d) and a gate_value buffer", N);
    nt = tiles;
  } else {
    while (nt > 1 && (long)rows16 * cdiv(tiles, nt) < 192) nt >>= 1;
  }
  // Two work-groups per CU.  (Round 3: under concurrent execution contexts this kernel returned, about once in 100 forwards,
  // a row whose LayerNorm mean was wrong -- only beside two particular LDS-DMA GEMM launches of ANOTHER context, only when
  // built with packed-FP32 VALU instructions.  Symptom eliminated by building the library without them (Makefile NOPK,
  // asserted on the built code by tools/check_device_isa.py); the mechanism is open, DESIGN.md 10.8.
  // -DM3_ROUTER_LDS_FLOOR=98304 restores the structural guard, one work-group per CU.)
  size_t lds = router_lds_bytes(De, D, nt);
  if (lds < kRouterLdsFloor) lds = kRouterLdsFloor;
  dim3 grid(rows16, cdiv(tiles, nt));
#define M3_ROUTER_CASE(NT_)                                                                                              \
  hipLaunchKernelGGL((moe_router_kernel<NT_>), grid, dim3(256), lds, stream, emb, lde, De, x, ldx, D, W, bias, gamma, beta, \
                     eps, xn, ldxn, Y, ldy, M, N, m_dev, gate_idx, gate_val, row_len, rows_per_batch, (unsigned char*)xq, xq_scale)
  if (nt == 1) M3_ROUTER_CASE(1); else if (nt == 2) M3_ROUTER_CASE(2); else M3_ROUTER_CASE(4);
#undef M3_ROUTER_CASE
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
