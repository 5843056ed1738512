// Grouped expert FFN with bf16 weights (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// The low-precision form of moe_expert.hip: the reference's FMoEExpertPlugin declares a half-precision mode
// (`data_type` field, fmoe_expert_plugin.cpp:331-354) but asserts on it (:264-266); here it is implemented with
// bf16 as the 16-bit type.  Same decomposition as the fp32 kernel -- work item = (expert, 64-wide hidden slice),
// token rows gathered into LDS, H kept in LDS, partial outputs to slab[slice] and a fixed-order combine -- with
//   * W1 [E][F][D] and W2 (slice-major [E][F/64][D][64] or reference [E][D][F]) stored bf16: 2.1 MB per touched expert,
//   * token rows rounded to bf16 once while they are staged in LDS, H = SiLU(. + b1) rounded to bf16 in LDS,
//   * b1 / b2 / gate / residual / LayerNorm in fp32 (b2 .. LayerNorm live in moe_combine_kernel, shared with fp32).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace m3 {

template <int MT>
__global__ __launch_bounds__(64 * (kExpertSliceW16 / 16)) void expert_ffn_bf16w_kernel(
    const float* __restrict__ x, int ldx, const int32_t* __restrict__ pos, const int32_t* __restrict__ acc_hist, int S,
    int D, int F, const bf16_t* __restrict__ w1, const float* __restrict__ b1, const bf16_t* __restrict__ w2,
    int w2_row_stride, int w2_slice_stride, float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int e = blockIdx.y, slice = blockIdx.x;
  const int row_lo = acc_hist[e], row_hi = acc_hist[e + 1];
  if (row_hi <= row_lo) return;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int xs_ld = D + 8;                    // bf16 elements; +16 B keeps the 16-B fragment reads spread over banks
  constexpr int hs_ld = kExpertSliceW16 + 8;
  constexpr int NWV = kExpertSliceW16 / 16;      // 4 waves, 16 hidden units each
  constexpr int KS2 = kExpertSliceW16 / 32;      // 32-deep k-steps of phase 2 (2)
  constexpr int SPG = 8 / KS2;                // phase-2 output tiles per 8-load group (4)
  static_assert(kExpertSliceW16 == 64, "bf16 expert kernel is laid out for 64-wide slices");
  bf16_t* xs = reinterpret_cast<bf16_t*>(lds_raw);   // [16*MT][D+8]
  bf16_t* hs = xs + 16 * MT * xs_ld;                 // [16*MT][64+8]
  const int f0 = slice * kExpertSliceW16;
  const int ksteps1 = D >> 5;

  const bf16_t* w1row = w1 + ((size_t)e * F + f0 + 16 * wave + col) * D + 8 * kq;
  const float bias1 = b1[(size_t)e * F + f0 + 16 * wave + col];
  const int nsub = D >> 4;
  const bf16_t* w2_slice = w2 + (size_t)e * D * F + (size_t)slice * w2_slice_stride;

  const int g1 = (ksteps1 + 7) >> 3;
  const int g2 = ((nsub + NWV - 1) / NWV + SPG - 1) / SPG;
  const int total = g1 + g2;

  // an expert's row tiles are spread over blockIdx.z (long batches, unbalanced routing: no serial tile loop)
  for (int r0 = row_lo + 16 * MT * blockIdx.z; r0 < row_hi; r0 += 16 * MT * gridDim.z) {
    const int nrows = min(16 * MT, row_hi - r0);
    float* slab_base = slab + ((size_t)slice * S + r0) * D;

    bf16x8 wb[2][8];
    auto load_group = [&](int g, int buf) {
      if (g < g1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int s = min(8 * g + i, ksteps1 - 1);
          wb[buf][i] = ldg8h_w(w1row + (s << 5));
        }
      } else {
#pragma unroll
        for (int j = 0; j < SPG; ++j) {
          const int sub = min(wave + NWV * (SPG * (g - g1) + j), nsub - 1);
          const bf16_t* p = w2_slice + (size_t)(16 * sub + col) * w2_row_stride + 8 * kq;
#pragma unroll
          for (int st = 0; st < KS2; ++st) wb[buf][KS2 * j + st] = ldg8h_w(p + 32 * st);
        }
      }
    };
    load_group(0, 0);

    // ---- gather token rows into LDS, rounded to bf16 (fused local_scatter) ----
    __syncthreads();
    for (int i = wave; i < 16 * MT; i += NWV) {
      bf16_t* dst = xs + i * xs_ld;
      if (i < nrows) {
        const float* src = x + (size_t)pos[r0 + i] * ldx;
        for (int c = lane * 8; c < D; c += 512)
          *reinterpret_cast<bf16x8*>(dst + c) = cvt8(ldg4(src + c), ldg4(src + c + 4));
      } else {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.f;
        for (int c = lane * 8; c < D; c += 512) *reinterpret_cast<bf16x8*>(dst + c) = z;
      }
    }
    __syncthreads();

    f32x4 acc1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 hfrag[MT][KS2];

    auto transition = [&]() {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hs[(16 * mt + 4 * kq + r) * hs_ld + 16 * wave + col] = (bf16_t)silu(acc1[mt][r] + bias1);
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int st = 0; st < KS2; ++st)
          hfrag[mt][st] = *reinterpret_cast<const bf16x8*>(hs + (16 * mt + col) * hs_ld + 32 * st + 8 * kq);
    };
    auto compute = [&](int g, int buf) {
      if (g < g1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int s = 8 * g + i;
          if (s < ksteps1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const bf16x8 a = *reinterpret_cast<const bf16x8*>(xs + (16 * mt + col) * xs_ld + (s << 5) + 8 * kq);
              acc1[mt] = mfma16h(a, wb[buf][i], acc1[mt]);
            }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < SPG; ++j) {
          const int sub = wave + NWV * (SPG * (g - g1) + j);
          if (sub < nsub) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int st = 0; st < KS2; ++st) acc2 = mfma16h(hfrag[mt][st], wb[buf][KS2 * j + st], acc2);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = 16 * mt + 4 * kq + r;
                if (i < nrows) slab_base[(size_t)i * D + 16 * sub + col] = acc2[r];
              }
            }
          }
        }
      }
    };

    for (int g0 = 0; g0 < total; g0 += 2) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int g = g0 + b;
        if (g < total) {
          if (g + 1 < total) load_group(g + 1, b ^ 1);
          if (g == g1) transition();
          compute(g, b);
        }
      }
    }
  }
}

// Long batches run as two grouped GEMMs on the LDS-tiled core instead (gemm_bf16_tiled.hip): the slab form writes and
// re-reads F/64 partial outputs per row (32 KB per row), which bounds it to ~65 TFLOP/s however many rows an expert has.
// The tiled form keeps H (bf16, S*F) and the sorted output rows (fp32, S*D) in the SAME workspace region the slabs use.
bool expert_ffn_bf16_tiled(int S, int E, int D, int F) {
  static const int min_rows = [] {
    const char* e = getenv("M3_EXPERT_TILED_MIN_ROWS");
    return e ? atoi(e) : 1024;
  }();
  const size_t slab = expert_ffn_slab_bytes(S, D, F);
  const size_t need = align_up((size_t)S * F * 2, 256) + (size_t)S * D * 4;
  return S >= min_rows && (D & 127) == 0 && (F & 127) == 0 && need <= slab;
}
float* expert_ffn_bf16_rows(float* slab, int S, int E, int D, int F) {   // what moe_combine reads
  return expert_ffn_bf16_tiled(S, E, D, F) ? (float*)((char*)slab + align_up((size_t)S * F * 2, 256)) : slab;
}
int expert_ffn_bf16_slices(int S, int E, int D, int F) { return expert_ffn_bf16_tiled(S, E, D, F) ? 1 : F / kExpertSliceW16; }

// fp8 arithmetic (wmode 3) takes the fused one-kernel form where it applies (its result: fsplit slabs of sorted rows at the
// start of the slab region); bf16 and fp8 weight-only experts keep the layouts above
static int w16_fsplit(int wmode, int S, int E, int D, int F) { return expert_ffn_fused_fp8_fsplit(S, E, D, F); }
static bool w16_fused(int wmode, int S, int E, int D, int F) {
  return wmode == 3 && expert_ffn_fused_fp8_applies(S, E, D, F) &&
         (size_t)w16_fsplit(wmode, S, E, D, F) * S * D * 4 <= expert_ffn_slab_bytes(S, D, F);
}
bool expert_ffn_w8a8_fused(int S, int E, int D, int F) { return w16_fused(3, S, E, D, F); }
float* expert_ffn_w16_rows(int wmode, float* slab, int S, int E, int D, int F) {
  return w16_fused(wmode, S, E, D, F) ? slab : expert_ffn_bf16_rows(slab, S, E, D, F);
}
int expert_ffn_w16_slices(int wmode, int S, int E, int D, int F) {
  return w16_fused(wmode, S, E, D, F) ? w16_fsplit(wmode, S, E, D, F) : expert_ffn_bf16_slices(S, E, D, F);
}
int expert_ffn_w16_launches(int wmode, int S, int E, int D, int F) {
  if (w16_fused(wmode, S, E, D, F)) return 1;
  if (wmode == 1 && expert_ffn_bf16_tiled(S, E, D, F) && expert_ffn_bf16_g256(S, E, D, F)) return 3;   // rows -> bf16, GEMM-1, GEMM-2
  return expert_ffn_bf16_tiled(S, E, D, F) ? 2 : 1;
}
const char* expert_ffn_w16_kernel(int wmode, int S, int E, int D, int F) {
  if (w16_fused(wmode, S, E, D, F)) return "expert_ffn_fused_fp8_kernel";
  if (wmode == 3) wmode = 2;
  if (wmode == 1 && expert_ffn_bf16_tiled(S, E, D, F) && expert_ffn_bf16_g256(S, E, D, F)) return "expert_gemm_g256_kernel";
  if (expert_ffn_bf16_tiled(S, E, D, F)) return wmode == 2 ? "gemm_bf16w_tiled_kernel<grouped,fp8>" : "gemm_bf16w_tiled_kernel<grouped>";
  return wmode == 2 ? "expert_ffn_w8_kernel" : "expert_ffn_bf16w_kernel";
}

int init_expert_ffn_bf16_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_bf16w_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  once.mark();
  return 0;
}

int launch_expert_ffn_bf16w(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D,
                            int F, const void* w1, const float* b1, const void* w2, int w2_sliced, float* slab,
                            hipStream_t stream, const float* b2, float* y_scatter) {
  M3_REQUIRE(S > 0 && E > 0, "expert_ffn_bf16w: empty problem S=%d E=%d", S, E);
  M3_REQUIRE((D & 31) == 0 && D <= 2048, "expert_ffn_bf16w: idim=%d must be a multiple of 32 (<=2048)", D);
  M3_REQUIRE(F % kExpertSliceW16 == 0, "expert_ffn_bf16w: hidden_units=%d must be a multiple of %d", F, kExpertSliceW16);
  M3_REQUIRE((ldx & 3) == 0, "expert_ffn_bf16w: ldx=%d must be a multiple of 4", ldx);
  if (expert_ffn_bf16_tiled(S, E, D, F))
    return launch_expert_ffn_bf16w_tiled(x, ldx, pos, acc_hist, S, E, D, F, w1, b1, w2, w2_sliced, slab,
                                         expert_ffn_bf16_rows(slab, S, E, D, F), stream, b2, y_scatter);
  M3_REQUIRE(y_scatter == nullptr, "expert_ffn_bf16w: the scattering epilogue exists in the tiled form only (S=%d E=%d)", S, E);
  const int mt = S <= 64 ? 1 : (S <= 512 ? 2 : 4);
  const size_t lds_bytes = (size_t)16 * mt * ((D + 8) + (kExpertSliceW16 + 8)) * sizeof(bf16_t);
  M3_REQUIRE(lds_bytes <= 160 * 1024, "expert_ffn_bf16w: LDS tile of %zu bytes does not fit", lds_bytes);
  int zt = cdiv(S, 16 * mt);
  dim3 grid(F / kExpertSliceW16, E, zt < 8 ? zt : 8);
  const int w2_row_stride = w2_sliced ? kExpertSliceW16 : F;
  const int w2_slice_stride = w2_sliced ? D * kExpertSliceW16 : kExpertSliceW16;
  if (int rc = init_expert_ffn_bf16_kernels()) return rc;
#define M3_EXPERT_CASE(MT_)                                                                                       \
  hipLaunchKernelGGL((expert_ffn_bf16w_kernel<MT_>), grid, dim3(64 * (kExpertSliceW16 / 16)), lds_bytes, stream, x, \
                     ldx, pos, acc_hist, S, D, F, (const bf16_t*)w1, b1, (const bf16_t*)w2, w2_row_stride,        \
                     w2_slice_stride, slab)
  if (mt == 1) M3_EXPERT_CASE(1); else if (mt == 2) M3_EXPERT_CASE(2); else M3_EXPERT_CASE(4);
#undef M3_EXPERT_CASE
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
