// Convolution pieces of the hot path that are not GEMM-shaped (fp32, channel-last).
//
//   dwconv_ln_silu_kernel : ConvolutionModule's depthwise Conv1d(k=15, groups=C) + LayerNorm(eps 1e-5)
//                           + SiLU (trainer_3m_fix/layer/convolution.py:134-152; TRT conv
//                           torch_network_helper.py:199-225, LayerNorm plugin, SiLU :827-841) fused,
//                           on (B,T',C) rows so the two "use_layer_norm_trans" shuffles disappear.
//   conv1_relu_kernel     : first subsampling Conv2d(1,C,3,stride 2) + ReLU
//                           (trainer_3m_fix/layer/subsampling.py:113-114) writing channel-last
//                           (B,T1,F1,C) so the second conv becomes an implicit GEMM (gemm.hip).
//   depthwise_conv1d_nct  : plain (B,C,T) depthwise conv for the op-by-op network_helper path.
#include "common.h"
#include "kernels.h"

namespace m3 {

// z, out: [B*T][D]; w_kc: [K][D] (repacked from (D,1,K)); one wave per frame.
template <int NV>
__global__ __launch_bounds__(256) void dwconv_ln_silu_kernel(const float* __restrict__ z, const float* __restrict__ w_kc,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int T,
                                                             int D, int K, float* __restrict__ out, int rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int b = row / T, t = row % T;
  const int pad = (K - 1) / 2;
  f32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = (c < D) ? ldg4(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // taps in chunks of 5: all loads of a chunk are issued before its FMAs (edge taps read a clamped row
  // and are multiplied by 0 = the conv's zero padding), so a frame costs ~3 memory round trips, not K
  for (int k0 = 0; k0 < K; k0 += 5) {
    f32x4 zz[5][NV], ww[5][NV];
    float on[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
      const int k = min(k0 + kk, K - 1);
      const int tt = t + k - pad;
      on[kk] = (k0 + kk < K && tt >= 0 && tt < T) ? 1.f : 0.f;
      const float* zr = z + ((size_t)b * T + min(max(tt, 0), T - 1)) * D;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
          zz[kk][i] = ldg4(zr + c);
          ww[kk][i] = ldg4(w_kc + (size_t)k * D + c);
        }
      }
    }
#pragma unroll
    for (int kk = 0; kk < 5; ++kk)
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if ((lane + 64 * i) * 4 < D) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[i][j] = fmaf(zz[kk][i][j] * on[kk], ww[kk][i][j], v[i][j]);
        }
  }
  float mean = 0.f, rstd = 1.f;
  if (gamma != nullptr) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if ((lane + 64 * i) * 4 < D) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if ((lane + 64 * i) * 4 < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = v[i][j] - mean;
          q += d * d;
        }
      }
    rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
      f32x4 o = v[i];
      if (gamma != nullptr) {
        const f32x4 g = ldg4(gamma + c), be = ldg4(beta + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (o[j] - mean) * rstd * g[j] + be[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = silu(o[j]);
      stg4(out + (size_t)row * D + c, o);
    }
  }
}

int launch_dwconv_ln_silu(const float* z, const float* w_kc, const float* bias, const float* gamma,
                          const float* beta, float eps, int B, int T, int D, int K, float* out, hipStream_t stream) {
  M3_REQUIRE((D & 3) == 0 && D <= 2048, "dwconv: channels=%d must be a multiple of 4 (<=2048)", D);
  M3_REQUIRE((K & 1) == 1, "dwconv: kernel size %d must be odd (non-causal)", K);
  const int rows = B * T;
  if (rows == 0) return 0;
  const int nv = cdiv(D, 256);
  dim3 grid(cdiv(rows, 4));
#define M3_DW_CASE(NV_)                                                                                          \
  hipLaunchKernelGGL((dwconv_ln_silu_kernel<NV_>), grid, dim3(256), 0, stream, z, w_kc, bias, gamma, beta, eps, T, \
                     D, K, out, rows)
  if (nv <= 1) M3_DW_CASE(1); else if (nv <= 2) M3_DW_CASE(2); else if (nv <= 4) M3_DW_CASE(4); else M3_DW_CASE(8);
#undef M3_DW_CASE
  M3_LAUNCH_CHECK();
  return 0;
}

// feat [B][T][idim] -> out [B][T1][F1][C] (channel-last), w9c [9][C] (repacked from (C,1,3,3)), ReLU.
__global__ __launch_bounds__(256) void conv1_relu_kernel(const float* __restrict__ feat, const float* __restrict__ w9c,
                                                         const float* __restrict__ bias, int T, int idim, int T1,
                                                         int F1, int C, float* __restrict__ out, size_t n4) {
  const int c4n = C >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    size_t rem = i / c4n;
    const int f1 = (int)(rem % F1);
    rem /= F1;
    const int t1 = (int)(rem % T1);
    const int b = (int)(rem / T1);
    f32x4 acc = ldg4(bias + c);
    const float* base = feat + ((size_t)b * T + 2 * t1) * idim + 2 * f1;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float xv = base[kh * idim + kw];
        const f32x4 w = ldg4(w9c + (kh * 3 + kw) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv, w[j], acc[j]);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
    stg4(out + i * 4, acc);
  }
}

int launch_conv1_relu(const float* feat, const float* w9c, const float* bias, int B, int T, int idim, int C,
                      float* out, hipStream_t stream) {
  M3_REQUIRE(T >= 3 && idim >= 3, "subsampling: input (T=%d, idim=%d) shorter than the 3x3 kernel", T, idim);
  M3_REQUIRE((C & 3) == 0, "subsampling: channels=%d must be a multiple of 4", C);
  const int T1 = (T - 3) / 2 + 1, F1 = (idim - 3) / 2 + 1;
  const size_t n4 = (size_t)B * T1 * F1 * (C / 4);
  hipLaunchKernelGGL(conv1_relu_kernel, dim3(grid1d(n4, 4096)), dim3(256), 0, stream,
                     feat, w9c, bias, T, idim, T1, F1, C, out, n4);
  M3_LAUNCH_CHECK();
  return 0;
}

__global__ void depthwise_conv1d_nct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                            const float* __restrict__ bias, int C, int T, int K, int pad,
                                            float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int c = (int)((i / T) % C);
    const float* xr = x + (i - t);
    float acc = bias ? bias[c] : 0.f;
    for (int k = 0; k < K; ++k) {
      const int tt = t + k - pad;
      if (tt >= 0 && tt < T) acc = fmaf(xr[tt], w[(size_t)c * K + k], acc);
    }
    y[i] = acc;
  }
}

int launch_depthwise_conv1d_nct(const float* x, const float* w, const float* bias, int B, int C, int T, int K,
                                int pad, float* y, hipStream_t stream) {
  const size_t n = (size_t)B * C * T;
  if (n == 0) return 0;
  hipLaunchKernelGGL(depthwise_conv1d_nct_kernel, dim3(grid1d(n, 4096)), dim3(256), 0,
                     stream, x, w, bias, C, T, K, pad, y, n);
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
