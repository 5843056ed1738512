// Row-wise / element-wise operators of the hot path (HBM- or latency-bound; fp32).
//
// Plugin kernels of the reference they replace (TRTAPI++/plugin/...):
//   layernorm_kernel          layer_norm_plugin/layer_norm_kernel.cu:33-162  (adds eps, two-pass variance)
//   softmax_top1_kernel       softmax_topk_plugin/softmax_topk_kernel.cu:26-120
//   att_masked_softmax_kernel att_masked_softmax_plugin/att_masked_softmax_kernel.cu:224-272, common/common.cuh:264-360
//   masked_fill_kernel        masked_fill_plugin/masked_fill_kernel.cu:27-54
//   glu_kernel                glu_plugin/glu_kernel.cu:26-44
//   mask_conv2d_sample_kernel mask_conv2d_sample_plugin/mask_conv2d_sample_kernel.cu:29-43
//   scale (x*sqrt(D))         rel_positional_encoding_plugin/rel_positional_encoding_kernel.cu:62-81
// and the TensorRT-native element-wise / shuffle / concat / softmax / matmul layers used through
// network_helper (trt_network_helper.py:87-114, tensor_network_helper.py:209-282,406-471,
// torch_network_helper.py:736-745,827-866).
#include "common.h"
#include "kernels.h"

namespace m3 {

// ---------------------------------------------------------------- LayerNorm (one wave per row)
// gamma2 != null: a SECOND LayerNorm applied to the first one's result, y2 = LN2(LN1(x)) -- the embed encoder ends in
// norm_final of its last block followed by after_norm (conformer_embed_domain_acc.py:171-181): one launch, same arithmetic
// (LN2's statistics are taken from the fp32 values LN1 stores).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* y, int rows, int D, bf16_t* yb, float* ystats,
                                                        const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                        float eps2, float* __restrict__ y2) {  // y may alias x (row-local)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * D;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      v[i] = ldg4(xr + c);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
      const f32x4 g = ldg4(gamma + c), b = ldg4(beta + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      stg4(y + (size_t)row * D + c, o);
      v[i] = o;                                       // (kept for the optional second LayerNorm)
      if (yb != nullptr) {                            // bf16 copy for the next GEMMs' A operand
        bf16x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          h[j] = (bf16_t)o[j];
          const float f = (float)h[j];
          t1 += f;
          t2 += f * f;
        }
        *reinterpret_cast<bf16x4*>(yb + (size_t)row * D + c) = h;
      }
    }
  }
  if (ystats != nullptr) {         // row statistics of the bf16 copy (kernels.h: Yb_stats): total in part 0, zeros elsewhere
    t1 = wave_sum(t1);
    t2 = wave_sum(t2);
    if (lane < 2 * kXbStatParts) ystats[(size_t)row * 2 * kXbStatParts + lane] = lane == 0 ? t1 : (lane == 1 ? t2 : 0.f);
  }
  if (gamma2 != nullptr) {
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < D) s2 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean2 = wave_sum(s2) / (float)D;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = v[i][j] - mean2;
          q2 += d * d;
        }
      }
    }
    const float rstd2 = rsqrtf(wave_sum(q2) / (float)D + eps2);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < D) {
        const f32x4 g = ldg4(gamma2 + c), b = ldg4(beta2 + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean2) * rstd2 * g[j] + b[j];
        stg4(y2 + (size_t)row * D + c, o);
      }
    }
  }
}

// (sum, sum of squares) of every row of a bf16 matrix (one wave per row): the statistics a folded-LayerNorm GEMM on the
// LDS-DMA kernel needs when the operand's producer could not leave them (subsampling Linear + row packing)
__global__ __launch_bounds__(256) void row_stats_bf16_kernel(const bf16_t* __restrict__ xb, int rows, int D, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  float t1 = 0.f, t2 = 0.f;
  for (int c = lane * 8; c < D; c += 512) {
    const bf16x8 h = ldg8h(xb + (size_t)row * D + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)h[j];
      t1 += f;
      t2 += f * f;
    }
  }
  t1 = wave_sum(t1);
  t2 = wave_sum(t2);
  if (lane < 2 * kXbStatParts) stats[(size_t)row * 2 * kXbStatParts + lane] = lane == 0 ? t1 : (lane == 1 ? t2 : 0.f);
}
int launch_row_stats_bf16(const void* xb, int rows, int D, float* stats, hipStream_t stream) {
  M3_REQUIRE((D & 7) == 0, "row_stats: D=%d must be a multiple of 8", D);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(row_stats_bf16_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, (const bf16_t*)xb, rows, D, stats);
  M3_LAUNCH_CHECK();
  return 0;
}

int launch_layernorm(const float* x, const float* gamma, const float* beta, float eps, float* y, int rows, int D,
                     hipStream_t stream, void* y_bf16, float* y_stats, const float* gamma2, const float* beta2, float eps2,
                     float* y2) {
  M3_REQUIRE(gamma2 == nullptr || (beta2 != nullptr && y2 != nullptr), "layernorm: the second LayerNorm needs beta2 and y2");
  M3_REQUIRE((D & 3) == 0 && D <= 2048, "layernorm: dim=%d must be a multiple of 4 (<=2048)", D);
  if (rows == 0) return 0;
  const int nv = cdiv(D, 256);
  dim3 grid(cdiv(rows, 4));
#define M3_LN_CASE(NV_) \
  hipLaunchKernelGGL((layernorm_kernel<NV_>), grid, dim3(256), 0, stream, x, gamma, beta, eps, y, rows, D, (bf16_t*)y_bf16, y_stats, \
                     gamma2, beta2, eps2, y2)
  if (nv <= 1) M3_LN_CASE(1); else if (nv <= 2) M3_LN_CASE(2); else if (nv <= 4) M3_LN_CASE(4); else M3_LN_CASE(8);
#undef M3_LN_CASE
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- softmax + top-1 (router gate)
// One wave per token, lane = expert (E <= 64, power of two).  arg-max = the reference's stride tree
// with strict '<' (ties: lower slot of each pair wins); value = 1 / sum(exp(x - max)).
// Padded frames (t >= len[b]) are written as idx -1 / value 0 (the reference leaves them unwritten).
__global__ __launch_bounds__(256) void softmax_top1_kernel(const float* __restrict__ logits, int ld,
                                                           const int32_t* __restrict__ row_len, int rows_per_batch,
                                                           int S, int E, int32_t* __restrict__ idx,
                                                           float* __restrict__ value) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x * 4 + wave;
  if (s >= S) return;
  if (row_len != nullptr) {
    const int b = s / rows_per_batch, t = s % rows_per_batch;
    if (t >= row_len[b]) {
      if (lane == 0) {
        idx[s] = -1;
        value[s] = 0.f;
      }
      return;
    }
  }
  const float x = (lane < E) ? logits[(size_t)s * ld + lane] : -INFINITY;
  float best = x;
  int bi = lane;
  for (int stride = E >> 1; stride > 0; stride >>= 1) {
    const float ov = __shfl_down(best, stride, 64);
    const int oi = __shfl_down(bi, stride, 64);
    if (lane < stride && best < ov) {
      best = ov;
      bi = oi;
    }
  }
  const float mx = __shfl(best, 0, 64);
  const int mi = __shfl(bi, 0, 64);
  const float ex = (lane < E) ? expf(x - mx) : 0.f;
  const float sum = wave_sum(ex);
  if (lane == 0) {
    idx[s] = mi;
    value[s] = 1.f / sum;
  }
}

int launch_softmax_top1(const float* logits, int ld, const int32_t* row_len, int rows_per_batch, int S, int E,
                        int32_t* idx, float* value, hipStream_t stream) {
  M3_REQUIRE(E >= 1 && E <= 64 && (E & (E - 1)) == 0, "softmax_topk: width=%d must be a power of two <= 64", E);
  M3_REQUIRE(row_len == nullptr || rows_per_batch > 0, "softmax_topk: rows_per_batch missing");
  if (S == 0) return 0;
  hipLaunchKernelGGL(softmax_top1_kernel, dim3(cdiv(S, 4)), dim3(256), 0, stream, logits, ld, row_len,
                     rows_per_batch, S, E, idx, value);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- masked scaled softmax (plugin compat path)
// scores (B,H,T1,T2): p_j = exp((x_j - max) * scale) / Z for j < len[b], 0 for padded keys.  One wave per row.
__global__ __launch_bounds__(256) void att_masked_softmax_kernel(const float* __restrict__ x,
                                                                 const int32_t* __restrict__ len, int H, int T1,
                                                                 int T2, float scale, float* __restrict__ y,
                                                                 size_t rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int b = (int)(row / ((size_t)H * T1));
  const int n = min(len[b], T2);
  const float* xr = x + row * T2;
  float* yr = y + row * T2;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, xr[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) sum += expf((xr[j] - mx) * scale);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int j = lane; j < T2; j += 64) yr[j] = (j < n) ? expf((xr[j] - mx) * scale) * inv : 0.f;
}

int launch_att_masked_softmax(const float* scores, const int32_t* len, int B, int H, int T1, int T2, float scale,
                              float* out, hipStream_t stream) {
  const size_t rows = (size_t)B * H * T1;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(att_masked_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, scores, len,
                     H, T1, T2, scale, out, rows);
  M3_LAUNCH_CHECK();
  return 0;
}

__global__ void softmax_lastdim_kernel(const float* __restrict__ x, float* __restrict__ y, size_t rows, int n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xr = x + row * n;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, xr[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) sum += expf(xr[j] - mx);
  sum = wave_sum(sum);
  for (int j = lane; j < n; j += 64) y[row * n + j] = expf(xr[j] - mx) / sum;
}

int launch_softmax_lastdim(const float* x, float* y, size_t rows, int n, hipStream_t stream) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax_lastdim_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, y, rows, n);
  M3_LAUNCH_CHECK();
  return 0;
}

// y = log_softmax(x) + bias  per row (bias may be null).  The acoustic-score back end the reference sketches:
// log-softmax commented out in builder.py:77-81, "- log prior" in builder.py:83-88, fused form in
// incomplete_plugin/prior_prob_plugin/prior_prob_kernel.cu:11-26 (log(p + 1e-20) + log_prior).
__global__ void log_softmax_bias_kernel(const float* x, const float* __restrict__ bias, float* y, size_t rows, int n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xr = x + row * n;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, xr[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) sum += expf(xr[j] - mx);
  const float lse = mx + logf(wave_sum(sum));
  for (int j = lane; j < n; j += 64) y[row * n + j] = xr[j] - lse + (bias ? bias[j] : 0.f);
}
int launch_log_softmax_bias(const float* x, const float* bias, float* y, size_t rows, int n, hipStream_t stream) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(log_softmax_bias_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, bias, y, rows, n);
  M3_LAUNCH_CHECK();
  return 0;
}

// global CMVN on (B,T,D) features, frames t < len[b] only (cmvn_plugin.cu:17-34); len may be null
__global__ void cmvn_kernel(const float* __restrict__ x, const int32_t* __restrict__ len, const float* __restrict__ mean,
                            const float* __restrict__ istd, int T, int D, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % T);
    const int b = (int)(i / ((size_t)D * T));
    y[i] = (len == nullptr || t < len[b]) ? (x[i] - mean[d]) * istd[d] : x[i];
  }
}
int launch_cmvn(const float* x, const int32_t* len, const float* mean, const float* istd, int B, int T, int D, float* y,
                hipStream_t stream) {
  const size_t n = (size_t)B * T * D;
  if (n == 0) return 0;
  hipLaunchKernelGGL(cmvn_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, x, len, mean, istd, T, D, y, n);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- masked_fill on (B,C,T)
__global__ void masked_fill_kernel(const float* __restrict__ x, const int32_t* __restrict__ len, int C, int T,
                                   float fill, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int b = (int)(i / ((size_t)C * T));
    y[i] = (t >= len[b]) ? fill : x[i];
  }
}

int launch_masked_fill(const float* x, const int32_t* len, int B, int C, int T, float fill, float* y,
                       hipStream_t stream) {
  const size_t n = (size_t)B * C * T;
  if (n == 0) return 0;
  hipLaunchKernelGGL(masked_fill_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, x,
                     len, C, T, fill, y, n);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- GLU along an arbitrary axis: x (outer, 2C, inner)
__global__ void glu_kernel(const float* __restrict__ x, int C, int inner, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t in = i % inner;
    const size_t c = (i / inner) % C;
    const size_t o = i / ((size_t)inner * C);
    const size_t a = (o * 2 * C + c) * inner + in;
    y[i] = x[a] * sigmoidf(x[a + (size_t)C * inner]);
  }
}

int launch_glu(const float* x, int outer, int C, int inner, float* y, hipStream_t stream) {
  const size_t n = (size_t)outer * C * inner;
  if (n == 0) return 0;
  hipLaunchKernelGGL(glu_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, x, C,
                     inner, y, n);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- flat element-wise
__global__ void scale_kernel(const float* __restrict__ x, float s, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = x[i] * s;
}
int launch_scale(const float* x, float scale, float* y, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(scale_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, x, scale,
                     y, n);
  M3_LAUNCH_CHECK();
  return 0;
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = a[i] + b[i];
}
int launch_add(const float* a, const float* b, float* y, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, a, b, y, n);
  M3_LAUNCH_CHECK();
  return 0;
}

__global__ void unary_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = act == ACT_RELU ? fmaxf(v, 0.f) : (act == ACT_SILU ? silu(v) : (act == 4 ? sigmoidf(v) : (act == 5 ? logf(v) : v)));
  }
}
int launch_unary(const float* x, float* y, size_t n, int act, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(unary_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, x, y, n,
                     act);
  M3_LAUNCH_CHECK();
  return 0;
}

struct Dims8 {
  int64_t v[8];
};

// y[idx] = a[bcast idx] (op) b[bcast idx]; op 0 = sum, 1 = prod
__global__ void binary_bcast_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                    Dims8 shape, Dims8 sa, Dims8 sb, int nd, int op, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    size_t rem = i, oa = 0, ob = 0;
    for (int d = nd - 1; d >= 0; --d) {
      const size_t c = rem % shape.v[d];
      rem /= shape.v[d];
      oa += c * sa.v[d];
      ob += c * sb.v[d];
    }
    y[i] = op == 0 ? a[oa] + b[ob] : a[oa] * b[ob];
  }
}
int launch_binary_bcast(const float* a, const float* b, float* y, const int64_t* shape, const int64_t* sa,
                        const int64_t* sb, int nd, int op, hipStream_t stream) {
  M3_REQUIRE(nd >= 1 && nd <= 8, "elementwise: rank %d unsupported", nd);
  Dims8 s{}, xa{}, xb{};
  size_t n = 1;
  for (int d = 0; d < nd; ++d) {
    s.v[d] = shape[d];
    xa.v[d] = sa[d];
    xb.v[d] = sb[d];
    n *= (size_t)shape[d];
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(binary_bcast_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, a,
                     b, y, s, xa, xb, nd, op, n);
  M3_LAUNCH_CHECK();
  return 0;
}

// y (contiguous, out_shape) [idx] = x[sum idx_d * in_strides_d]   (TensorRT shuffle = permute + reshape)
__global__ void permute_kernel(const float* __restrict__ x, float* __restrict__ y, Dims8 shape, Dims8 st, int nd,
                               size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    size_t rem = i, o = 0;
    for (int d = nd - 1; d >= 0; --d) {
      const size_t c = rem % shape.v[d];
      rem /= shape.v[d];
      o += c * st.v[d];
    }
    y[i] = x[o];
  }
}
int launch_permute(const float* x, float* y, const int64_t* out_shape, const int64_t* in_strides, int nd,
                   hipStream_t stream) {
  M3_REQUIRE(nd >= 1 && nd <= 8, "shuffle: rank %d unsupported", nd);
  Dims8 s{}, st{};
  size_t n = 1;
  for (int d = 0; d < nd; ++d) {
    s.v[d] = out_shape[d];
    st.v[d] = in_strides[d];
    n *= (size_t)out_shape[d];
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(permute_kernel, dim3(grid1d(n, 4096)), dim3(256), 0, stream, x, y, s,
                     st, nd, n);
  M3_LAUNCH_CHECK();
  return 0;
}

__global__ void concat_last_kernel(const float* __restrict__ a, int da, const float* __restrict__ b, int db,
                                   float* __restrict__ y, size_t n) {
  const int d = da + db;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / d;
    const int c = (int)(i % d);
    y[i] = c < da ? a[r * da + c] : b[r * db + (c - da)];
  }
}
int launch_concat_last(const float* a, int da, const float* b, int db, float* y, size_t rows, hipStream_t stream) {
  const size_t n = rows * (size_t)(da + db);
  if (n == 0) return 0;
  hipLaunchKernelGGL(concat_last_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, a,
                     da, b, db, y, n);
  M3_LAUNCH_CHECK();
  return 0;
}

__global__ void mask_conv2d_sample_kernel(const int32_t* __restrict__ in, int B, int left_padding, int stride,
                                          int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) out[i] = (in[i] - left_padding - 1) / stride + 1;
}
int launch_mask_conv2d_sample(const int32_t* len_in, int B, int left_padding, int stride, int32_t* len_out,
                              hipStream_t stream) {
  M3_REQUIRE(stride > 0, "mask_conv2d_sample: stride must be positive");
  if (B == 0) return 0;
  hipLaunchKernelGGL(mask_conv2d_sample_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, len_in, B, left_padding,
                     stride, len_out);
  M3_LAUNCH_CHECK();
  return 0;
}

// streaming state helpers (engine.hip, m3_engine_forward_chunk): the device-side chunk counter and the initial conv cache
__global__ void advance_counter_kernel(int32_t* __restrict__ counter, int by) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter += by;
}
int launch_advance_counter(int32_t* counter, int by, hipStream_t stream) {
  hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(64), 0, stream, counter, by);
  M3_LAUNCH_CHECK();
  return 0;
}
__global__ void fill_rows_kernel(const float* __restrict__ row, int D, float* __restrict__ out, size_t rows) {
  const size_t r = blockIdx.x;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) stg4(out + r * D + c, ldg4(row + c));
}
int launch_fill_rows(const float* row, int D, float* out, size_t rows, hipStream_t stream) {
  M3_REQUIRE(row && out && (D & 3) == 0, "fill_rows: null pointer or D %% 4 != 0");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)rows), dim3(128), 0, stream, row, D, out, rows);
  M3_LAUNCH_CHECK();
  return 0;
}

// both MaskConv2dSample applications of Conv2dSubsampling4 in one launch (subsampling.py:119-137)
__global__ void subsample_lens_kernel(const int32_t* __restrict__ in, int B, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    const int l1 = (in[i] - 3) / 2 + 1;
    out[i] = (l1 - 3) / 2 + 1;
  }
}
int launch_subsample_lens(const int32_t* len_in, int B, int32_t* len_out, hipStream_t stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(subsample_lens_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, len_in, B, len_out);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- packed (padding-free) row plan of a ragged batch
// len [B] valid frames per utterance (<= T) -> row0 [B+1] exclusive prefix (row0[B] = P, the packed row count) and
// pad_of [B*T]: packed row p -> its row b*T + t in the padded layout, -1 for p >= P.  One work-group.
// feat_len != null: `len` is an OUTPUT too -- the subsampled lengths are formed here from the raw feature lengths (what
// subsample_lens_kernel does), so that a packed forward starts with one launch instead of two.
__global__ __launch_bounds__(256) void pack_plan_kernel(int32_t* __restrict__ len, int B, int T, int32_t* __restrict__ row0,
                                                        int32_t* __restrict__ pad_of, const int32_t* __restrict__ feat_len) {
  __shared__ int start[1025];
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 0; b < B; ++b) {
      start[b] = run;
      int l = len[b];
      if (feat_len != nullptr) {
        l = ((feat_len[b] - 3) / 2 + 1 - 3) / 2 + 1;
        len[b] = l;
      }
      run += min(max(l, 0), T);
    }
    start[B] = run;
  }
  __syncthreads();
  for (int b = threadIdx.x; b <= B; b += blockDim.x) row0[b] = start[b];
  const int P = start[B];
  for (int i = threadIdx.x; i < B * T; i += blockDim.x) {
    const int b = i / T, t = i - b * T;
    if (t < start[b + 1] - start[b]) pad_of[start[b] + t] = i;
    if (i >= P) pad_of[i] = -1;          // disjoint from the writes above (those go to rows < P)
  }
}
int launch_pack_plan(int32_t* len, int B, int T, int32_t* row0, int32_t* pad_of, hipStream_t stream, const int32_t* feat_len) {
  M3_REQUIRE(B > 0 && B <= 1024 && T > 0, "pack_plan: batch %d out of range [1,1024]", B);
  hipLaunchKernelGGL(pack_plan_kernel, dim3(1), dim3(256), 0, stream, len, B, T, row0, pad_of, feat_len);
  M3_LAUNCH_CHECK();
  return 0;
}

// padded (B,T,n) <- packed rows: out[b][t] = in[row0[b] + t] for t < len[b], zeros beyond.  One wave per padded row.
__global__ __launch_bounds__(256) void unpack_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ row0,
                                                          int B, int T, int n, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= B * T) return;
  const int b = r / T, t = r - b * T;
  const int len = row0[b + 1] - row0[b];
  float* dst = out + (size_t)r * n;
  if (t < len) {
    const float* src = in + (size_t)(row0[b] + t) * n;
    for (int j = lane; j < n; j += 64) dst[j] = src[j];
  } else {
    for (int j = lane; j < n; j += 64) dst[j] = 0.f;
  }
}
int launch_unpack_rows(const float* in, const int32_t* row0, int B, int T, int n, float* out, hipStream_t stream) {
  if (B * T == 0) return 0;
  hipLaunchKernelGGL(unpack_rows_kernel, dim3(cdiv(B * T, 4)), dim3(256), 0, stream, in, row0, B, T, n, out);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- small strided-batched matmul (compat path only)
// c[b] (M,N) = a[b] (M,K) . b[b] (K,N)  or  a[b] . b[b]^T with b (N,K); sa/sb = batch strides (0 = broadcast).
__global__ __launch_bounds__(256) void bmm_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ c, int M, int N, int K, int64_t sa,
                                                  int64_t sb, int trans_b) {
  __shared__ float as[16][17], bs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int bz = blockIdx.z;
  const float* A = a + (size_t)bz * sa;
  const float* Bm = b + (size_t)bz * sb;
  const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    as[ty][tx] = (m < M && k0 + tx < K) ? A[(size_t)m * K + k0 + tx] : 0.f;
    const int nn = blockIdx.x * 16 + ty;  // bs[k][n]
    if (trans_b)
      bs[tx][ty] = (nn < N && k0 + tx < K) ? Bm[(size_t)nn * K + k0 + tx] : 0.f;
    else
      bs[ty][tx] = (k0 + ty < K && n < N) ? Bm[(size_t)(k0 + ty) * N + n] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += as[ty][k] * bs[k][tx];
    __syncthreads();
  }
  if (m < M && n < N) c[((size_t)bz * M + m) * N + n] = acc;
}
int launch_bmm(const float* a, const float* b, float* c, int batch, int M, int N, int K, int64_t sa, int64_t sb,
               int trans_b, hipStream_t stream) {
  if (batch == 0 || M == 0 || N == 0) return 0;
  dim3 grid(cdiv(N, 16), cdiv(M, 16), batch);
  hipLaunchKernelGGL(bmm_kernel, grid, dim3(256), 0, stream, a, b, c, M, N, K, sa, sb, trans_b);
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
