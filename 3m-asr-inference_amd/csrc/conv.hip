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

// z, out: [B*T][D]; w_kc: [K][D] (repacked from (D,1,K)).  One workgroup per frame, one float4 of
// channels per thread (D/4 threads): all K taps (z rows t-pad..t+pad and their weights) are loaded
// before the first FMA, so a frame costs one memory round trip; edge taps read a clamped row and are
// multiplied by 0 (= the conv's zero padding).  LayerNorm statistics go through a small LDS tree.
//
// Causal form (convolution.py:43-49,118-123: lorder = K - 1 frames are padded on the LEFT of the module's input, in front of
// pointwise_conv1, and the depthwise conv runs without padding): out[t] = sum_k w[k] z[t - (K-1) + k], no taps to the right.
// The K-1 frames left of frame 0 are not zeros at the depthwise conv's input: pointwise_conv1 + GLU turn a zero frame into
// the constant row GLU(bias).  `cs.left` supplies them: one broadcast row ("left_fill" of the plan; full-utterance forward),
// or, chunk by chunk, the last K-1 frames of the previous chunk's z from the streaming state (a ping-pong pair selected by
// the parity of the device-side chunk counter; the work-groups past the B*T frames write the other half: the new cache =
// the last K-1 frames of [cache | z[0 : len)], cat_split_cache_kernel.cu:30-55's contract on frames instead of elements).
struct DwCausal {
  int causal = 0;                          // 1: lorder = K - 1, no right taps
  const float* left = nullptr;             // causal: frames t < 0.  left_t_stride 0: one row [D] for every t < 0 and utterance
  long left_b_stride = 0; int left_t_stride = 0;
  const int32_t* step = nullptr;           // streaming: device-side chunk counter; `left` then is the [2][B][K-1][D] ping-pong pair
  long half = 0;                           // floats in one half of the pair
  const int32_t* chunk_len = nullptr;      // streaming: valid frames of this chunk per utterance
};

// CAUSAL is a template parameter: the symmetric form is the round-3 kernel instruction for instruction (making it a run-time
// field of one kernel cost 4.9 -> 8.3 us per launch at B = 1: a pointer select in front of every tap load).
// (the body as a device function of the work-group's row: dwconv_ln_silu_dual_kernel runs two independent problems in one launch,
//  see gemm.hip / engine.hip "horizontal fusion")
template <int KT, bool CAUSAL>
__device__ __forceinline__ void dwconv_ln_silu_body(const float* __restrict__ z, const float* __restrict__ w_kc,
                                                    const float* __restrict__ bias, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, int T, int D, int K,
                                                    float* __restrict__ out, int out_bf16, const int32_t* __restrict__ pad_of,
                                                    const int32_t* __restrict__ row0, const int32_t* __restrict__ row_len,
                                                    int n_rows, const DwCausal& cs, const int row) {
  __shared__ float red[2][16];
  // one batch of kernel-argument loads instead of one per first use (see gemm.hip: ~6 dependent s_load rounds otherwise)
  asm volatile("" ::"s"(z), "s"(w_kc), "s"(bias), "s"(gamma), "s"(beta), "s"(eps), "s"(T), "s"(D), "s"(K), "s"(out), "s"(out_bf16),
               "s"(pad_of), "s"(row0), "s"(row_len), "s"(n_rows));
  const int c = threadIdx.x * 4;
  const bool live = c < D;
  const float* leftp = cs.left;
  if (CAUSAL && cs.step != nullptr) {                         // streaming: read half (step & 1), write the other one
    const int par = *cs.step & 1;
    leftp = cs.left + (size_t)par * cs.half;
    if (row >= n_rows) {                            // cache update: row i of utterance b <- frame i + len of [cache | z]
      const int i = row - n_rows, b = i / (K - 1), ci = i - b * (K - 1);
      const int s = ci + min(max(cs.chunk_len[b], 0), T);
      const float* src = s < K - 1 ? leftp + (size_t)b * cs.left_b_stride + (size_t)s * cs.left_t_stride
                                   : z + ((size_t)b * T + (s - (K - 1))) * D;
      float* dst = const_cast<float*>(cs.left) + (size_t)(par ^ 1) * cs.half + (size_t)b * cs.left_b_stride + (size_t)ci * cs.left_t_stride;
      if (live) stg4(dst + c, ldg4(src + c));
      return;
    }
  }
  // padded rows: row = b T + t.  Packed rows (pad_of != null): row p is frame pad_of[p] = b T + t of the padded layout
  // (-1 past the last packed row P); utterance b owns rows [row0[b], row0[b] + len).  The reference runs the conv on the
  // padded batch, where the frames len <= tt < T hold the constant row pointwise_conv1 produces from a zeroed input
  // (masked_fill before the conv module, convolution.py:101-104): the packed layout keeps one copy of it at row P
  // (written by the pw1 GEMM with rows >= P masked), and tt < 0 or tt >= T are the conv's zero padding as before.
  int b = row / T, t = row % T, len = T;
  size_t r0 = (size_t)b * T, rpad = 0;
  if (pad_of != nullptr) {
    const int pr = pad_of[row];
    if (pr < 0) return;
    b = pr / T;
    t = pr - b * T;
    len = row_len[b];
    r0 = (size_t)row0[b];
    rpad = (size_t)row0[n_rows / T];                  // P = row0[B]
  }
  const int pad = CAUSAL ? K - 1 : (K - 1) / 2;
  if (CAUSAL) leftp += (size_t)b * cs.left_b_stride + (size_t)(K - 1) * cs.left_t_stride;   // frame tt < 0 is leftp + tt * left_t_stride
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    v = ldg4(bias + c);
    for (int k0 = 0; k0 < K; k0 += KT) {
      f32x4 zz[KT], ww[KT];
      float on[KT];
#pragma unroll
      for (int kk = 0; kk < KT; ++kk) {
        const int k = min(k0 + kk, K - 1);
        const int tt = t + k - pad;
        const int tc = min(max(tt, 0), T - 1);
        const float* src = z + (tc < len ? r0 + tc : rpad) * D;
        if (CAUSAL) {                                 // no taps to the right; frames left of the utterance come from `left`
          on[kk] = (k0 + kk < K) ? 1.f : 0.f;
          if (tt < 0) src = leftp + (long)tt * cs.left_t_stride;
        } else {
          on[kk] = (k0 + kk < K && tt >= 0 && tt < T) ? 1.f : 0.f;
        }
        zz[kk] = ldg4(src + c);
        ww[kk] = ldg4(w_kc + (size_t)k * D + c);
      }
#pragma unroll
      for (int kk = 0; kk < KT; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(zz[kk][j] * on[kk], ww[kk][j], v[j]);
    }
  }
  if (gamma != nullptr) {
    float s = live ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
    s = wave_sum(s);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < nwaves; ++w) tot += red[0][w];
    const float mean = tot / (float)D;
    float q = 0.f;
    if (live) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[j] - mean;
        q += d * d;
      }
    }
    q = wave_sum(q);
    if (lane == 0) red[1][wave] = q;
    __syncthreads();
    float qt = 0.f;
    for (int w = 0; w < nwaves; ++w) qt += red[1][w];
    const float rstd = rsqrtf(qt / (float)D + eps);
    if (live) {
      const f32x4 g = ldg4(gamma + c), be = ldg4(beta + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (v[j] - mean) * rstd * g[j] + be[j];
    }
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = silu(v[j]);
    if (out_bf16) {
      bf16x4 h;
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(out) + (size_t)row * D + c) = h;
    } else {
      stg4(out + (size_t)row * D + c, v);
    }
  }
}

template <int KT, bool CAUSAL>
__global__ __launch_bounds__(1024) void dwconv_ln_silu_kernel(const float* __restrict__ z, const float* __restrict__ w_kc,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, int T,
                                                              int D, int K, float* __restrict__ out, int out_bf16,
                                                              const int32_t* __restrict__ pad_of,
                                                              const int32_t* __restrict__ row0,
                                                              const int32_t* __restrict__ row_len, int n_rows, DwCausal cs) {
  dwconv_ln_silu_body<KT, CAUSAL>(z, w_kc, bias, gamma, beta, eps, T, D, K, out, out_bf16, pad_of, row0, row_len, n_rows, cs, (int)blockIdx.x);
}
struct DwKernArgs {
  const float *z, *w_kc, *bias, *gamma, *beta; float eps; int T, D, K; float* out; int out_bf16;
  const int32_t *pad_of, *row0, *row_len; int n_rows; DwCausal cs;
};
template <int KT, bool CAUSAL>
__global__ __launch_bounds__(1024) void dwconv_ln_silu_dual_kernel(const DwKernArgs a, const DwKernArgs b) {
  if ((int)blockIdx.x < a.n_rows)
    dwconv_ln_silu_body<KT, CAUSAL>(a.z, a.w_kc, a.bias, a.gamma, a.beta, a.eps, a.T, a.D, a.K, a.out, a.out_bf16, a.pad_of, a.row0, a.row_len,
                                    a.n_rows, a.cs, (int)blockIdx.x);
  else
    dwconv_ln_silu_body<KT, CAUSAL>(b.z, b.w_kc, b.bias, b.gamma, b.beta, b.eps, b.T, b.D, b.K, b.out, b.out_bf16, b.pad_of, b.row0, b.row_len,
                                    b.n_rows, b.cs, (int)blockIdx.x - a.n_rows);
}

static int launch_dwconv_impl(const float* z, const float* w_kc, const float* bias, const float* gamma, const float* beta, float eps,
                              int B, int T, int D, int K, float* out, hipStream_t stream, int out_bf16, const int32_t* pad_of,
                              const int32_t* row0, const int32_t* row_len, const DwCausal& cs) {
  M3_REQUIRE((D & 3) == 0 && D <= 4096, "dwconv: channels=%d must be a multiple of 4 (<=4096)", D);
  M3_REQUIRE(cs.causal || (K & 1) == 1, "dwconv: kernel size %d must be odd (non-causal)", K);
  M3_REQUIRE(!cs.causal || (cs.left != nullptr && K >= 2), "dwconv (causal): the frames left of frame 0 must be supplied (left_fill / cache)");
  const int rows = B * T;
  if (rows == 0) return 0;
  const int threads = (int)align_up(D / 4, 64);
  const int grid = rows + (cs.step != nullptr ? B * (K - 1) : 0);     // streaming: + the work-groups that write the new cache
#define M3_DW_CASE(KT_, C_)                                                                                                  \
  hipLaunchKernelGGL((dwconv_ln_silu_kernel<KT_, C_>), dim3(grid), dim3(threads), 0, stream, z, w_kc, bias, gamma, beta, eps, T, \
                     D, K, out, out_bf16, pad_of, row0, row_len, rows, cs)
  if (K <= 15) { if (cs.causal) M3_DW_CASE(15, true); else M3_DW_CASE(15, false); }
  else { if (cs.causal) M3_DW_CASE(8, true); else M3_DW_CASE(8, false); }
#undef M3_DW_CASE
  M3_LAUNCH_CHECK();
  return 0;
}

int launch_dwconv_ln_silu(const float* z, const float* w_kc, const float* bias, const float* gamma,
                          const float* beta, float eps, int B, int T, int D, int K, float* out, hipStream_t stream, int out_bf16,
                          const int32_t* pad_of, const int32_t* row0, const int32_t* row_len, const float* causal_left_fill) {
  DwCausal cs;
  if (causal_left_fill != nullptr) { cs.causal = 1; cs.left = causal_left_fill; }
  return launch_dwconv_impl(z, w_kc, bias, gamma, beta, eps, B, T, D, K, out, stream, out_bf16, pad_of, row0, row_len, cs);
}

int launch_dwconv_ln_silu_args(const DwArgs& a, hipStream_t stream) {
  return launch_dwconv_ln_silu(a.z, a.w_kc, a.bias, a.gamma, a.beta, a.eps, a.B, a.T, a.D, a.K, a.out, stream, a.out_bf16, a.pad_of, a.row0,
                               a.row_len, a.causal_left_fill);
}
// two independent conv modules of the same shape class (channels, taps, causal or not) in ONE launch
bool dwconv_dual_fusable(const DwArgs& a, const DwArgs& b) {
  return a.D == b.D && a.K == b.K && (a.causal_left_fill != nullptr) == (b.causal_left_fill != nullptr) && a.B * a.T > 0 && b.B * b.T > 0 &&
         (a.D & 3) == 0 && a.D <= 4096 && (a.causal_left_fill != nullptr || (a.K & 1) == 1);
}
int launch_dwconv_ln_silu_dual(const DwArgs& a, const DwArgs& b, hipStream_t stream) {
  M3_REQUIRE(dwconv_dual_fusable(a, b), "dwconv dual: the two problems do not share an instantiation");
  auto pack = [](const DwArgs& q) {
    DwKernArgs k;
    k.z = q.z; k.w_kc = q.w_kc; k.bias = q.bias; k.gamma = q.gamma; k.beta = q.beta; k.eps = q.eps; k.T = q.T; k.D = q.D; k.K = q.K;
    k.out = q.out; k.out_bf16 = q.out_bf16; k.pad_of = q.pad_of; k.row0 = q.row0; k.row_len = q.row_len; k.n_rows = q.B * q.T;
    if (q.causal_left_fill != nullptr) { k.cs.causal = 1; k.cs.left = q.causal_left_fill; }
    return k;
  };
  const DwKernArgs ka = pack(a), kb = pack(b);
  const int threads = (int)align_up(a.D / 4, 64), grid = ka.n_rows + kb.n_rows;
  const bool causal = a.causal_left_fill != nullptr;
#define M3_DWD_CASE(KT_, C_) hipLaunchKernelGGL((dwconv_ln_silu_dual_kernel<KT_, C_>), dim3(grid), dim3(threads), 0, stream, ka, kb)
  if (a.K <= 15) { if (causal) M3_DWD_CASE(15, true); else M3_DWD_CASE(15, false); }
  else { if (causal) M3_DWD_CASE(8, true); else M3_DWD_CASE(8, false); }
#undef M3_DWD_CASE
  M3_LAUNCH_CHECK();
  return 0;
}

// chunk-by-chunk (streaming) form of the causal conv: cache_pair [2][B][K-1][D], chunk counter and valid frames on the device
int launch_dwconv_ln_silu_stream(const float* z, const float* w_kc, const float* bias, const float* gamma, const float* beta,
                                 float eps, int B, int T, int D, int K, float* out, float* cache_pair, const int32_t* step,
                                 const int32_t* chunk_len, hipStream_t stream, int out_bf16) {
  M3_REQUIRE(cache_pair && step && chunk_len, "dwconv (stream): null state");
  DwCausal cs;
  cs.causal = 1; cs.left = cache_pair; cs.left_b_stride = (long)(K - 1) * D; cs.left_t_stride = D;
  cs.step = step; cs.half = (long)B * (K - 1) * D; cs.chunk_len = chunk_len;
  return launch_dwconv_impl(z, w_kc, bias, gamma, beta, eps, B, T, D, K, out, stream, out_bf16, nullptr, nullptr, nullptr, cs);
}

// feat [B][T][idim] -> out [B][T1][F1][C] (channel-last), w9c [9][C] (repacked from (C,1,3,3)), ReLU.
// Optional global CMVN folded into the read: x <- (x - mean[f]) * istd[f]   (the reference's unfinished CmvnPlugin,
// incomplete_plugin/cmvn_plugin/cmvn_plugin.cu:17-43; builder.sh:9 passes --cmvn_file but builder.py ignores it).
// One work-group per output row (b, t1): thread = 4 channels.  The thread's 9 x 4 weights and bias stay in registers,
// the three input rows (CMVN applied on the way) sit in LDS and are read as broadcasts, and the work-group walks the F1
// output columns writing C contiguous channels each -- no per-element index arithmetic, no weight reloads, 1-2 KB stores.
__global__ __launch_bounds__(256) void conv1_relu_kernel(const float* __restrict__ feat, const float* __restrict__ w9c,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ cmvn_mean,
                                                         const float* __restrict__ cmvn_istd, int T, int idim, int T1,
                                                         int F1, int C, float* __restrict__ out, int relu, int out_bf16,
                                                         const int32_t* __restrict__ feat_len, int32_t* __restrict__ lens_out, int B) {
  extern __shared__ float xs[];                       // [3][idim]
  const int row = blockIdx.x;                         // b * T1 + t1
  // the engine's first launch also forms the valid lengths after the two stride-2 convs (both MaskConv2dSample applications,
  // subsampling.py:119-137) -- one launch less at the head of every forward
  if (feat_len != nullptr && blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i < B; i += blockDim.x) lens_out[i] = ((feat_len[i] - 3) / 2 + 1 - 3) / 2 + 1;
  const int b = row / T1, t1 = row - b * T1;
  const float* src = feat + ((size_t)b * T + 2 * t1) * idim;
  for (int i = threadIdx.x; i < 3 * idim; i += blockDim.x) {
    float xv = src[i];
    if (cmvn_mean != nullptr) {
      const int f = i % idim;
      xv = (xv - cmvn_mean[f]) * cmvn_istd[f];
    }
    xs[i] = xv;
  }
  const int c = threadIdx.x * 4;
  const bool live = c < C;
  f32x4 w[9], bv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    bv = ldg4(bias + c);
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = ldg4(w9c + (size_t)k * C + c);
  }
  __syncthreads();
  if (!live) return;
  const size_t obase = (size_t)row * F1 * C + c;
  // short inputs: the F1 columns of a row are split over blockIdx.y so that a single utterance still fills the chip
  const int fchunk = (F1 + gridDim.y - 1) / gridDim.y;
  const int f_lo = blockIdx.y * fchunk, f_hi = min(F1, f_lo + fchunk);
  for (int f1 = f_lo; f1 < f_hi; ++f1) {
    f32x4 acc = bv;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float xv = xs[kh * idim + 2 * f1 + kw];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv, w[kh * 3 + kw][j], acc[j]);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = relu ? fmaxf(acc[j], 0.f) : acc[j];
    if (out_bf16) {
      bf16x4 h;
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = (bf16_t)acc[j];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(out) + obase + (size_t)f1 * C) = h;
    } else {
      stg4(out + obase + (size_t)f1 * C, acc);
    }
  }
}

int launch_conv1_relu(const float* feat, const float* w9c, const float* bias, const float* cmvn_mean,
                      const float* cmvn_istd, int B, int T, int idim, int C, float* out, hipStream_t stream, int relu, int out_bf16,
                      const int32_t* feat_len, int32_t* lens_out) {
  M3_REQUIRE(T >= 3 && idim >= 3, "subsampling: input (T=%d, idim=%d) shorter than the 3x3 kernel", T, idim);
  M3_REQUIRE((C & 3) == 0 && C <= 1024, "subsampling: channels=%d must be a multiple of 4 (<= 1024)", C);
  const int T1 = (T - 3) / 2 + 1, F1 = (idim - 3) / 2 + 1;
  if (B * T1 == 0) return 0;
  const int threads = (int)align_up(C / 4, 64);
  const int fsplit = B * T1 >= 2048 ? 1 : (B * T1 >= 512 ? 2 : 5);
  hipLaunchKernelGGL(conv1_relu_kernel, dim3(B * T1, fsplit), dim3(threads), 3 * idim * sizeof(float), stream,
                     feat, w9c, bias, cmvn_mean, cmvn_istd, T, idim, T1, F1, C, out, relu, out_bf16, feat_len, lens_out, B);
  M3_LAUNCH_CHECK();
  return 0;
}

// y (B,C,To) = depthwise conv of x (B,C,T), To = T + 2 pad - K + 1 (nn.Conv1d: pad = (K-1)/2 keeps the length, the
// causal module pads its input itself and convolves with pad = 0, convolution.py:43-49)
__global__ void depthwise_conv1d_nct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                            const float* __restrict__ bias, int C, int T, int To, int K, int pad,
                                            float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % To);
    const size_t bc = i / To;
    const int c = (int)(bc % C);
    const float* xr = x + bc * T;
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
  const int To = T + 2 * pad - K + 1;
  M3_REQUIRE(pad >= 0 && To > 0, "depthwise_conv1d: T=%d K=%d pad=%d leaves no output", T, K, pad);
  const size_t n = (size_t)B * C * To;
  if (n == 0) return 0;
  hipLaunchKernelGGL(depthwise_conv1d_nct_kernel, dim3(grid1d(n, 4096)), dim3(256), 0,
                     stream, x, w, bias, C, T, To, K, pad, y, n);
  M3_LAUNCH_CHECK();
  return 0;
}

// zero padding of the last two dims: x (outer, H, W) -> y (outer, H + pre_h + post_h, W + pre_w + post_w)
// (TensorRT IPaddingLayer, network.add_padding in the causal conv module, convolution.py:118-123)
__global__ void pad2d_kernel(const float* __restrict__ x, int H, int W, int pre_h, int pre_w, int Ho, int Wo,
                             float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int wo = (int)(i % Wo), ho = (int)((i / Wo) % Ho);
    const size_t o = i / ((size_t)Wo * Ho);
    const int h = ho - pre_h, w_ = wo - pre_w;
    y[i] = (h >= 0 && h < H && w_ >= 0 && w_ < W) ? x[(o * H + h) * W + w_] : 0.f;
  }
}
int launch_pad2d(const float* x, size_t outer, int H, int W, int pre_h, int post_h, int pre_w, int post_w, float* y, hipStream_t stream) {
  M3_REQUIRE(pre_h >= 0 && post_h >= 0 && pre_w >= 0 && post_w >= 0, "pad2d: negative padding (cropping) is not implemented");
  const int Ho = H + pre_h + post_h, Wo = W + pre_w + post_w;
  const size_t n = outer * Ho * Wo;
  if (n == 0) return 0;
  hipLaunchKernelGGL(pad2d_kernel, dim3(grid1d(n, 4096)), dim3(256), 0, stream, x, H, W, pre_h, pre_w, Ho, Wo, y, n);
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
