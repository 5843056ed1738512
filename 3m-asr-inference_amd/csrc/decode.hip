// what sits after the encoder (CTC search on the logits) and the reference's streaming operators.
//
//  * CTC greedy search (trainer_3m_fix/model/encoder.py:156-180): per frame argmax over the vocabulary, then per
//    utterance "drop repeats, drop blank" over the frames t < len[b].  The reference does the argmax on the device and the
//    collapse in a Python loop on the host; here both are kernels (one wave per frame, one wave per utterance with a
//    ballot / prefix-popcount compaction), so the result that leaves the GPU is the token list, not (B,T',V) scores.
//  * CTC prefix beam search (encoder.py:182-275): the per-frame log-softmax and first beam prune (`logp.topk(beam)`,
//    :224-231) run on the device -- only T'*k (value, index) pairs cross to the host instead of T'*V scores -- and the
//    sequential prefix recursion (:232-273) is a host routine over those pairs, in double precision like the Python floats
//    of the reference.
//  * The streaming operators of the plugin library (registration commented out in trt_plugin_plus.cpp:155-156; no model in
//    the tree uses them): CatSplitCache (cat_split_cache_kernel.cu:30-107), AttStreamSoftmax
//    (att_stream_softmax_kernel.cu:28-191) and RelPositionalEncoding with a frame offset
//    (rel_positional_encoding_kernel.cu:62-69,108-123).
#include <algorithm>
#include <cfloat>
#include <limits>
#include <cmath>
#include <map>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace m3 {

// ---------------------------------------------------------------- CTC greedy search
// (value, index) argmax with "first maximum wins" (torch.argmax): lexicographic (value desc, index asc)
__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float ov = __shfl_xor(v, off, 64);
    const int oi = __shfl_xor(i, off, 64);
    argmax_merge(v, i, ov, oi);
  }
}

// one wave per frame: ids[row] = argmax_j logits[row][j]
__global__ __launch_bounds__(256) void ctc_argmax_kernel(const float* __restrict__ logits, size_t rows, int V,
                                                         int32_t* __restrict__ ids) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xr = logits + row * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < V; j += 64) {
    const float v = xr[j];
    if (v > best || bi == 0x7fffffff) {   // ascending j: a later equal value never replaces an earlier one
      best = v;
      bi = j;
    }
  }
  wave_argmax(best, bi);
  if (lane == 0) ids[row] = bi;
}

// one wave per utterance: keep frame t iff t < len, ids[t] != blank and ids[t] != ids[t-1]; stable compaction
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ len,
                                                          int T, int blank, int32_t* __restrict__ tokens,
                                                          int32_t* __restrict__ n_tokens) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = len ? min(max(len[b], 0), T) : T;
  const int32_t* row = ids + (size_t)b * T;
  int32_t* out = tokens + (size_t)b * T;
  int count = 0;
  for (int t0 = 0; t0 < n; t0 += 64) {
    const int t = t0 + lane;
    bool keep = false;
    int id = blank;
    if (t < n) {
      id = row[t];
      keep = id != blank && (t == 0 || id != row[t - 1]);
    }
    const unsigned long long m = __ballot(keep);
    if (keep) out[count + __popcll(m & ((1ull << lane) - 1ull))] = id;
    count += __popcll(m);
  }
  for (int t = count + lane; t < T; t += 64) out[t] = -1;
  if (lane == 0) n_tokens[b] = count;
}

int launch_ctc_greedy(const float* logits, const int32_t* len, int B, int T, int V, int blank, int32_t* frame_ids,
                      int32_t* tokens, int32_t* n_tokens, hipStream_t stream) {
  M3_REQUIRE(B >= 0 && T >= 0 && V > 0, "ctc_greedy: bad shape B=%d T=%d V=%d", B, T, V);
  if (B == 0) return 0;
  const size_t rows = (size_t)B * T;
  if (rows)
    hipLaunchKernelGGL(ctc_argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, logits, rows, V, frame_ids);
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(64), 0, stream, frame_ids, len, T, blank, tokens, n_tokens);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- per-frame log-softmax + top-k (first beam prune)
// One wave per frame.  Selection order is (value desc, index asc); round r picks the best element strictly after the
// previous pick in that order, so nothing is mutated and the row is only re-read (from L1/L2) k times.
__global__ __launch_bounds__(256) void ctc_topk_kernel(const float* __restrict__ logits, size_t rows, int V, int k,
                                                       float* __restrict__ top_logp, int32_t* __restrict__ top_idx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xr = logits + row * V;
  float mx = -INFINITY;
  for (int j = lane; j < V; j += 64) mx = fmaxf(mx, xr[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < V; j += 64) sum += expf(xr[j] - mx);
  const float lse = mx + logf(wave_sum(sum));
  float pv = INFINITY;
  int pi = -1;
  for (int r = 0; r < k; ++r) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < V; j += 64) {
      const float v = xr[j];
      const bool after = v < pv || (v == pv && j > pi);
      if (after && (bi == 0x7fffffff || v > best)) {
        best = v;
        bi = j;
      }
    }
    wave_argmax(best, bi);
    if (lane == 0) {
      top_logp[row * k + r] = best - lse;
      top_idx[row * k + r] = bi;
    }
    pv = best;
    pi = bi;
  }
}

int launch_ctc_topk(const float* logits, size_t rows, int V, int k, float* top_logp, int32_t* top_idx, hipStream_t stream) {
  M3_REQUIRE(V > 0 && k > 0 && k <= V, "ctc_topk: need 0 < k <= V (k=%d, V=%d)", k, V);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(ctc_topk_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, logits, rows, V, k, top_logp,
                     top_idx);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- CTC prefix beam search, host side
namespace {
const double NEG_INF = -std::numeric_limits<double>::infinity();
// utils/common.py:148-156, for up to three terms
inline double log_add3(double a, double b, double c) {
  if (a == NEG_INF && b == NEG_INF && c == NEG_INF) return NEG_INF;
  const double m = std::max(a, std::max(b, c));
  return m + std::log(std::exp(a - m) + std::exp(b - m) + std::exp(c - m));
}
inline double log_add2(double a, double b) {
  if (a == NEG_INF && b == NEG_INF) return NEG_INF;
  const double m = std::max(a, b);
  return m + std::log(std::exp(a - m) + std::exp(b - m));
}
struct Hyp {
  std::vector<int32_t> prefix;
  double pb, pnb;   // log prob of the prefix ending in blank / in a non-blank
};
}  // namespace

int ctc_prefix_beam_search_host(const float* top_logp, const int32_t* top_idx, int T, int k, int beam, int blank,
                                int32_t* hyp_tokens, int32_t* hyp_len, float* hyp_score, int32_t* n_hyps) {
  M3_REQUIRE(T >= 0 && k > 0 && beam > 0, "ctc_prefix_beam_search: bad sizes T=%d k=%d beam=%d", T, k, beam);
  M3_REQUIRE(top_logp && top_idx && hyp_tokens && hyp_len && hyp_score && n_hyps, "ctc_prefix_beam_search: null pointer");
  std::vector<Hyp> cur(1);
  cur[0].pb = 0.0;
  cur[0].pnb = NEG_INF;
  std::vector<Hyp> next;
  std::map<std::vector<int32_t>, int> where;   // prefix -> slot in `next` (slots keep first-touch order, as a dict does)
  std::vector<int> order;
  std::vector<double> key;
  auto slot = [&](const std::vector<int32_t>& p) -> Hyp& {
    auto it = where.find(p);
    if (it == where.end()) {
      it = where.emplace(p, (int)next.size()).first;
      next.push_back(Hyp{p, NEG_INF, NEG_INF});
    }
    return next[it->second];
  };
  std::vector<int32_t> ext;
  for (int t = 0; t < T; ++t) {
    next.clear();
    where.clear();
    for (int j = 0; j < k; ++j) {
      const int32_t s = top_idx[(size_t)t * k + j];
      const double ps = (double)top_logp[(size_t)t * k + j];
      for (size_t h = 0; h < cur.size(); ++h) {
        // `cur` is not touched while `next` grows: copy what is needed before slot() may reallocate `next`
        const double pb = cur[h].pb, pnb = cur[h].pnb;
        const std::vector<int32_t>& prefix = cur[h].prefix;
        const bool has_last = !prefix.empty();
        if (s == blank) {
          Hyp& n = slot(prefix);
          n.pb = log_add3(n.pb, pb + ps, pnb + ps);
        } else if (has_last && s == prefix.back()) {
          {
            Hyp& n = slot(prefix);              // ...s s  -> ...s
            n.pnb = log_add2(n.pnb, pnb + ps);
          }
          ext = prefix;
          ext.push_back(s);
          Hyp& n = slot(ext);                   // ...s - s -> ...s s
          n.pnb = log_add2(n.pnb, pb + ps);
        } else {
          ext = prefix;
          ext.push_back(s);
          Hyp& n = slot(ext);
          n.pnb = log_add3(n.pnb, pb + ps, pnb + ps);
        }
      }
    }
    // second prune: best `beam` by total score, ties in first-touch order (Python's sorted(reverse=True) is stable)
    order.resize(next.size());
    key.resize(next.size());
    for (size_t i = 0; i < next.size(); ++i) {
      order[i] = (int)i;
      key[i] = log_add2(next[i].pb, next[i].pnb);
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] > key[b]; });
    const size_t keep = std::min(order.size(), (size_t)beam);
    std::vector<Hyp> pruned;
    pruned.reserve(keep);
    for (size_t i = 0; i < keep; ++i) pruned.push_back(std::move(next[order[i]]));
    cur.swap(pruned);
  }
  const int n = (int)std::min(cur.size(), (size_t)beam);
  for (int i = 0; i < n; ++i) {
    const int L = (int)cur[i].prefix.size();
    hyp_len[i] = L;
    for (int j = 0; j < L; ++j) hyp_tokens[(size_t)i * T + j] = cur[i].prefix[j];
    for (int j = L; j < T; ++j) hyp_tokens[(size_t)i * T + j] = -1;
    hyp_score[i] = (float)log_add2(cur[i].pb, cur[i].pnb);
  }
  *n_hyps = n;
  return 0;
}

// ---------------------------------------------------------------- CatSplitCache
// output[b] = in_cache[b] ++ input[b]  (cache_dim + input_dim values);  out_cache[b] = the last cache_dim values of
// output[b] (cat_split_cache_kernel.cu:30-107: both of the reference's branches, input_dim >= cache_dim and
// input_dim < cache_dim, compute exactly this).  One pass, 4-byte elements (f32 or i32 alike).
__global__ void cat_split_cache_kernel(const uint32_t* __restrict__ in_cache, const uint32_t* __restrict__ input, int cache_dim,
                                       int input_dim, uint32_t* __restrict__ output, uint32_t* __restrict__ out_cache,
                                       size_t n) {
  const int tot = cache_dim + input_dim;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / tot;
    const int o = (int)(i - b * tot);
    const uint32_t v = o < cache_dim ? in_cache[b * cache_dim + o] : input[b * input_dim + (o - cache_dim)];
    output[i] = v;
    if (o >= input_dim) out_cache[b * cache_dim + (o - input_dim)] = v;   // o >= tot - cache_dim
  }
}
int launch_cat_split_cache(const void* in_cache, const void* input, int B, int cache_dim, int input_dim, void* output,
                           void* out_cache, hipStream_t stream) {
  M3_REQUIRE(B >= 0 && cache_dim >= 0 && input_dim >= 0, "cat_split_cache: bad shape");
  M3_REQUIRE(out_cache != in_cache || cache_dim == 0, "cat_split_cache: out_cache must not alias in_cache");
  const size_t n = (size_t)B * (cache_dim + input_dim);
  if (n == 0) return 0;
  hipLaunchKernelGGL(cat_split_cache_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, (const uint32_t*)in_cache,
                     (const uint32_t*)input, cache_dim, input_dim, (uint32_t*)output, (uint32_t*)out_cache, n);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- AttStreamSoftmax
// scores [B][N][ld] (N = heads x query frames, ld = cache + chunk key positions).  Row (b, n) is valid on
// [first, last): first = max(0, ld - decode_frame_num[b]), last = min(ld, mask_idx[b]) + cache_len, and
//   out[j] = exp((x[j] - max_valid x) * scale) / sum_valid exp((x - max) * scale),   0 outside [first, last)
// (att_stream_softmax_kernel.cu:28-71,136-191; note the scale is applied AFTER the max is subtracted, as there).
// `last` is clamped to ld: the reference's small kernels do the same by returning for threadIdx.x >= ld, its large
// kernel would run into the next row.  Every position of the row is written (the large reference kernel leaves
// [last, ld) untouched); a row with no valid position gives zeros.
__global__ __launch_bounds__(256) void att_stream_softmax_kernel(const float* __restrict__ x, const int32_t* __restrict__ dfn,
                                                                 const int32_t* __restrict__ mask_idx, int N, int ld,
                                                                 int cache_len, float scale, float* __restrict__ y,
                                                                 size_t rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int b = (int)(row / N);
  const int first = max(0, ld - dfn[b]);
  const int last = min(ld, min(ld, mask_idx[b]) + cache_len);
  const float* xr = x + row * ld;
  float* yr = y + row * ld;
  float mx = -FLT_MAX;
  for (int j = first + lane; j < last; j += 64) mx = fmaxf(mx, xr[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = first + lane; j < last; j += 64) sum += expf((xr[j] - mx) * scale);
  sum = wave_sum(sum);
  const float rz = 1.f / sum;
  for (int j = lane; j < ld; j += 64) yr[j] = (j >= first && j < last) ? expf((xr[j] - mx) * scale) * rz : 0.f;
}
int launch_att_stream_softmax(const float* scores, const int32_t* decode_frame_num, const int32_t* mask_idx, int B, int N,
                              int ld, int cache_len, float scale, float* out, hipStream_t stream) {
  M3_REQUIRE(B >= 0 && N >= 0 && ld > 0 && cache_len >= 0, "att_stream_softmax: bad shape");
  const size_t rows = (size_t)B * N;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(att_stream_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, scores,
                     decode_frame_num, mask_idx, N, ld, cache_len, scale, out, rows);
  M3_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- RelPositionalEncoding (+ streaming offset)
// y = x * scale on (B,T,D); pos_emb = pe[off : off+T] with off = frame_num[0] (0 when frame_num is null: the full
// utterance form, rel_positional_encoding_kernel.cu:62-69); frame_num_out[b] = frame_num[b] + T.  The streaming kernel of
// the reference (:108-123) states this contract in its comment but still copies pe[0:T] and never advances the counter;
// the stated contract is what is implemented.
__global__ void rel_pos_enc_kernel(const float* __restrict__ x, const float* __restrict__ pe, const int32_t* __restrict__ frame_num,
                                   float scale, int B, int T, int D, float* __restrict__ y, float* __restrict__ pos_emb,
                                   int32_t* __restrict__ frame_num_out) {
  const size_t pos_n = (size_t)T * D, n = pos_n * B;
  const size_t off = frame_num ? (size_t)frame_num[0] * D : 0;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = tid; i < n; i += (size_t)gridDim.x * blockDim.x) {
    y[i] = x[i] * scale;
    if (i < pos_n) pos_emb[i] = pe[off + i];
  }
  if (frame_num && frame_num_out && tid < (size_t)B) frame_num_out[tid] = frame_num[tid] + T;
}
int launch_rel_positional_encoding(const float* x, const float* pe, int pe_len, const int32_t* frame_num, int max_offset,
                                   float scale, int B, int T, int D, float* y, float* pos_emb, int32_t* frame_num_out,
                                   hipStream_t stream) {
  M3_REQUIRE(B > 0 && T > 0 && D > 0, "rel_positional_encoding: empty problem");
  M3_REQUIRE((frame_num ? max_offset : 0) + T <= pe_len,
             "rel_positional_encoding: offset %d + %d frames exceeds the %d-position table", frame_num ? max_offset : 0, T, pe_len);
  // every work-group reads frame_num[0] as the offset, so the advanced counters go to a distinct buffer
  M3_REQUIRE(frame_num == nullptr || frame_num_out != frame_num, "rel_positional_encoding: frame_num_out must not alias frame_num");
  const size_t n = (size_t)B * T * D;
  hipLaunchKernelGGL(rel_pos_enc_kernel, dim3(grid1d(n, 2048)), dim3(256), 0, stream, x, pe, frame_num, scale, B, T, D, y,
                     pos_emb, frame_num_out);
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
