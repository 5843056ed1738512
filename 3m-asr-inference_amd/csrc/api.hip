// C-ABI entry points of libm3asr_hip.so that are thin wrappers over the kernel launchers
// (include/m3asr.h documents which reference interface each one replaces).
#include <string.h>

#include "../../include/m3asr.h"
#include "common.h"
#include "kernels.h"

namespace m3 {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

// workspace layout of the fused FMoE op (all sections 256-B aligned, as kAlignment in
// the reference's common/common.h:43)
struct MoeWorkspace {
  int32_t* mapping;
  int32_t* acc;
  int32_t* pos;
  float* slab;
  size_t bytes;
};
MoeWorkspace carve_moe_workspace(void* base, int S, int E, int D, int F) {
  MoeWorkspace w;
  size_t off = 0;
  char* p = (char*)base;
  w.mapping = (int32_t*)(p + off); off += align_up((size_t)S * 4, 256);
  w.acc = (int32_t*)(p + off);     off += align_up((size_t)(E + 1) * 4, 256);
  w.pos = (int32_t*)(p + off);     off += align_up((size_t)S * 4, 256);
  w.slab = (float*)(p + off);      off += align_up(expert_ffn_slab_bytes(S, D, F), 256);
  w.bytes = off;
  return w;
}

int moe_expert_ffn_dt(const float* x, const int32_t* gate_idx, const float* w1, const float* b1, const float* w2,
                      const float* b2, int S, int E, int D, int F, const float* gate_value, const float* resid,
                      float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps, float* y, void* ws,
                      size_t ws_bytes, hipStream_t stream, int wmode /* 0 fp32, 1 bf16, 2 fp8 weights, 3 fp8 weights + activations */,
                      const float* s1 = nullptr, const float* s2 = nullptr, float h_scale = 0.f, const void* xq = nullptr,
                      const float* xq_scale = nullptr) {
  M3_REQUIRE(S >= 0 && E > 0 && D > 0 && F > 0, "fmoe_expert: bad sizes S=%d E=%d D=%d F=%d", S, E, D, F);
  if (S == 0) return 0;
  MoeWorkspace w = carve_moe_workspace(ws, S, E, D, F);
  M3_REQUIRE(ws != nullptr && ws_bytes >= w.bytes, "fmoe_expert: workspace %zu bytes < required %zu", ws_bytes,
             w.bytes);
  int rc = launch_moe_index(gate_idx, S, E, w.mapping, w.acc, w.pos, stream);
  if (rc) return rc;
  if (wmode == 3) rc = launch_expert_ffn_w8a8(x, D, w.pos, w.acc, S, E, D, F, w1, s1, b1, w2, s2, 0, h_scale, w.slab, stream, xq, xq_scale);
  else if (wmode == 2) rc = launch_expert_ffn_w8(x, D, w.pos, w.acc, S, E, D, F, w1, s1, b1, w2, s2, 0, w.slab, stream);
  else if (wmode) rc = launch_expert_ffn_bf16w(x, D, w.pos, w.acc, S, E, D, F, w1, b1, w2, 0, w.slab, stream);
  else rc = launch_expert_ffn_f32(x, D, w.pos, w.acc, S, E, D, F, w1, b1, w2, 0, w.slab, nullptr, nullptr, 0.f, stream);
  if (rc) return rc;
  const float* rows = wmode ? expert_ffn_w16_rows(wmode, w.slab, S, E, D, F) : expert_ffn_f32_rows(w.slab, S, E, D, F);
  const int n_slices = wmode ? expert_ffn_w16_slices(wmode, S, E, D, F) : expert_ffn_f32_slices(S, E, D, F);
  return launch_moe_combine(rows, n_slices, w.mapping, gate_idx, gate_value, b2, resid, alpha, ln_gamma, ln_beta, ln_eps,
                            y, S, D, stream);
}

int moe_expert_ffn(const float* x, const int32_t* gate_idx, const float* w1, const float* b1, const float* w2,
                   const float* b2, int S, int E, int D, int F, const float* gate_value, const float* resid,
                   float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps, float* y, void* ws,
                   size_t ws_bytes, hipStream_t stream) {
  return moe_expert_ffn_dt(x, gate_idx, w1, b1, w2, b2, S, E, D, F, gate_value, resid, alpha, ln_gamma, ln_beta, ln_eps,
                           y, ws, ws_bytes, stream, 0);
}

}  // namespace m3

using namespace m3;

extern "C" {

int m3_abi_version(void) { return M3ASR_ABI_VERSION; }
const char* m3_last_error(void) { return m3::last_error(); }

int m3_moe_scatter_mapping(const int32_t* gate_idx, int S, int num_expert, int32_t* mapping, int32_t* acc_histogram,
                           int32_t* pos, m3_stream stream) {
  M3_REQUIRE(gate_idx && mapping && acc_histogram, "moe_scatter_mapping: null pointer");
  return launch_moe_index(gate_idx, S, num_expert, mapping, acc_histogram, pos, (hipStream_t)stream);
}
int m3_moe_local_scatter(const void* x, const int32_t* mapping, int S, int row_bytes, void* out, m3_stream stream) {
  return launch_local_scatter(x, mapping, S, row_bytes, out, (hipStream_t)stream);
}
int m3_moe_local_gather(const void* buf, const int32_t* mapping, int S, int row_bytes, void* out, m3_stream stream) {
  return launch_local_gather(buf, mapping, S, row_bytes, out, (hipStream_t)stream);
}
int m3_moe_expert_slice(void) { return kExpertSlice; }
size_t m3_moe_expert_workspace_size(int S, int num_expert, int idim, int hidden_units) {
  if (S <= 0 || hidden_units % kExpertSlice) return 0;
  return carve_moe_workspace(nullptr, S, num_expert, idim, hidden_units).bytes;
}
int m3_moe_expert_ffn(const float* x, const int32_t* gate_idx, const float* w1, const float* b1, const float* w2,
                      const float* b2, int S, int num_expert, int idim, int hidden_units, const float* gate_value,
                      const float* resid, float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps,
                      float* y, void* workspace, size_t workspace_bytes, m3_stream stream) {
  return moe_expert_ffn(x, gate_idx, w1, b1, w2, b2, S, num_expert, idim, hidden_units, gate_value, resid, alpha,
                        ln_gamma, ln_beta, ln_eps, y, workspace, workspace_bytes, (hipStream_t)stream);
}
int m3_moe_expert_ffn_bf16(const float* x, const int32_t* gate_idx, const void* w1, const float* b1, const void* w2,
                           const float* b2, int S, int num_expert, int idim, int hidden_units, const float* gate_value,
                           const float* resid, float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps,
                           float* y, void* workspace, size_t workspace_bytes, m3_stream stream) {
  return moe_expert_ffn_dt(x, gate_idx, (const float*)w1, b1, (const float*)w2, b2, S, num_expert, idim, hidden_units,
                           gate_value, resid, alpha, ln_gamma, ln_beta, ln_eps, y, workspace, workspace_bytes,
                           (hipStream_t)stream, 1);
}
int m3_moe_expert_ffn_fp8(const float* x, const int32_t* gate_idx, const void* w1, const float* w1_scale, const float* b1,
                          const void* w2, const float* w2_scale, const float* b2, int S, int num_expert, int idim,
                          int hidden_units, const float* gate_value, const float* resid, float alpha,
                          const float* ln_gamma, const float* ln_beta, float ln_eps, float* y, void* workspace,
                          size_t workspace_bytes, m3_stream stream) {
  M3_REQUIRE(w1_scale && w2_scale, "fmoe_expert fp8: null scale");
  return moe_expert_ffn_dt(x, gate_idx, (const float*)w1, b1, (const float*)w2, b2, S, num_expert, idim, hidden_units,
                           gate_value, resid, alpha, ln_gamma, ln_beta, ln_eps, y, workspace, workspace_bytes,
                           (hipStream_t)stream, 2, w1_scale, w2_scale);
}
int m3_moe_expert_ffn_fp8a8(const float* x, const int32_t* gate_idx, const void* w1, const float* w1_scale, const float* b1,
                            const void* w2, const float* w2_scale, const float* b2, float h_scale, int S, int num_expert, int idim,
                            int hidden_units, const float* gate_value, const float* resid, float alpha,
                            const float* ln_gamma, const float* ln_beta, float ln_eps, float* y, void* workspace,
                            size_t workspace_bytes, m3_stream stream) {
  M3_REQUIRE(w1_scale && w2_scale, "fmoe_expert fp8a8: null scale");
  M3_REQUIRE(h_scale > 0.f, "fmoe_expert fp8a8: h_scale must be positive");
  return moe_expert_ffn_dt(x, gate_idx, (const float*)w1, b1, (const float*)w2, b2, S, num_expert, idim, hidden_units,
                           gate_value, resid, alpha, ln_gamma, ln_beta, ln_eps, y, workspace, workspace_bytes,
                           (hipStream_t)stream, 3, w1_scale, w2_scale, h_scale);
}
int m3_moe_expert_ffn_fp8a8_xq(const float* x, const void* xq, const float* xq_scale, const int32_t* gate_idx, const void* w1,
                               const float* w1_scale, const float* b1, const void* w2, const float* w2_scale, const float* b2,
                               float h_scale, int S, int num_expert, int idim, int hidden_units, const float* gate_value,
                               const float* resid, float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps,
                               float* y, void* workspace, size_t workspace_bytes, m3_stream stream) {
  M3_REQUIRE(w1_scale && w2_scale, "fmoe_expert fp8a8_xq: null scale");
  M3_REQUIRE(h_scale > 0.f, "fmoe_expert fp8a8_xq: h_scale must be positive");
  M3_REQUIRE(xq && xq_scale, "fmoe_expert fp8a8_xq: null quantised rows / scales");
  M3_REQUIRE(expert_ffn_fused_fp8_applies(S, num_expert, idim, hidden_units) || x != nullptr,
             "fmoe_expert fp8a8_xq: this shape takes the weight-only form, which needs the fp32 rows (x)");
  return moe_expert_ffn_dt(x, gate_idx, (const float*)w1, b1, (const float*)w2, b2, S, num_expert, idim, hidden_units,
                           gate_value, resid, alpha, ln_gamma, ln_beta, ln_eps, y, workspace, workspace_bytes,
                           (hipStream_t)stream, 3, w1_scale, w2_scale, h_scale, xq, xq_scale);
}
int m3_quantize_rows_e4m3(const float* x, int ldx, int S, int idim, void* xq, float* scale, m3_stream stream) {
  return launch_quantize_rows_e4m3(x, ldx, S, idim, xq, scale, (hipStream_t)stream);
}
int m3_moe_expert_ffn_fp8a8_active(int S, int num_expert, int idim, int hidden_units) {
  return expert_ffn_fused_fp8_applies(S, num_expert, idim, hidden_units) ? 1 : 0;
}
int m3_moe_combine(const float* rows, const int32_t* mapping, const float* gate_value, const float* resid, float alpha,
                   const float* ln_gamma, const float* ln_beta, float ln_eps, float* out, int S, int idim,
                   m3_stream stream) {
  M3_REQUIRE(rows && mapping && out, "moe_combine: null pointer");
  return launch_moe_combine(rows, 1, mapping, nullptr, gate_value, nullptr, resid, alpha, ln_gamma, ln_beta, ln_eps, out,
                            S, idim, (hipStream_t)stream);
}
int m3_moe_combine_bf16(const float* rows, const int32_t* mapping, const float* gate_value, const float* resid, float alpha,
                        const float* ln_gamma, const float* ln_beta, float ln_eps, float* out, void* out_bf16, int S, int idim,
                        m3_stream stream) {
  M3_REQUIRE(rows && mapping && out, "moe_combine_bf16: null pointer");
  return launch_moe_combine(rows, 1, mapping, nullptr, gate_value, nullptr, resid, alpha, ln_gamma, ln_beta, ln_eps, out,
                            S, idim, (hipStream_t)stream, out_bf16);
}
int m3_ep_send_map(const int32_t* gate_idx, const int32_t* mapping, const int32_t* acc_histogram, int S, int world, int e_loc,
                   int capacity, int32_t* map_send, void* wire, int row_bytes, m3_stream stream) {
  M3_REQUIRE(gate_idx && mapping && acc_histogram && map_send && wire, "ep_send_map: null pointer");
  return launch_ep_send_map(gate_idx, mapping, acc_histogram, S, world, e_loc, capacity, map_send, wire, row_bytes,
                            (hipStream_t)stream);
}
int m3_ep_recv_gate(const void* wire, int world, int e_loc, int capacity, int row_bytes, int32_t* gate_recv, m3_stream stream) {
  M3_REQUIRE(wire && gate_recv, "ep_recv_gate: null pointer");
  return launch_ep_recv_gate(wire, world, e_loc, capacity, row_bytes, gate_recv, (hipStream_t)stream);
}
int m3_moe_router(const float* embed, int ld_embed, int embed_dim, const float* x, int ldx, int idim, const float* w,
                  const float* bias, const float* ln_gamma, const float* ln_beta, float ln_eps, float* xn, int ld_xn,
                  float* logits, int ld_logits, int S, int num_expert, m3_stream stream) {
  return launch_moe_router(embed, ld_embed, embed_dim, x, ldx, idim, w, bias, ln_gamma, ln_beta, ln_eps, xn, ld_xn, logits,
                           ld_logits, S, num_expert, nullptr, (hipStream_t)stream);
}
int m3_softmax_top1(const float* logits, int ld, const int32_t* len, int rows_per_batch, int S, int width,
                    int32_t* idx, float* value, m3_stream stream) {
  return launch_softmax_top1(logits, ld, len, rows_per_batch, S, width, idx, value, (hipStream_t)stream);
}

static int linear_params(const m3_linear_desc* d, GemmParams* out) {
  M3_REQUIRE(d != nullptr, "linear: null descriptor");
  GemmParams p;
  p.A = d->a; p.lda = d->lda;
  p.A2 = d->a2; p.lda2 = d->lda2; p.K1 = d->k1;
  p.mode = d->a2 ? GEMM_A_CONCAT2 : GEMM_A_PLAIN;
  p.W = (const float*)d->w; p.bias = d->bias; p.Y = d->y; p.ldy = d->ldy;
  M3_REQUIRE(d->weight_dtype == M3_F32 || d->weight_dtype == M3_BF16, "linear: weight_dtype %d", d->weight_dtype);
  p.w_bf16 = d->weight_dtype == M3_BF16;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta; p.ln_eps = d->ln_eps;
  p.ln_wsum = d->ln_wsum; p.ln_wbeta = d->ln_wbeta;
  p.row_len = d->len; p.rows_per_batch = d->rows_per_batch; p.mask_in = d->mask_in; p.mask_out = d->mask_out;
  p.act = d->act; p.alpha = d->alpha; p.resid = d->resid; p.ldr = d->ldr;
  M3_REQUIRE((d->a_dtype == M3_F32 || d->a_dtype == M3_BF16) && (d->y_dtype == M3_F32 || d->y_dtype == M3_BF16),
             "linear: a_dtype / y_dtype must be f32 or bf16");
  if (d->a_dtype == M3_BF16 || d->y_dtype == M3_BF16 || d->y_copy_bf16)
    M3_REQUIRE(p.w_bf16 && p.mode == GEMM_A_PLAIN && !p.ln_gamma, "linear: bf16 activations need bf16 weights, plain A, no affine LayerNorm");
  p.a_bf16 = d->a_dtype == M3_BF16; p.y_bf16 = d->y_dtype == M3_BF16;
  p.Yb = d->y_copy_bf16; p.ldyb = d->ld_copy; p.Yb_stats = d->y_copy_stats;
  p.ln_stats = d->ln_stats; p.ln_stat_parts = d->ln_stat_parts;
  *out = p;
  return 0;
}
int m3_linear(const m3_linear_desc* d, m3_stream stream) {
  GemmParams p;
  if (int rc = linear_params(d, &p)) return rc;
  return launch_gemm_f32(p, (hipStream_t)stream);
}
size_t m3_linear_workspace_size(const m3_linear_desc* d) {
  GemmParams p;
  size_t need = 0;
  if (linear_params(d, &p) == 0) gemm_f32_splitk_plan(p, &need);
  return need;
}
int m3_linear_ws(const m3_linear_desc* d, void* workspace, size_t workspace_bytes, m3_stream stream) {
  GemmParams p;
  if (int rc = linear_params(d, &p)) return rc;
  size_t need = 0;
  if (gemm_f32_splitk_plan(p, &need) >= 2 && workspace != nullptr && workspace_bytes >= need)
    return launch_gemm_f32_splitk(p, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  return launch_gemm_f32(p, (hipStream_t)stream);
}

int m3_layer_norm(const float* x, const float* gamma, const float* beta, float eps, float* y, int rows, int dim,
                  m3_stream stream) {
  return launch_layernorm(x, gamma, beta, eps, y, rows, dim, (hipStream_t)stream);
}
int m3_relpos_attention(const float* qkv, int ldq, const float* p, int ldp, const float* pos_u, const float* pos_v,
                        const int32_t* len, int B, int T, int H, int dk, float scale, float* out, int ldo,
                        m3_stream stream) {
  return launch_relpos_attention(qkv, ldq, p, ldp, pos_u, pos_v, len, B, T, H, dk, scale, out, ldo,
                                 (hipStream_t)stream);
}
int m3_relpos_attention_bf16(const void* qkv, int ldq, const float* p, int ldp, const float* pos_u, const float* pos_v,
                             const int32_t* len, int B, int T, int H, int dk, float scale, int chunk, int left_chunks,
                             void* out, int ldo, m3_stream stream) {
  return launch_relpos_attention_bf16(qkv, ldq, p, ldp, pos_u, pos_v, len, B, T, H, dk, scale, out, ldo, (hipStream_t)stream,
                                      nullptr, chunk, left_chunks);
}
int m3_relpos_attention_chunk(const float* qkv, int ldq, const float* p, int ldp, const float* pos_u, const float* pos_v,
                              const int32_t* len, int B, int T, int H, int dk, float scale, int chunk, int left_chunks,
                              float* out, int ldo, m3_stream stream) {
  return launch_relpos_attention(qkv, ldq, p, ldp, pos_u, pos_v, len, B, T, H, dk, scale, out, ldo, (hipStream_t)stream, 0,
                                 nullptr, chunk, left_chunks);
}
int m3_dwconv_ln_silu(const float* z, const float* w_kc, const float* bias, const float* gamma, const float* beta,
                      float eps, int B, int T, int D, int K, float* out, m3_stream stream) {
  return launch_dwconv_ln_silu(z, w_kc, bias, gamma, beta, eps, B, T, D, K, out, (hipStream_t)stream);
}
int m3_subsample_conv1(const float* feat, const float* w9c, const float* bias, int B, int T, int idim, int C,
                       float* out, m3_stream stream) {
  return launch_conv1_relu(feat, w9c, bias, nullptr, nullptr, B, T, idim, C, out, (hipStream_t)stream);
}
int m3_subsample_conv1_cmvn(const float* feat, const float* w9c, const float* bias, const float* cmvn_mean,
                            const float* cmvn_istd, int B, int T, int idim, int C, float* out, m3_stream stream) {
  M3_REQUIRE((cmvn_mean == nullptr) == (cmvn_istd == nullptr), "subsample_conv1: cmvn mean and istd go together");
  return launch_conv1_relu(feat, w9c, bias, cmvn_mean, cmvn_istd, B, T, idim, C, out, (hipStream_t)stream);
}
int m3_conv2d_3x3s2_first(const float* feat, const float* w9c, const float* bias, int B, int T, int idim, int C, int act,
                          float* out, m3_stream stream) {
  M3_REQUIRE(act == M3_ACT_NONE || act == M3_ACT_RELU, "conv2d: act %d (none / relu)", act);
  return launch_conv1_relu(feat, w9c, bias, nullptr, nullptr, B, T, idim, C, out, (hipStream_t)stream, act == M3_ACT_RELU);
}
int m3_cmvn(const float* x, const int32_t* len, const float* mean, const float* istd, int B, int T, int D, float* y,
            m3_stream stream) {
  M3_REQUIRE(x && mean && istd && y, "cmvn: null pointer");
  return launch_cmvn(x, len, mean, istd, B, T, D, y, (hipStream_t)stream);
}
int m3_log_softmax_bias(const float* x, const float* bias, float* y, size_t rows, int n, m3_stream stream) {
  M3_REQUIRE(x && y && n > 0, "log_softmax_bias: bad arguments");
  return launch_log_softmax_bias(x, bias, y, rows, n, (hipStream_t)stream);
}
int m3_conv2d_3x3s2(const float* in, const float* w, const float* bias, int B, int T1, int F1, int C, int act, float* out,
                    m3_stream stream);
int m3_subsample_conv2(const float* in, const float* w, const float* bias, int B, int T1, int F1, int C, float* out,
                       m3_stream stream) {
  return m3_conv2d_3x3s2(in, w, bias, B, T1, F1, C, M3_ACT_RELU, out, stream);
}
int m3_conv2d_3x3s2(const float* in, const float* w, const float* bias, int B, int T1, int F1, int C, int act, float* out,
                    m3_stream stream) {
  M3_REQUIRE(T1 >= 3 && F1 >= 3, "subsample_conv2: input (%d,%d) smaller than the kernel", T1, F1);
  M3_REQUIRE(act == M3_ACT_NONE || act == M3_ACT_RELU || act == M3_ACT_SILU, "conv2d: act %d", act);
  GemmParams p;
  p.mode = GEMM_A_CONV3X3S2;
  p.A = in; p.lda = 4;
  p.conv_T1 = T1; p.conv_F1 = F1; p.conv_T2 = (T1 - 3) / 2 + 1; p.conv_F2 = (F1 - 3) / 2 + 1; p.conv_C = C;
  p.W = w; p.bias = bias; p.Y = out; p.ldy = C;
  p.M = B * p.conv_T2 * p.conv_F2; p.N = C; p.K = 9 * C;
  p.act = act;
  return launch_gemm_f32(p, (hipStream_t)stream);
}

int m3_att_masked_softmax(const float* scores, const int32_t* len, int B, int H, int T1, int T2, float scale,
                          float* out, m3_stream stream) {
  return launch_att_masked_softmax(scores, len, B, H, T1, T2, scale, out, (hipStream_t)stream);
}
int m3_ctc_greedy(const float* logits, const int32_t* len, int B, int T, int V, int blank, int32_t* frame_ids,
                  int32_t* tokens, int32_t* n_tokens, m3_stream stream) {
  return launch_ctc_greedy(logits, len, B, T, V, blank, frame_ids, tokens, n_tokens, (hipStream_t)stream);
}
int m3_ctc_topk(const float* logits, size_t rows, int V, int k, float* top_logp, int32_t* top_idx, m3_stream stream) {
  return launch_ctc_topk(logits, rows, V, k, top_logp, top_idx, (hipStream_t)stream);
}
int m3_ctc_prefix_beam_search(const float* top_logp, const int32_t* top_idx, int T, int k, int beam, int blank,
                              int32_t* hyp_tokens, int32_t* hyp_len, float* hyp_score, int32_t* n_hyps) {
  return ctc_prefix_beam_search_host(top_logp, top_idx, T, k, beam, blank, hyp_tokens, hyp_len, hyp_score, n_hyps);
}
int m3_cat_split_cache(const void* in_cache, const void* input, int B, int cache_dim, int input_dim, void* output,
                       void* out_cache, m3_stream stream) {
  return launch_cat_split_cache(in_cache, input, B, cache_dim, input_dim, output, out_cache, (hipStream_t)stream);
}
int m3_att_stream_softmax(const float* scores, const int32_t* decode_frame_num, const int32_t* mask_idx, int B, int N,
                          int ld, int cache_len, float scale, float* out, m3_stream stream) {
  return launch_att_stream_softmax(scores, decode_frame_num, mask_idx, B, N, ld, cache_len, scale, out, (hipStream_t)stream);
}
int m3_rel_positional_encoding(const float* x, const float* pe, int pe_len, const int32_t* frame_num, int max_offset,
                               float scale, int B, int T, int D, float* y, float* pos_emb, int32_t* frame_num_out,
                               m3_stream stream) {
  return launch_rel_positional_encoding(x, pe, pe_len, frame_num, max_offset, scale, B, T, D, y, pos_emb, frame_num_out,
                                        (hipStream_t)stream);
}
int m3_masked_fill(const float* x, const int32_t* len, int B, int C, int T, float fill, float* y, m3_stream stream) {
  return launch_masked_fill(x, len, B, C, T, fill, y, (hipStream_t)stream);
}
int m3_glu(const float* x, int outer, int C, int inner, float* y, m3_stream stream) {
  return launch_glu(x, outer, C, inner, y, (hipStream_t)stream);
}
int m3_mask_conv2d_sample(const int32_t* len_in, int B, int left_padding, int stride, int32_t* len_out,
                          m3_stream stream) {
  return launch_mask_conv2d_sample(len_in, B, left_padding, stride, len_out, (hipStream_t)stream);
}
int m3_scale(const float* x, float scale, float* y, size_t n, m3_stream stream) {
  return launch_scale(x, scale, y, n, (hipStream_t)stream);
}
int m3_unary(const float* x, float* y, size_t n, int act, m3_stream stream) {
  return launch_unary(x, y, n, act, (hipStream_t)stream);
}
int m3_binary(const float* a, const float* b, float* y, const int64_t* shape, const int64_t* strides_a,
              const int64_t* strides_b, int ndim, int op, m3_stream stream) {
  return launch_binary_bcast(a, b, y, shape, strides_a, strides_b, ndim, op, (hipStream_t)stream);
}
int m3_permute(const float* x, float* y, const int64_t* out_shape, const int64_t* in_strides, int ndim,
               m3_stream stream) {
  return launch_permute(x, y, out_shape, in_strides, ndim, (hipStream_t)stream);
}
int m3_concat_last(const float* a, int da, const float* b, int db, float* y, size_t rows, m3_stream stream) {
  return launch_concat_last(a, da, b, db, y, rows, (hipStream_t)stream);
}
int m3_softmax(const float* x, float* y, size_t rows, int n, m3_stream stream) {
  return launch_softmax_lastdim(x, y, rows, n, (hipStream_t)stream);
}
int m3_batched_matmul(const float* a, const float* b, float* c, int batch, int M, int N, int K, int64_t stride_a,
                      int64_t stride_b, int transpose_b, m3_stream stream) {
  return launch_bmm(a, b, c, batch, M, N, K, stride_a, stride_b, transpose_b, (hipStream_t)stream);
}
int m3_pad2d(const float* x, size_t outer, int H, int W, int pre_h, int post_h, int pre_w, int post_w, float* y, m3_stream stream) {
  M3_REQUIRE(x && y, "pad2d: null pointer");
  return launch_pad2d(x, outer, H, W, pre_h, post_h, pre_w, post_w, y, (hipStream_t)stream);
}
int m3_depthwise_conv1d(const float* x, const float* w, const float* bias, int B, int C, int T, int K, int pad,
                        float* y, m3_stream stream) {
  return launch_depthwise_conv1d_nct(x, w, bias, B, C, T, K, pad, y, (hipStream_t)stream);
}

}  // extern "C"
