// Internal launcher interface of libm3asr_hip.so (C++ side; the C-ABI lives in include/m3asr.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace m3 {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_GLU = 3 };
enum { GEMM_A_PLAIN = 0, GEMM_A_CONCAT2 = 1, GEMM_A_CONV3X3S2 = 2 };

struct GemmParams {
  // operands: Y[M,Nout] = epi( pro(A)[M,K] . W[N,K]^T )
  const float* A = nullptr; int lda = 0;
  const float* A2 = nullptr; int lda2 = 0; int K1 = 0;   // CONCAT2: k<K1 from A, else A2[k-K1]
  const float* W = nullptr;                               // [N][K] row-major (fp32, or bf16 when w_bf16)
  int w_bf16 = 0;                                         // W holds bf16: operands rounded at the MFMA input, fp32 accumulate
  const float* bias = nullptr;                            // [N] or null
  float* Y = nullptr; int ldy = 0;
  int M = 0, N = 0, K = 0;
  int mode = GEMM_A_PLAIN;
  // prologue
  const float* ln_gamma = nullptr; const float* ln_beta = nullptr; float ln_eps = 0.f;
  // folded LayerNorm (affine already inside W / bias): wsum[n] = sum_k W'[n][k]; wbeta[n] = (W . beta)[n] (only with mask_in)
  const float* ln_wsum = nullptr; const float* ln_wbeta = nullptr;
  int ln_on_a2 = 0;                                       // CONCAT2: LayerNorm only the A2 half (router input)
  float* ln_out = nullptr; int ld_ln_out = 0;             // optional side output of the normalised rows
  const int32_t* row_len = nullptr; int rows_per_batch = 0;  // frame t = row % rows_per_batch is padded if t >= row_len[row / rows_per_batch]
  int mask_in = 0, mask_out = 0;
  // implicit 3x3 stride-2 conv over a channel-last (B,T1,F1,C) input -> rows (b,t2,f2)
  int conv_T1 = 0, conv_F1 = 0, conv_T2 = 0, conv_F2 = 0, conv_C = 0;
  // epilogue
  int act = ACT_NONE;
  float alpha = 1.f;
  const float* resid = nullptr; int ldr = 0;
  // grouped (per-expert) form of the tiled bf16 kernel: row tiles cut from acc_hist, optional row gather, slice-major W
  const int32_t* grp_acc = nullptr; int grp_E = 0; const int32_t* grp_pos = nullptr; int w_sliced = 0;
  const float* w_scale = nullptr;                         // fp8 weights (grouped forms): per-output-row scale [E][N]
  // bf16 activation copies (16-bit modes, tiled kernel only): A given as bf16 [M][lda], Y written as bf16, and / or an
  // additional bf16 copy Yb of the fp32 output (the residual stream stays fp32, its GEMM consumers read the copy)
  int a_bf16 = 0, y_bf16 = 0; void* Yb = nullptr; int ldyb = 0;
  // Row statistics of a bf16 operand for the folded LayerNorm of its consumer (gemm_bf16_dma.hip, where nothing passes
  // through registers while staging): per row kXbStatParts partial (sum, sum of squares) pairs of the bf16 values.
  // Yb_stats: the kernel that writes Yb also writes its column tile's partial; ln_stats: the LayerNorm GEMM reads them
  float* Yb_stats = nullptr; const float* ln_stats = nullptr; int ln_stat_parts = 0;
  // packed ragged batches: device-side count of live rows; work-groups whose first row lies beyond it exit (the row AT
  // the count is still computed: it carries the conv module's pad-frame constant, see dwconv_ln_silu_kernel)
  const int32_t* m_dev = nullptr;
  const int32_t* y_rows = nullptr;                        // grouped GEMM-2 only: output row m goes to row y_rows[m] of Y (expert-parallel: straight to its wire row)
  // implicit conv on a packed ragged batch: valid output frames per utterance; a tile whose rows (b, t2, f2) all lie past
  // the utterance's last frame is skipped (its output rows are never gathered into the packed layout)
  const int32_t* conv_len = nullptr;
  // filled by launch_gemm_f32
  int n_tiles = 0, m_tiles = 0, xcd_swizzle = 0;
};
int launch_gemm_f32(const GemmParams& p, hipStream_t stream);
bool gemm_f32_dual_fusable(const GemmParams& a, const GemmParams& b);   // two independent skinny fp32 GEMMs of one instantiation
int launch_gemm_f32_dual(const GemmParams& a, const GemmParams& b, hipStream_t stream);   // ... in ONE launch   // dispatches to launch_gemm_bf16w when p.w_bf16
int launch_gemm_bf16w(const GemmParams& p, hipStream_t stream);
const char* gemm_kernel_label(const GemmParams& p, bool splitk);   // the kernel these dispatchers will run (observability)
bool gemm_bf16w_uses_tiled(const GemmParams& p);   // the choice launch_gemm_bf16w makes for this problem (sizes / mode only)
// deep-K, few-tile fp32 problems (conv2 / subsampling Linear at short inputs): split-K tiled kernel + reduce (gemm_f32_splitk.hip)
int gemm_f32_splitk_plan(const GemmParams& p, size_t* ws_bytes);   // number of K ranges (0 = not applicable) and workspace
int launch_gemm_f32_splitk(const GemmParams& p, float* ws, size_t ws_bytes, hipStream_t stream);
int init_gemm_f32_splitk_kernels();
int init_gemm_bf16_tiled_kernels();
// bf16 A x bf16 W, LDS-DMA fed 128 x 128 x 64 tiles (gemm_bf16_dma.hip): the dense GEMMs of long batches in the 16-bit modes
constexpr int kXbStatParts = 4;                          // partial row statistics kept per row of a bf16 activation copy
bool gemm_bf16_dma_supports(const GemmParams& p);
int gemm_bf16_dma_col_tiles(const GemmParams& p);
bool gemm_bf16w_uses_dma(const GemmParams& p);           // the choice launch_gemm_bf16w makes
int launch_gemm_bf16_dma(const GemmParams& p, hipStream_t stream);
int init_gemm_bf16_dma_kernels();
// (sum, sum of squares) of every row of a bf16 matrix -> stats[row][kXbStatParts][2] (total in part 0, zeros elsewhere)
int launch_row_stats_bf16(const void* xb, int rows, int D, float* stats, hipStream_t stream);
int init_gemm_f32_tiled_kernels();    // same for the fp32 tiled kernels (gemm_f32_tiled.hip)   // once, outside graph capture (dynamic-LDS opt-in of the tiled kernels)

// ---- MoE indexing / scatter / gather (moe_index.hip) ----
int launch_moe_index(const int32_t* gate_idx, int S, int E, int32_t* mapping, int32_t* acc_hist,
                     int32_t* pos, hipStream_t stream);
// SoftmaxTopK + ScatterMapping fused (router logits [S][width] -> gate_idx, gate_value, mapping, acc, pos)
int launch_moe_gate_index(const float* logits, int width, const int32_t* row_len, int rows_per_batch, int S,
                          int32_t* gate_idx, float* gate_value, int32_t* mapping, int32_t* acc_hist, int32_t* pos,
                          hipStream_t stream);
// router (x half, folded LayerNorm) + softmax-top1 + index in one single-workgroup launch (S <= 256)
int launch_moe_route(const float* x, int ldx, int D, const float* wx, const float* wsum, const float* bias,
                     const float* eall, int ld_e, float ln_eps, const int32_t* row_len, int rows_per_batch, int S, int E,
                     int32_t* gate_idx, float* gate_value, int32_t* mapping, int32_t* acc_hist, int32_t* pos,
                     hipStream_t stream);
// router product on cat([embed, LayerNorm(x)]) with the normalised rows written out (moe_router.hip): one work-group per 16
// rows and all N <= 64 experts
bool moe_router_supports(int De, int D, int N);
bool moe_router_fuses_top1(int N);   // SoftmaxTopK in the router kernel's tail (gate_idx / gate_val arguments of launch_moe_router)
int init_moe_router_kernels();
int launch_moe_router(const float* emb, int lde, int De, const float* x, int ldx, int D, const float* W, const float* bias,
                      const float* gamma, const float* beta, float eps, float* xn, int ldxn, float* Y, int ldy, int M, int N,
                      const int32_t* m_dev, hipStream_t stream, int32_t* gate_idx = nullptr, float* gate_val = nullptr,
                      const int32_t* row_len = nullptr, int rows_per_batch = 0, void* xq = nullptr, float* xq_scale = nullptr);
int launch_local_scatter(const void* x, const int32_t* mapping, int S, int row_bytes, void* out, hipStream_t stream);
int launch_local_gather(const void* buf, const int32_t* mapping, int S, int row_bytes, void* out, hipStream_t stream);

// ---- expert-parallel exchange bookkeeping on the device (ep_exchange.hip) ----
int launch_ep_send_map(const int32_t* gate_idx, const int32_t* mapping, const int32_t* acc_hist, int S, int world, int e_loc,
                       int capacity, int32_t* map_send, void* wire, int row_bytes, hipStream_t stream, int32_t* overflow = nullptr);
int launch_ep_send_rows(const int32_t* gate_idx, const int32_t* mapping, const int32_t* acc_hist, int S, int world, int e_loc,
                        int capacity, int32_t* map_send, const void* x, void* wire, int row_bytes, hipStream_t stream, int32_t* overflow = nullptr);   // send map + scatter, one launch
int launch_ep_recv_gate(const void* wire, int world, int e_loc, int capacity, int row_bytes, int32_t* gate_recv,
                        hipStream_t stream);

// ---- grouped expert FFN (moe_expert.hip) ----
#ifndef M3_EXPERT_SLICE
#define M3_EXPERT_SLICE 64        // (16 / 32: experiment builds next to the tree, tools/exp_ffn_pair.py; the plan format assumes 64)
#endif
constexpr int kExpertSliceW16 = 64;            // the bf16 / e4m3 slab kernels are laid out for 64-wide slices
constexpr int kExpertSlice = M3_EXPERT_SLICE;  // hidden units per workgroup (16 per wave); m3asr/plan.py EXPERT_SLICE must match
size_t expert_ffn_slab_bytes(int S, int D, int F);
int init_expert_ffn_kernels();
int launch_expert_ffn_f32(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E,
                          int D, int F, const float* w1, const float* b1, const float* w2, int w2_sliced, float* slab,
                          const float* ln_gamma, const float* ln_beta, float ln_eps, hipStream_t stream);
// long batches (S >= 1024): two grouped LDS-tiled fp32 GEMMs (moe_expert_tiled_f32.hip); launch_expert_ffn_f32 switches
// to it by itself; these tell the combine step where / in how many slabs the result rows are
bool expert_ffn_f32_tiled(int S, int E, int D, int F);
float* expert_ffn_f32_rows(float* slab, int S, int E, int D, int F);
int expert_ffn_f32_slices(int S, int E, int D, int F);
int init_expert_ffn_f32_tiled_kernels();
int launch_expert_ffn_f32_tiled(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E,
                                int D, int F, const float* w1, const float* b1, const float* w2, int w2_sliced,
                                float* hbuf, float* ybuf, hipStream_t stream);
// bf16 weights (w1 [E][F][D], w2 as above), fp32 rows in / fp32 slab out (moe_expert_bf16.hip)
int init_expert_ffn_bf16_kernels();
int launch_expert_ffn_bf16w(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E,
                            int D, int F, const void* w1, const float* b1, const void* w2, int w2_sliced, float* slab,
                            hipStream_t stream, const float* b2 = nullptr, float* y_scatter = nullptr);   // y_scatter: tiled form only (expert_ffn_bf16_tiled)
// fp8 (e4m3) expert weights + per-row scales, dequantised to bf16 at the MFMA input (moe_expert_fp8.hip); same result layout as bf16
int init_expert_ffn_w8_kernels();
int launch_expert_ffn_w8(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D, int F,
                         const void* w1, const float* s1, const float* b1, const void* w2, const float* s2, int w2_sliced,
                         float* slab, hipStream_t stream);
// which form launch_expert_ffn_bf16w takes for this shape, and where / in how many slabs its result rows are
bool expert_ffn_bf16_tiled(int S, int E, int D, int F);
float* expert_ffn_bf16_rows(float* slab, int S, int E, int D, int F);
int expert_ffn_bf16_slices(int S, int E, int D, int F);
// what the 16-bit / fp8 dispatchers above run for a shape (wmode 1 = bf16 weights, 2 = fp8 weights), for the combine
// step and for launch accounting: where the result rows are, in how many partial slabs, how many kernel launches
float* expert_ffn_w16_rows(int wmode, float* slab, int S, int E, int D, int F);
int expert_ffn_w16_slices(int wmode, int S, int E, int D, int F);
int expert_ffn_w16_launches(int wmode, int S, int E, int D, int F);
const char* expert_ffn_w16_kernel(int wmode, int S, int E, int D, int F);
// fp8 arithmetic (e4m3 weights x e4m3 activations, fp8 MFMA), long batches (D = 512): moe_expert_fused_fp8.hip.
// wmode 3 in the helpers above = "fp8 weights + fp8 activations where this kernel applies, else the weight-only form"
bool expert_ffn_fused_fp8_applies(int S, int E, int D, int F);
int expert_ffn_fused_fp8_fsplit(int S, int E, int D, int F);
int init_expert_ffn_fused_fp8_kernels();
int launch_expert_ffn_fused_fp8(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D, int F,
                                const void* w1, const float* s1, const float* b1, const void* w2, const float* s2, int w2_sliced,
                                float h_scale, float* ybuf, hipStream_t stream, const void* xq = nullptr, const float* xq_scale = nullptr,
                                int32_t* fs_dev = nullptr);   // fs_dev: the kernel may split F finer than the host's choice and leaves the slab count there
int launch_quantize_rows_e4m3(const float* x, int ldx, int S, int D, void* xq, float* scale, hipStream_t stream);   // moe_expert_fused_fp8.hip
// fp8 weights, dispatcher: h_scale > 0 asks for fp8 activations (taken where the fused kernel applies)
bool expert_ffn_w8a8_fused(int S, int E, int D, int F);   // ... i.e. when this holds (moe_expert_bf16.hip)
int launch_expert_ffn_w8a8(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D, int F,
                           const void* w1, const float* s1, const float* b1, const void* w2, const float* s2, int w2_sliced,
                           float h_scale, float* slab, hipStream_t stream, const void* xq = nullptr, const float* xq_scale = nullptr,
                           int32_t* fs_dev = nullptr);
// long batches: two grouped GEMMs on the LDS-tiled bf16 core (gemm_bf16_tiled.hip); hbuf S*F bf16, ybuf S*D fp32
// b2 / y_scatter (optional): GEMM-2 adds the expert's b2 and writes row i of the sorted order to row pos[i] of y_scatter
// (the un-permute of the expert-parallel receive side, folded into the epilogue)
int launch_expert_ffn_bf16w_tiled(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E,
                                  int D, int F, const void* w1, const float* b1, const void* w2, int w2_sliced,
                                  void* hbuf, float* ybuf, hipStream_t stream, const float* b2 = nullptr, float* y_scatter = nullptr);
// out[s] = resid[s] + alpha * gate[s] * (b2[g_s] + sum_slices slab[slice][mapping[s]]), optional LayerNorm after
// SoftmaxTopK + ScatterMapping + grouped expert FFN in ONE launch (S <= 256 rows, all experts local, fp32): see moe_expert.hip
bool expert_ffn_f32_self_routing(int S, int E);
int launch_expert_route_ffn_f32(const float* x, int ldx, const float* logits, const int32_t* row_len, int rows_per_batch, int S, int E,
                                int D, int F, const float* w1, const float* b1, const float* w2, int w2_sliced, const float* b2,
                                float* slab, int32_t* gate_idx, float* gate_value, int32_t* mapping, int32_t* acc_hist, int32_t* pos,
                                hipStream_t stream, const float* ln_gamma = nullptr, const float* ln_beta = nullptr, float ln_eps = 0.f);
// grouped bf16 expert GEMMs on 256 x 256 x 64 LDS-DMA tiles (expert_gemm_g256.hip): saturating row counts (>= 512 rows per expert)
bool expert_ffn_bf16_g256(int S, int E, int D, int F);
int init_expert_gemm_g256_kernels();
int launch_rows_to_bf16(const float* x, int ldx, int S, int D, void* xb, hipStream_t stream);
int launch_expert_ffn_bf16_g256(const void* xb, int ldxb, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D, int F,
                                const void* w1, const float* b1, const void* w2, int w2_sliced, void* hbuf, float* ybuf,
                                hipStream_t stream);
int launch_moe_combine(const float* slab, int n_slices, const int32_t* mapping, const int32_t* gate_idx,
                       const float* gate_value, const float* b2, const float* resid, float alpha,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, float* out, int S, int D,
                       hipStream_t stream, void* out_bf16 = nullptr, float* out_stats = nullptr,
                       const int32_t* n_slices_dev = nullptr);   // n_slices_dev: the slab count is a device value (<= n_slices)

// ---- row-wise ops (rowops.hip) ----
int launch_layernorm(const float* x, const float* gamma, const float* beta, float eps, float* y, int rows, int D,
                     hipStream_t stream, void* y_bf16 = nullptr, float* y_stats = nullptr, const float* gamma2 = nullptr,
                     const float* beta2 = nullptr, float eps2 = 0.f, float* y2 = nullptr);   // gamma2: y2 = LN2(LN1(x)) in the same launch
int launch_softmax_top1(const float* logits, int ld, const int32_t* row_len, int rows_per_batch, int S, int E,
                        int32_t* idx, float* value, hipStream_t stream);
int launch_att_masked_softmax(const float* scores, const int32_t* len, int B, int H, int T1, int T2, float scale,
                              float* out, hipStream_t stream);
int launch_masked_fill(const float* x, const int32_t* len, int B, int C, int T, float fill, float* y,
                       hipStream_t stream);
int launch_glu(const float* x, int outer, int C, int inner, float* y, hipStream_t stream);
int launch_scale(const float* x, float scale, float* y, size_t n, hipStream_t stream);
int launch_mask_conv2d_sample(const int32_t* len_in, int B, int left_padding, int stride, int32_t* len_out,
                              hipStream_t stream);
int launch_subsample_lens(const int32_t* len_in, int B, int32_t* len_out, hipStream_t stream);
int launch_add(const float* a, const float* b, float* y, size_t n, hipStream_t stream);
int launch_binary_bcast(const float* a, const float* b, float* y, const int64_t* shape, const int64_t* sa,
                        const int64_t* sb, int nd, int op, hipStream_t stream);
int launch_unary(const float* x, float* y, size_t n, int act, hipStream_t stream);
int launch_permute(const float* x, float* y, const int64_t* out_shape, const int64_t* in_strides, int nd,
                   hipStream_t stream);
int launch_concat_last(const float* a, int da, const float* b, int db, float* y, size_t rows, hipStream_t stream);
int launch_softmax_lastdim(const float* x, float* y, size_t rows, int n, hipStream_t stream);
int launch_bmm(const float* a, const float* b, float* c, int batch, int M, int N, int K, int64_t sa, int64_t sb,
               int trans_b, hipStream_t stream);

// ---- fused rel-pos attention (attention.hip) ----
int launch_relpos_attention(const float* qkv, int ldq, const float* pmat, int ldp, const float* pos_u,
                            const float* pos_v, const int32_t* row_len, int B, int T, int H, int dk, float scale,
                            float* out, int ldo, hipStream_t stream, int out_bf16 = 0, const int32_t* row0 = nullptr,
                            int chunk = 0, int left_chunks = -1);
// the same as a value (what the engine keeps per stage so that two independent attention stages can share a launch)
struct AttArgs {
  const float* qkv = nullptr; int ldq = 0; const float* pmat = nullptr; int ldp = 0; const float *pos_u = nullptr, *pos_v = nullptr;
  const int32_t* row_len = nullptr; int B = 0, T = 0, H = 0, dk = 0; float scale = 1.f; float* out = nullptr; int ldo = 0, out_bf16 = 0;
  const int32_t* row0 = nullptr; int chunk = 0, left_chunks = 0;
};
int launch_relpos_attention_args(const AttArgs& a, hipStream_t stream);
bool relpos_attention_dual_fusable(const AttArgs& a, const AttArgs& b);
int launch_relpos_attention_dual(const AttArgs& a, const AttArgs& b, hipStream_t stream);   // chunk > 0: static chunk mask (utils/mask.py:42-75)
// chunk-by-chunk form: C query frames per utterance, K / V history [B][cap][2D] appended to in place, device-side chunk counter
int launch_relpos_attention_stream(const float* qkv, int ldq, float* hist, int cap, const float* pmat, int ldp, const float* pos_u,
                                   const float* pos_v, const int32_t* chunk_len, const int32_t* step, int B, int C, int H, int dk,
                                   float scale, float* out, int ldo, int left_chunks, hipStream_t stream);

// the same on bf16 rows (16-bit modes, T' <= 128): qkv bf16 [B*T][ldq], out bf16; one work-group per (utterance, head)
bool relpos_attention_bf16_supports(int T, int dk);
int init_relpos_attention_bf16_kernels();
int launch_relpos_attention_bf16(const void* qkv, int ldq, const float* pmat, int ldp, const float* pos_u, const float* pos_v,
                                 const int32_t* row_len, int B, int T, int H, int dk, float scale, void* out, int ldo,
                                 hipStream_t stream, const int32_t* row0 = nullptr, int chunk = 0, int left_chunks = -1);

// ---- conv module / subsampling (conv.hip) ----
int launch_dwconv_ln_silu(const float* z, const float* w_kc, const float* bias, const float* gamma,
                          const float* beta, float eps, int B, int T, int D, int K, float* out, hipStream_t stream,
                          int out_bf16 = 0, const int32_t* pad_of = nullptr, const int32_t* row0 = nullptr,
                          const int32_t* row_len = nullptr, const float* causal_left_fill = nullptr);   // non-null: causal conv (lorder K-1)
int launch_pad2d(const float* x, size_t outer, int H, int W, int pre_h, int post_h, int pre_w, int post_w, float* y, hipStream_t stream);
int launch_advance_counter(int32_t* counter, int by, hipStream_t stream);
int launch_fill_rows(const float* row, int D, float* out, size_t rows, hipStream_t stream);
// the same as a value (what the engine keeps per stage so that two independent conv modules can share a launch)
struct DwArgs {
  const float *z = nullptr, *w_kc = nullptr, *bias = nullptr, *gamma = nullptr, *beta = nullptr; float eps = 1e-5f;
  int B = 0, T = 0, D = 0, K = 0; float* out = nullptr; int out_bf16 = 0;
  const int32_t *pad_of = nullptr, *row0 = nullptr, *row_len = nullptr; const float* causal_left_fill = nullptr;
};
int launch_dwconv_ln_silu_args(const DwArgs& a, hipStream_t stream);
bool dwconv_dual_fusable(const DwArgs& a, const DwArgs& b);
int launch_dwconv_ln_silu_dual(const DwArgs& a, const DwArgs& b, hipStream_t stream);
int launch_dwconv_ln_silu_stream(const float* z, const float* w_kc, const float* bias, const float* gamma, const float* beta,
                                 float eps, int B, int T, int D, int K, float* out, float* cache_pair, const int32_t* step,
                                 const int32_t* chunk_len, hipStream_t stream, int out_bf16 = 0);
// packed (padding-free) rows of a ragged batch (rowops.hip): plan from the valid lengths; padded output from packed rows
int launch_pack_plan(int32_t* len, int B, int T, int32_t* row0, int32_t* pad_of, hipStream_t stream,
                     const int32_t* feat_len = nullptr);   // feat_len: form the subsampled lengths here too (len becomes an output)
int launch_unpack_rows(const float* in, const int32_t* row0, int B, int T, int n, float* out, hipStream_t stream);
int launch_conv1_relu(const float* feat, const float* w9c, const float* bias, const float* cmvn_mean,
                      const float* cmvn_istd, int B, int T, int idim, int C, float* out, hipStream_t stream, int relu = 1,
                      int out_bf16 = 0, const int32_t* feat_len = nullptr, int32_t* lens_out = nullptr);   // feat_len: also the subsampled lengths
int launch_cmvn(const float* x, const int32_t* len, const float* mean, const float* istd, int B, int T, int D,
                float* y, hipStream_t stream);
int launch_log_softmax_bias(const float* x, const float* bias, float* y, size_t rows, int n, hipStream_t stream);
// decode.hip: CTC search on the logits + the streaming operators
int launch_ctc_greedy(const float* logits, const int32_t* len, int B, int T, int V, int blank, int32_t* frame_ids,
                      int32_t* tokens, int32_t* n_tokens, hipStream_t stream);
int launch_ctc_topk(const float* logits, size_t rows, int V, int k, float* top_logp, int32_t* top_idx, hipStream_t stream);
int ctc_prefix_beam_search_host(const float* top_logp, const int32_t* top_idx, int T, int k, int beam, int blank,
                                int32_t* hyp_tokens, int32_t* hyp_len, float* hyp_score, int32_t* n_hyps);
int launch_cat_split_cache(const void* in_cache, const void* input, int B, int cache_dim, int input_dim, void* output,
                           void* out_cache, hipStream_t stream);
int launch_att_stream_softmax(const float* scores, const int32_t* decode_frame_num, const int32_t* mask_idx, int B, int N,
                              int ld, int cache_len, float scale, float* out, hipStream_t stream);
int launch_rel_positional_encoding(const float* x, const float* pe, int pe_len, const int32_t* frame_num, int max_offset,
                                   float scale, int B, int T, int D, float* y, float* pos_emb, int32_t* frame_num_out,
                                   hipStream_t stream);
int launch_depthwise_conv1d_nct(const float* x, const float* w, const float* bias, int B, int C, int T, int K,
                                int pad, float* y, hipStream_t stream);

}  // namespace m3
