/* m3asr.h -- C ABI of libm3asr_hip.so, the MI355X (gfx950) replacement for the reference's
 * TRTAPI++/plugin library (libtrtplugin++.so) on the 3M-ASR Conformer-MoE encoder hot path.
 *
 * Conventions (mirroring the reference's plugin ABI, fmoe_expert_plugin.h:45-73):
 *   - every tensor and every workspace is a CALLER-OWNED DEVICE pointer; weights are ordinary inputs,
 *     never copied or owned by an op (README.md:225); plain pointers and sizes only, no framework types;
 *   - `stream` is a hipStream_t passed as void*; ops only enqueue work on it (no host sync, no
 *     allocation -> every entry point is hipGraph-capturable), unlike the reference's FMoE enqueue
 *     which synchronises twice per layer (fmoe_expert_plugin.cpp:75-78,130).  One caveat: kernels that need more than
 *     64 KB of LDS opt in with hipFuncSetAttribute the FIRST time their entry point runs in a process (not a stream
 *     operation) -- call an entry point once outside a capture before capturing it; m3_engine_* does this at prepare;
 *   - return value: 0 = ok, non-zero = failure (reference: `int enqueue(...)`), message via
 *     m3_last_error() (reference only logs, common/common.h:26-38);
 *   - dtype codes 0/1/2 = the reference's HelperConfig.plugin_data_type (builder_helper.py:47-57).
 * Row layouts are row-major; "S" = B*T' tokens, D = idim, F = hidden_units, E = num_expert.
 */
#ifndef M3ASR_H_
#define M3ASR_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3ASR_ABI_VERSION 9

typedef void* m3_stream; /* hipStream_t */

enum m3_dtype { M3_F32 = 0, M3_F16 = 1, M3_I8 = 2, M3_I32 = 3, M3_BF16 = 4, M3_FP8 = 5 /* OCP e4m3 */ };

/* activation / element-wise codes shared by several entry points */
enum m3_act { M3_ACT_NONE = 0, M3_ACT_RELU = 1, M3_ACT_SILU = 2, M3_ACT_GLU = 3, M3_ACT_SIGMOID = 4, M3_ACT_LOG = 5 };
enum m3_binop { M3_OP_SUM = 0, M3_OP_PROD = 1 };

/* ------------------------------------------------------------------------------------------------
 * Library / registry.   Replaces: initLibNvInferPlugins + getPluginRegistry (plugin/exports.map:18-27),
 * init_trt_plugin_plus (trt_plugin_plus.h:23) and PluginCreatorRegistry lookup (trt_plugin_plus.cpp:56-123).
 * ---------------------------------------------------------------------------------------------- */
int m3_abi_version(void);
const char* m3_last_error(void);
/* 1 if a creator for (name, version) is registered, else 0.  Names are the reference's plugin names. */
int m3_registry_lookup(const char* plugin_name, const char* plugin_version);
int m3_registry_count(void);
const char* m3_registry_name(int index);

/* ------------------------------------------------------------------------------------------------
 * Generic plugin objects.   Replaces IPluginCreator::createPlugin / IPluginV2DynamicExt
 * {getOutputDimensions, getWorkspaceSize, enqueue, serialize, clone, destroy} for the eight plugins of
 * the hot path (fmoe_expert_plugin.h:45-73 is the template all of them follow).
 * ---------------------------------------------------------------------------------------------- */
typedef struct m3_tensor { /* = nvinfer1::PluginTensorDesc + data pointer */
  void* data;
  int32_t dtype; /* enum m3_dtype */
  int32_t ndim;
  int64_t shape[8];
} m3_tensor;

enum m3_field_type { M3_FIELD_FLOAT32 = 1, M3_FIELD_INT32 = 5 }; /* values of nvinfer1::PluginFieldType */
typedef struct m3_field {                                          /* = nvinfer1::PluginField */
  const char* name;
  const void* data;
  int32_t type;
  int32_t length;
} m3_field;

typedef struct m3_plugin m3_plugin;

/* NULL on unknown plugin or bad/missing attributes (reference creators return nullptr,
 * fmoe_expert_plugin.cpp:356-359). */
m3_plugin* m3_plugin_create(const char* plugin_name, const char* plugin_version, const m3_field* fields,
                            int n_fields);
m3_plugin* m3_plugin_clone(const m3_plugin* plugin);
void m3_plugin_destroy(m3_plugin* plugin);
const char* m3_plugin_type(const m3_plugin* plugin);
int m3_plugin_num_outputs(const m3_plugin* plugin);
/* fills outputs[i].{dtype,ndim,shape} from the input descriptors (data pointers ignored) */
int m3_plugin_output_dims(const m3_plugin* plugin, const m3_tensor* inputs, int n_in, m3_tensor* outputs,
                          int n_out);
size_t m3_plugin_workspace_size(const m3_plugin* plugin, const m3_tensor* inputs, int n_in,
                                const m3_tensor* outputs, int n_out);
int m3_plugin_enqueue(m3_plugin* plugin, const m3_tensor* inputs, int n_in, m3_tensor* outputs, int n_out,
                      void* workspace, size_t workspace_bytes, m3_stream stream);
/* POD serialisation of the attributes (reference: serialize.hpp:36-52) */
size_t m3_plugin_serialization_size(const m3_plugin* plugin);
int m3_plugin_serialize(const m3_plugin* plugin, void* buffer, size_t bytes);
m3_plugin* m3_plugin_deserialize(const char* plugin_name, const char* plugin_version, const void* buffer,
                                 size_t bytes);

/* ------------------------------------------------------------------------------------------------
 * MoE hot path, direct entry points (what FMoEExpertPluginDynamic's enqueue is made of).
 * ---------------------------------------------------------------------------------------------- */
/* Replaces ComputeScatterMapping (fmoe_expert_kernel.h:26-27; fmoe_expert_kernel.cu:25-90).
 * gate_idx[S] int32 in [0,E) (or <0 = dropped row) -> mapping[S], acc_histogram[E+1], pos[S] (inverse
 * permutation, may be NULL).  Stable within an expert. */
int m3_moe_scatter_mapping(const int32_t* gate_idx, int S, int num_expert, int32_t* mapping,
                           int32_t* acc_histogram, int32_t* pos, m3_stream stream);
/* Replaces ComputeScatterMappingCopy (fmoe_expert_kernel.cu:92-128) = FastMoE local_scatter
 * (fmoe/functions.py:72):  out[mapping[s]] = x[s];  rows of row_bytes (multiple of 16). */
int m3_moe_local_scatter(const void* x, const int32_t* mapping, int S, int row_bytes, void* out,
                         m3_stream stream);
/* Replaces ComputeGatherrMappingCopy (fmoe_expert_kernel.cu:191-227) = FastMoE local_gather
 * (fmoe/functions.py:194):  out[s] = buf[mapping[s]] (0 for dropped rows). */
int m3_moe_local_gather(const void* buf, const int32_t* mapping, int S, int row_bytes, void* out,
                        m3_stream stream);
/* Hidden units per work-group of the grouped expert FFN: the engine's expert w_2 is stored slice-major
 * [E][F/slice][D][slice] (m3asr/plan.py); plugin-path weights keep the reference layout [E][D][F]. */
int m3_moe_expert_slice(void);
/* Workspace of m3_moe_expert_ffn / FMoEExpertPluginDynamic (reference layout: fmoe_expert_plugin.cpp:224-239). */
size_t m3_moe_expert_workspace_size(int S, int num_expert, int idim, int hidden_units);
/* Replaces compute_fmoe_expert (fmoe_expert_plugin.cpp:36-142): y[s] = SiLU(x[s] W1[g]^T + b1[g]) W2[g]^T + b2[g]
 * for g = gate_idx[s] >= 0, else 0.  x,y [S][D] f32; w1 [E][F][D], b1 [E][F], w2 [E][D][F], b2 [E][D]
 * (FMoELinear layout, fmoe/layers.py:34-38).  Optional fused epilogue (all may be NULL / 1.0):
 *   y[s] = resid[s] + alpha * gate_value[s] * y[s], then LayerNorm(ln_gamma, ln_beta, ln_eps). */
int m3_moe_expert_ffn(const float* x, const int32_t* gate_idx, const float* w1, const float* b1,
                      const float* w2, const float* b2, int S, int num_expert, int idim, int hidden_units,
                      const float* gate_value, const float* resid, float alpha, const float* ln_gamma,
                      const float* ln_beta, float ln_eps, float* y, void* workspace, size_t workspace_bytes,
                      m3_stream stream);
/* The same with bf16 expert weights (w1 / w2 point at bf16 [E][F][D] / [E][D][F]; biases, rows, epilogue fp32):
 * the half-precision mode the reference declares (`data_type` plugin field, fmoe_expert_plugin.cpp:331-354) but
 * asserts on (:264-266).  Rows and H are rounded to bf16 at the MFMA inputs, accumulation is fp32. */
int m3_moe_expert_ffn_bf16(const float* x, const int32_t* gate_idx, const void* w1, const float* b1,
                           const void* w2, const float* b2, int S, int num_expert, int idim, int hidden_units,
                           const float* gate_value, const float* resid, float alpha, const float* ln_gamma,
                           const float* ln_beta, float ln_eps, float* y, void* workspace, size_t workspace_bytes,
                           m3_stream stream);
/* fp8 expert weights (W8A16): w1 / w2 hold OCP e4m3 bytes, W[e][n][k] ~ scale[e][n] * q[e][n][k] with w1_scale [E][F],
 * w2_scale [E][D]; the weights are dequantised to bf16 at the MFMA input (exact), rows and H are rounded to bf16 as in
 * the bf16 form, accumulation fp32.  1.05 MB per touched expert at D=512, F=1024. */
int m3_moe_expert_ffn_fp8(const float* x, const int32_t* gate_idx, const void* w1, const float* w1_scale,
                          const float* b1, const void* w2, const float* w2_scale, const float* b2, int S, int num_expert,
                          int idim, int hidden_units, const float* gate_value, const float* resid, float alpha,
                          const float* ln_gamma, const float* ln_beta, float ln_eps, float* y, void* workspace,
                          size_t workspace_bytes, m3_stream stream);
/* fp8 ARITHMETIC (A8W8; the path the reference's --int8 flag names, builder.py:39-49): e4m3 weights as above, the rows
 * quantised to e4m3 with a per-row dynamic scale (amax / 448) while they are loaded, H quantised with the static per-layer
 * scale h_scale (calibrated: amax of H x 1.25 / 448), products on v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulation.  Taken
 * where the fused fp8 kernel applies (m3_moe_expert_ffn_fp8a8_active: idim 512, >= 4096 rows, >= 64 rows per expert);
 * shorter inputs are weight-streaming bound and run the weight-only form of m3_moe_expert_ffn_fp8 (identical signature
 * otherwise). */
int m3_moe_expert_ffn_fp8a8(const float* x, const int32_t* gate_idx, const void* w1, const float* w1_scale,
                            const float* b1, const void* w2, const float* w2_scale, const float* b2, float h_scale, int S,
                            int num_expert, int idim, int hidden_units, const float* gate_value, const float* resid,
                            float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps, float* y,
                            void* workspace, size_t workspace_bytes, m3_stream stream);
int m3_moe_expert_ffn_fp8a8_active(int S, int num_expert, int idim, int hidden_units);
/* ABI 9.  The same operator on rows that are ALREADY quantised the way it quantises them itself: xq [S][idim] e4m3, xq_scale [S]
 * (x = xq * xq_scale per row; what m3_quantize_rows_e4m3 and, inside the engine, the router kernel write).  Where the fused
 * kernel applies (m3_moe_expert_ffn_fp8a8_active) x is not read and may be NULL, and the result is bit-identical to
 * m3_moe_expert_ffn_fp8a8 on the fp32 rows; elsewhere the weight-only form runs on x.  Replaces nothing in the reference (its
 * --int8 path asserts, builder.py:39-49); it is the hand-over the whole-encoder engine uses between its router kernel and its
 * expert kernel, exposed so that it can be tested at the boundary. */
int m3_moe_expert_ffn_fp8a8_xq(const float* x, const void* xq, const float* xq_scale, const int32_t* gate_idx, const void* w1,
                               const float* w1_scale, const float* b1, const void* w2, const float* w2_scale, const float* b2,
                               float h_scale, int S, int num_expert, int idim, int hidden_units, const float* gate_value,
                               const float* resid, float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps,
                               float* y, void* workspace, size_t workspace_bytes, m3_stream stream);
/* rows -> OCP e4m3 with one dynamic scale per row: scale[s] = amax(x[s]) / 448 (1e-30 floor), xq = round-to-nearest-even,
 * saturating (x[s] / scale[s]).  idim must be 512 (one wave per row). */
int m3_quantize_rows_e4m3(const float* x, int ldx, int S, int idim, void* xq, float* scale, m3_stream stream);
/* The tail of the MoE layer on rows that are already in scattered (expert-sorted) order, e.g. rows that came back
 * from the expert-parallel all-to-all:  out[s] = LayerNorm( resid[s] + alpha * gate_value[s] * rows[mapping[s]] )
 * (rows with mapping < 0 contribute 0; gate_value / resid / ln_* may be NULL).  = local_gather
 * (fmoe/functions.py:194) + addProd + addScale + addAdd + norm_final (fmoe_transformer.py:145-166). */
int m3_moe_combine(const float* rows, const int32_t* mapping, const float* gate_value, const float* resid,
                   float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps, float* out, int S,
                   int idim, m3_stream stream);
/* The same, also writing the bf16 copy of `out` that engines with bf16 activation operands keep of the residual stream
 * (engine buffer "xb"; out_bf16 = [S][idim] bf16, may be NULL): lets the expert-parallel driver stand in for the
 * engine's own combine stage in the 16-bit modes. */
int m3_moe_combine_bf16(const float* rows, const int32_t* mapping, const float* gate_value, const float* resid,
                        float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps, float* out,
                        void* out_bf16, int S, int idim, m3_stream stream);
/* Expert-parallel exchange without a host round trip (replaces the host logic of FastMoE's moe_prepare_forward /
 * MOEScatter / MOEGather, trainer_3m_fix/fmoe/functions.py:13-52,63-86,175-199, which reads the counts back to size its
 * all-to-all-v).  The wire buffer has a FIXED shape [world][1 + capacity][row_bytes]: chunk j = what this rank sends to
 * (after the equal-split all-to-all: received from) rank j = one header row with the e_loc row counts of the chunk
 * (int32: the count exchange rides in the payload) + up to `capacity` rows sorted by rank j's local expert id.
 *   m3_ep_send_map : from the local index step (gate_idx = GLOBAL expert id or -1, mapping, acc_histogram over
 *                    world * e_loc experts) -> map_send[s] = wire row of token s (-1: dropped) and the headers written
 *                    into `wire`; local_scatter(x, map_send) then fills the payload, and the reply comes back at the same
 *                    row.  capacity >= S (a rank may send everything to one peer).
 *   m3_ep_recv_gate: from the headers of the received chunks -> gate_recv[world * (1 + capacity)] = local expert id of
 *                    every received wire row (-1: header / unused), the gate input of m3_moe_expert_ffn. */
int m3_ep_send_map(const int32_t* gate_idx, const int32_t* mapping, const int32_t* acc_histogram, int S, int world, int e_loc,
                   int capacity, int32_t* map_send, void* wire, int row_bytes, m3_stream stream);
int m3_ep_recv_gate(const void* wire, int world, int e_loc, int capacity, int row_bytes, int32_t* gate_recv, m3_stream stream);
/* Router of the MoE feed-forward (positionwise_feed_forward.py:169-180,225 + norm_ff, fmoe_transformer.py:138-141):
 * logits[S][num_expert] = cat([embed (S, embed_dim), LayerNorm(x) (S, idim)]) . w^T (+ bias), w [num_expert][embed_dim + idim]
 * fp32 row-major; xn (may be NULL) receives LayerNorm(x), the expert FFN's input.  num_expert <= 64, dims multiples of 64. */
int m3_moe_router(const float* embed, int ld_embed, int embed_dim, const float* x, int ldx, int idim, const float* w,
                  const float* bias, const float* ln_gamma, const float* ln_beta, float ln_eps, float* xn, int ld_xn,
                  float* logits, int ld_logits, int S, int num_expert, m3_stream stream);
/* Replaces ComputeSoftmaxAndTop1 (softmax_topk_kernel.cu:88-120): logits [S][ld] -> idx[S], value[S];
 * frames t >= len[b] (t = s % rows_per_batch, b = s / rows_per_batch) get idx -1 / value 0; len may be NULL. */
int m3_softmax_top1(const float* logits, int ld, const int32_t* len, int rows_per_batch, int S, int width,
                    int32_t* idx, float* value, m3_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Dense building blocks (TensorRT-native layers of the reference + the small plugins).
 * ---------------------------------------------------------------------------------------------- */
/* Linear / point-wise conv with fused prologue+epilogue.  Replaces addLinear_ (torch_network_helper.py:573-605),
 * addConv1d k=1 (:199-225), LayerNorm plugin (layer_norm_plugin.cpp:78-113), masked_fill, GLU, SiLU/ReLU,
 * addScale + addAdd.   y[M][ldy] = resid + alpha * mask_out( act( LN(mask_in(a))[M][K] . w[N][K]^T + bias ) ).
 * a2 != NULL: A = cat([a (K1 cols), a2 (K-K1 cols)], -1) (router input, positionwise_feed_forward.py:225).
 * act = M3_ACT_GLU halves the output width (columns n and n+N/2 are paired, torch GLU dim=-1). */
typedef struct m3_linear_desc {
  const float* a; int32_t lda;
  const float* a2; int32_t lda2; int32_t k1;
  const void* w;   /* [N][K] row-major; fp32, or bf16 when weight_dtype = M3_BF16 */
  const float* bias;
  float* y; int32_t ldy;
  int32_t M, N, K;
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  /* folded LayerNorm (affine pre-multiplied into w / bias by the plan packer): the kernel normalises its
   * OUTPUT, y = rstd*(a.w^T - mean*ln_wsum) + bias; ln_wbeta = w.beta, needed only together with mask_in */
  const float* ln_wsum; const float* ln_wbeta;
  const int32_t* len; int32_t rows_per_batch; int32_t mask_in; int32_t mask_out;
  int32_t act; float alpha;
  const float* resid; int32_t ldr;
  /* M3_F32 (exact fp32 MFMA) or M3_BF16: weights stored bf16, A rounded to bf16 at the MFMA input,
   * fp32 accumulate and fp32 epilogue (the reference's plugin_data_type = 1, builder_helper.py:47-57; bf16
   * replaces fp16 on CDNA4).  bf16 supports plain A (no a2) and the folded LayerNorm only; K % 32 == 0. */
  int32_t weight_dtype;
  /* bf16 activation operands (16-bit modes, long batches; what the engine does between the GEMMs of a block):
   *   a_dtype = M3_BF16: `a` points at bf16 rows (lda in elements, % 8 == 0);  y_dtype = M3_BF16: `y` receives bf16;
   *   y_copy_bf16: an additional bf16 copy of the fp32 result (ld_copy in elements), e.g. of the residual stream;
   *   y_copy_stats: with it, per row and per 128-column tile the (sum, sum of squares) of the bf16 values stored
   *     ([M][N/128][2] floats) -- the row statistics a folded-LayerNorm GEMM needs when its bf16 operand goes to LDS
   *     without passing through registers (LDS-DMA kernel); ln_stats / ln_stat_parts: such statistics of `a` (their parts
   *     are summed), required with ln_wsum when a_dtype = M3_BF16 and the LDS-DMA kernel is to be used.  Only the LDS-DMA
   *     kernel (M >= 4096 rows, bf16 A, K % 64 == 0) reads or writes them: a call that passes either and lands on another
   *     kernel FAILS (non-zero status) rather than leaving stale statistics behind.
   * All zero / NULL = fp32 activations as before. */
  int32_t a_dtype, y_dtype;
  void* y_copy_bf16; int32_t ld_copy;
  float* y_copy_stats;
  const float* ln_stats; int32_t ln_stat_parts;
} m3_linear_desc;
int m3_linear(const m3_linear_desc* desc, m3_stream stream);
/* The same with a caller-owned workspace: deep-K problems with few output tiles (K >= 4096, e.g. the subsampling Linear of
 * a single utterance) run as a split-K tiled kernel + fixed-order reduce when m3_linear_workspace_size(desc) > 0 bytes
 * are provided; otherwise identical to m3_linear. */
size_t m3_linear_workspace_size(const m3_linear_desc* desc);
int m3_linear_ws(const m3_linear_desc* desc, void* workspace, size_t workspace_bytes, m3_stream stream);

/* LayerNormPluginDynamic (layer_norm_plugin.cpp:78-113) -- with eps, as PyTorch. */
int m3_layer_norm(const float* x, const float* gamma, const float* beta, float eps, float* y, int rows, int dim,
                  m3_stream stream);
/* Fused rel-pos attention core; replaces attention.py:347-384 + :199-236 (shuffles, 3 batched matmuls,
 * AttMaskedSoftmaxPluginDynamic).  qkv [B*T][ldq] = (q|k|v), p [T][ldp], pos_u/pos_v [H][dk], out [B*T][ldo]. */
int m3_relpos_attention(const float* qkv, int ldq, const float* p, int ldp, const float* pos_u,
                        const float* pos_v, const int32_t* len, int B, int T, int H, int dk, float scale,
                        float* out, int ldo, m3_stream stream);
/* The same operator on bf16 rows (16-bit modes of long batches): qkv and out are bf16 ([B*T][ldq] / [B*T][ldo], strides in
 * elements), p / pos_u / pos_v stay fp32; bf16 MFMA, fp32 softmax; T <= 128 keys, dk 64 or 128. */
int m3_relpos_attention_bf16(const void* qkv, int ldq, const float* p, int ldp, const float* pos_u, const float* pos_v,
                             const int32_t* len, int B, int T, int H, int dk, float scale, int chunk, int left_chunks,
                             void* out, int ldo, m3_stream stream);
/* The fp32 core with the static chunk mask of the streaming encoders (utils/mask.py:42-75,127-134): chunk > 0: query i
 * sees keys [max((i / chunk - left_chunks) chunk, 0), min((i / chunk + 1) chunk, T)) (all left chunks when left_chunks < 0)
 * and < len[b]; rows with no visible key give zeros.  chunk <= 0: identical to m3_relpos_attention. */
int m3_relpos_attention_chunk(const float* qkv, int ldq, const float* p, int ldp, const float* pos_u, const float* pos_v,
                              const int32_t* len, int B, int T, int H, int dk, float scale, int chunk, int left_chunks,
                              float* out, int ldo, m3_stream stream);
/* Depthwise conv (k odd, pad (k-1)/2) + LayerNorm (gamma NULL = none) + SiLU on channel-last rows;
 * replaces convolution.py:134-152.  w_kc [K][D] = depthwise weight (D,1,K) transposed. */
int m3_dwconv_ln_silu(const float* z, const float* w_kc, const float* bias, const float* gamma,
                      const float* beta, float eps, int B, int T, int D, int K, float* out, m3_stream stream);
/* Conv2dSubsampling4 (subsampling.py:103-145) on channel-last data: conv1 (1->C, 3x3, s2) + ReLU. */
int m3_subsample_conv1(const float* feat, const float* w9c, const float* bias, int B, int T, int idim, int C,
                       float* out, m3_stream stream);
/* the same with global CMVN folded into the input read (mean / istd [idim], may be NULL) */
int m3_subsample_conv1_cmvn(const float* feat, const float* w9c, const float* bias, const float* cmvn_mean,
                            const float* cmvn_istd, int B, int T, int idim, int C, float* out, m3_stream stream);
/* The plain operators behind network_helper.addConv2d (torch_network_helper.py:227-251; nn.Conv2d 3x3 / stride 2 / no
 * padding, channel-last data): act = M3_ACT_NONE gives the convolution alone, M3_ACT_RELU the fused form above. */
int m3_conv2d_3x3s2_first(const float* feat, const float* w9c, const float* bias, int B, int T, int idim, int C, int act,
                          float* out, m3_stream stream);
int m3_conv2d_3x3s2(const float* in, const float* w, const float* bias, int B, int T1, int F1, int C, int act, float* out,
                    m3_stream stream);
/* second conv (C->C, 3x3, s2) + ReLU as implicit GEMM: in (B,T1,F1,C) -> out (B,T2,F2,C); w [C][3][3][C]. */
int m3_subsample_conv2(const float* in, const float* w, const float* bias, int B, int T1, int F1, int C,
                       float* out, m3_stream stream);

/* Front / back end of the acoustic score.  Global CMVN: y = (x - mean[d]) * istd[d] on frames
 * t < len[b] (the reference's unfinished CmvnPlugin, incomplete_plugin/cmvn_plugin/cmvn_plugin.cu:17-43).
 * Score: y = log_softmax(x) + bias per row, bias = -log(prior) (builder.py:77-88, prior_prob_kernel.cu:11-26). */
int m3_cmvn(const float* x, const int32_t* len, const float* mean, const float* istd, int B, int T, int D, float* y,
            m3_stream stream);
int m3_log_softmax_bias(const float* x, const float* bias, float* y, size_t rows, int n, m3_stream stream);

/* After the encoder: CTC search on the logits.
 * Greedy (model/encoder.py:156-180): ids = argmax over V per frame (first maximum wins), then per utterance drop repeats
 * and blanks over frames t < len[b] (len NULL = all T).  frame_ids [B*T] receives the per-frame argmax (also the
 * kernel's scratch), tokens [B][T] the collapsed ids padded with -1, n_tokens [B] their counts.  All device pointers. */
int m3_ctc_greedy(const float* logits, const int32_t* len, int B, int T, int V, int blank, int32_t* frame_ids,
                  int32_t* tokens, int32_t* n_tokens, m3_stream stream);
/* First beam prune of the prefix beam search on the device (encoder.py:224-231): per row log_softmax, then the k best
 * (value desc, index asc) -> top_logp / top_idx [rows][k]. */
int m3_ctc_topk(const float* logits, size_t rows, int V, int k, float* top_logp, int32_t* top_idx, m3_stream stream);
/* The prefix recursion and second prune (encoder.py:232-275) -- a HOST routine over HOST copies of m3_ctc_topk's output
 * for one utterance of T frames.  Writes at most `beam` hypotheses, best first: hyp_tokens [beam][T] (-1 padded),
 * hyp_len [beam], hyp_score [beam] = log(p_blank + p_non_blank), *n_hyps. */
int m3_ctc_prefix_beam_search(const float* top_logp, const int32_t* top_idx, int T, int k, int beam, int blank,
                              int32_t* hyp_tokens, int32_t* hyp_len, float* hyp_score, int32_t* n_hyps);

/* Streaming operators of the reference's plugin library (built there but not registered, trt_plugin_plus.cpp:155-156).
 * CatSplitCachePluginDynamic (cat_split_cache_kernel.cu:30-107), 4-byte elements: output [B][cache_dim+input_dim] =
 * in_cache ++ input, out_cache [B][cache_dim] = the last cache_dim values of output. */
int m3_cat_split_cache(const void* in_cache, const void* input, int B, int cache_dim, int input_dim, void* output,
                       void* out_cache, m3_stream stream);
/* AttStreamSoftmaxPluginDynamic (att_stream_softmax_kernel.cu:28-191): scores [B][N][ld]; row (b,n) valid on
 * [max(0, ld - decode_frame_num[b]), min(ld, min(ld, mask_idx[b]) + cache_len)); out = exp((x - max) * scale) / sum
 * there, 0 elsewhere. */
int m3_att_stream_softmax(const float* scores, const int32_t* decode_frame_num, const int32_t* mask_idx, int B, int N,
                          int ld, int cache_len, float scale, float* out, m3_stream stream);
/* RelPositionalEncodingPluginDynamic (rel_positional_encoding_kernel.cu:62-69; streaming contract :108-111):
 * y = x * scale on (B,T,D); pos_emb [T][D] = pe[off : off+T], off = frame_num[0] (device int32 [B]; NULL = 0);
 * frame_num_out[b] = frame_num[b] + T (distinct buffer; may be NULL).  pe has pe_len positions; max_offset is the
 * caller's bound on frame_num[0], checked against pe_len on the host. */
int m3_rel_positional_encoding(const float* x, const float* pe, int pe_len, const int32_t* frame_num, int max_offset,
                               float scale, int B, int T, int D, float* y, float* pos_emb, int32_t* frame_num_out,
                               m3_stream stream);

/* small plugins */
int m3_att_masked_softmax(const float* scores, const int32_t* len, int B, int H, int T1, int T2, float scale,
                          float* out, m3_stream stream);                 /* att_masked_softmax_plugin.cpp:84-108 */
int m3_masked_fill(const float* x, const int32_t* len, int B, int C, int T, float fill, float* y,
                   m3_stream stream);                                     /* masked_fill_plugin.cpp:87-108 */
int m3_glu(const float* x, int outer, int C, int inner, float* y, m3_stream stream); /* glu_plugin.cpp:90-134 */
int m3_mask_conv2d_sample(const int32_t* len_in, int B, int left_padding, int stride, int32_t* len_out,
                          m3_stream stream);                              /* mask_conv2d_sample_plugin.cpp:70-80 */
int m3_scale(const float* x, float scale, float* y, size_t n, m3_stream stream); /* rel_positional_encoding_kernel.cu:62-69 */

/* TensorRT-native element-wise / shuffle / concat / matmul layers used through network_helper */
int m3_unary(const float* x, float* y, size_t n, int act, m3_stream stream);
int m3_binary(const float* a, const float* b, float* y, const int64_t* shape, const int64_t* strides_a,
              const int64_t* strides_b, int ndim, int op, m3_stream stream);
int m3_permute(const float* x, float* y, const int64_t* out_shape, const int64_t* in_strides, int ndim,
               m3_stream stream);
int m3_concat_last(const float* a, int da, const float* b, int db, float* y, size_t rows, m3_stream stream);
int m3_softmax(const float* x, float* y, size_t rows, int n, m3_stream stream);
int m3_batched_matmul(const float* a, const float* b, float* c, int batch, int M, int N, int K,
                      int64_t stride_a, int64_t stride_b, int transpose_b, m3_stream stream);
/* zero padding of the last two dims (TensorRT IPaddingLayer; network.add_padding of the causal conv module,
 * convolution.py:118-123): x (outer, H, W) -> y (outer, H + pre_h + post_h, W + pre_w + post_w) */
int m3_pad2d(const float* x, size_t outer, int H, int W, int pre_h, int post_h, int pre_w, int post_w, float* y,
             m3_stream stream);
/* (B,C,T) -> (B,C,T + 2 pad - K + 1), as nn.Conv1d(groups = C, padding = pad) */
int m3_depthwise_conv1d(const float* x, const float* w, const float* bias, int B, int C, int T, int K, int pad,
                        float* y, m3_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-encoder engine.  Replaces the TensorRT engine built by builder.py:36-98 and run by
 * infer.py:38-103 (IExecutionContext::execute_v2): feat (B,T,idim) f32 + feat_len (1,B) i32 -> logits (B,T',V).
 * The weight blob is the "plan" payload produced by the builder (packed fp32, offsets in `table`).
 * ---------------------------------------------------------------------------------------------- */
typedef struct m3_engine m3_engine;

typedef struct m3_engine_config {
  int32_t input_dim, output_dim;
  int32_t attention_dim, attention_heads, num_blocks;
  int32_t embed_dim, embed_heads, embed_linear_units, embed_blocks;
  int32_t num_experts, hidden_units;
  int32_t cnn_module_kernel;
  int32_t cnn_layer_norm;        /* 1 = LayerNorm in the conv module, 0 = (folded) batch norm */
  int32_t embed_cnn_layer_norm;
  int32_t router_with_bias, keep_expert_output;
  int32_t ep_world_size, ep_rank; /* expert parallel: this rank owns experts [rank*E_loc, (rank+1)*E_loc) */
  int32_t fold_pos_proj;         /* 1 = linear_pos(pos_emb) computed once per T' at shape set-up */
  int32_t debug_taps;            /* 1 = keep every block's output (the reference's DumpTensor taps) */
  int32_t log_softmax_out;       /* 1 = output log_softmax(logits) (+ "output_bias" weight entry if present, e.g. -log prior) */
  int32_t fuse_route;            /* 1 = router + SoftmaxTopK + ScatterMapping in one launch per layer (S <= 256, 1 rank);
                                  * 2 = split route: embed half of all routers in one GEMM, x half with folded LayerNorm,
                                  *     norm_ff applied by the expert kernel (1 rank, fp32) */
  int32_t shape_cache;           /* bound (shape, buffers) sets kept besides the current one, each with its stage list and
                                  * captured hipGraph (LRU): 0 = default 7, -1 = none.  Nothing the engine needs between two
                                  * forwards lives in the caller's workspace (the folded positional projection is
                                  * engine-owned device memory), so one workspace may serve every shape. */
  int32_t bf16_activations;      /* 16-bit modes, long batches: activations that only feed GEMMs are kept as bf16 and a bf16
                                  * copy of the residual stream is maintained (0 = automatic, -1 = never; the expert-parallel
                                  * host driver needs -1 because it replaces the stage that writes the copy) */
  int32_t weight_dtype;          /* M3_F32 / M3_BF16: storage of the GEMM weights (linear / point-wise conv /
                                  * conv2 / expert w_1, w_2 / pos_all); router, norms, biases, conv1, depthwise stay fp32.
                                  * M3_FP8: expert w_1 / w_2 in e4m3 with per-row scales ("...w_1.scale", "...w_2.scale"),
                                  * the other GEMM weights bf16 */
  int32_t packed_rows;           /* ragged batches (B > 1): run every row-wise kernel of the blocks on the sum of the valid
                                  * frames instead of B x T' padded rows (0 = automatic: on for B > 1 on one rank without
                                  * debug taps / fused routing, -1 = never, 1 = also for B = 1).  The interface does not
                                  * change: logits come back as (B, T', V), zeros past each utterance's last frame; the
                                  * "x" / "xn" / "embed" buffers then hold packed rows ("row0" = first row per utterance) */
  int32_t fp8_activations;       /* weight_dtype M3_FP8 only: 1 = fp8 ARITHMETIC in the grouped expert FFN where the fused
                                  * fp8 kernel applies (long batches): rows quantised per row, H with the static per-layer
                                  * scale "blocks.N.feed_forward.experts.h_scale" of the plan (calibrated); elsewhere the
                                  * weight-only form runs */
  int32_t ep_stages;             /* 1 = build the expert-parallel stage list ("blocks.N.moe_ep.send / .expert / .combine")
                                  * even with ep_world_size <= 1: a one-rank rehearsal of the exchange, the two all-to-alls
                                  * being device copies ("moe_ep.exchange1 / 2" stages).  ep_world_size > 1 always builds it */
  int32_t fork_embed;            /* the embed encoder is independent of the main encoder until blocks.0's router reads the
                                  * embedding (conformer_fmoe_..._hier.py:206-215): in the captured hipGraph it runs as a second
                                  * branch beside the main subsampler and block 0's macaron FFN / attention / conv module, on
                                  * its own scratch buffers.  1 = on, 0 / -1 = off (default: with several execution contexts the
                                  * extra branch costs more queue concurrency than it saves latency).  Stage-wise
                                  * runs (m3_engine_run) stay one chain; results are identical either way */
  int32_t static_chunk_size;     /* > 0: static chunk mask in every attention (add_optional_chunk_mask, utils/mask.py:127-134;
                                  * subsequent_chunk_mask :42-75): query frame i sees keys [max((i / c - left) c, 0),
                                  * min((i / c + 1) c, T')) and < len.  0 = full context */
  int32_t num_left_chunks;       /* chunks to the left a query sees with static_chunk_size > 0; < 0: all */
  int32_t causal;                /* 1 = causal ConvolutionModule in the main encoder (convolution.py:43-49,118-123: lorder = K - 1
                                  * frames padded on the left in front of pointwise_conv1, depthwise conv without padding); needs
                                  * the plan entry "blocks.N.conv_module.left_fill" [D] = GLU(pointwise_conv1.bias) */
  int32_t embed_causal;          /* the same for the embed encoder (conformer_embed_domain_acc.py:51,127; embed_conf['causal']) */
} m3_engine_config;

typedef struct m3_weight_entry {
  const char* name; /* packed tensor name, see m3asr/plan.py */
  const void* data; /* device pointer */
  int64_t numel;
  int32_t dtype;    /* enum m3_dtype; checked against what the engine expects for that tensor */
} m3_weight_entry;

m3_engine* m3_engine_create(const m3_engine_config* config, const m3_weight_entry* table, int n_entries);
void m3_engine_destroy(m3_engine* engine);
/* T' for T input frames (MaskConv2dSample twice, mask_conv2d_sample_kernel.cu:34-35) */
int m3_engine_output_frames(int T);
size_t m3_engine_workspace_size(const m3_engine* engine, int B, int T);
/* Enqueue one encoder forward.  All pointers device, caller-owned.  use_graph=1 replays a hipGraph
 * captured for this (B, T, pointers) on first use. */
int m3_engine_forward(m3_engine* engine, const float* feat, const int32_t* feat_len, int B, int T, float* logits,
                      void* workspace, size_t workspace_bytes, int use_graph, m3_stream stream);
/* Staged execution (used by the expert-parallel host driver, m3asr/ep.py, and by per-stage timing):
 * m3_engine_prepare binds shape + caller-owned buffers and builds the ordered kernel-stage list;
 * stages [first, last) are then enqueued with m3_engine_run.  Stage names are
 * "<prefix>.<op>", e.g. "embed.blocks.0.ffn_macaron.w1", "blocks.3.moe_router", "blocks.3.moe_local.expert",
 * "logits"; one kernel per stage.  In expert-parallel mode the host replaces the "blocks.N.moe_local.*" stages by
 * local index -> RCCL all-to-all -> m3_moe_expert_ffn on the received rows -> all-to-all back -> combine. */
int m3_engine_prepare(m3_engine* engine, const float* feat, const int32_t* feat_len, int B, int T, float* logits,
                      void* workspace, size_t workspace_bytes);

/* ---- chunk-by-chunk (streaming) execution -------------------------------------------------------------------------------
 * Decoding-chunk semantics of trainer_3m_fix/model/encoder.py:100-140 (decoding_chunk_size, num_decoding_left_chunks) with
 * the caches the reference's streaming plugins carry (cat_split_cache_kernel.cu:30-107, att_stream_softmax_kernel.cu:136-191,
 * rel_positional_encoding_kernel.cu:108-123).  The engine must have static_chunk_size = c > 0 and causal = embed_causal = 1.
 * B utterances are decoded side by side, one chunk of c output frames per call: the caller hands over the window of
 * m3_engine_chunk_input_frames() = 4c + 3 feature frames that starts at input frame 4 c n (windows overlap by 3 frames, the
 * context of the two stride-2 convs) and, per utterance, how many of its frames are real (0 for an utterance that has ended
 * or has fewer than 7 frames left).  logits (B, c, V): rows past an utterance's valid frames are undefined.
 * State (caller-owned device memory, m3_engine_stream_state_size bytes): a device-side chunk counter, per block the K | V
 * history [B][history_frames][2D] (a ring when num_left_chunks >= 0: history_frames >= (num_left_chunks + 1) c; otherwise it
 * must hold the whole stream) and the depthwise conv's K-1 frame cache.  max_frames bounds the stream's length in output
 * frames (positions; < rows of "pe").  m3_engine_stream_reset starts a new set of B streams.  chunk_index is the host's
 * count of chunks already decoded (validated against max_frames; the kernels use the device-side counter, so the call is a
 * hipGraph replay from the second chunk on).  Contract: chunk n's logits equal rows [n c, (n+1) c) of m3_engine_forward on
 * the whole utterances up to fp32 rounding of the GEMMs. */
typedef struct m3_stream_desc {
  int32_t B;
  int32_t history_frames;
  int32_t max_frames;
} m3_stream_desc;
int m3_engine_chunk_input_frames(const m3_engine* engine);
size_t m3_engine_stream_state_size(const m3_engine* engine, const m3_stream_desc* desc);
int m3_engine_stream_reset(m3_engine* engine, const m3_stream_desc* desc, void* state, size_t state_bytes, m3_stream stream);
int m3_engine_forward_chunk(m3_engine* engine, const m3_stream_desc* desc, void* state, size_t state_bytes,
                            const float* feat_chunk, const int32_t* chunk_feat_len, float* logits, void* workspace,
                            size_t workspace_bytes, int chunk_index, int use_graph, m3_stream stream);
/* Expert parallel: rows per wire chunk for the bindings made from now on (what the ranks agreed on: the largest row
 * count B*T' of any rank, so that a rank may send all of its rows to one peer; 0 = this rank's own row count).  The wire
 * buffers "ep.wire_a" / "ep.wire_b" ([world][1 + rows_per_chunk][D] fp32 each, inside the workspace) are what the host
 * hands to the all-to-all: after "blocks.N.moe_ep.send" exchange wire_a -> wire_b, after "blocks.N.moe_ep.expert" again
 * wire_a -> wire_b, then "blocks.N.moe_ep.combine".  Semantics: trainer_3m_fix/fmoe/functions.py:13-86,175-199. */
int m3_engine_set_ep_capacity(m3_engine* engine, int rows_per_chunk);
int m3_engine_num_stages(const m3_engine* engine);
int m3_engine_num_captures(const m3_engine* engine); /* hipGraphs captured so far (a cache hit replays, it does not capture) */
const char* m3_engine_stage_name(const m3_engine* engine, int index);
int m3_engine_run(m3_engine* engine, int first_stage, int last_stage, m3_stream stream);
/* Device address (inside the bound workspace) and size of a named intermediate of the prepared shape:
 * "x" (residual stream, S*D), "xn" (LayerNorm'd MoE input), "embed", "lens" (B int32),
 * "blocks.N.gate_idx" / "gate_value" / "mapping" / "acc_histogram", "blocks.N.out" (debug_taps only). */
int m3_engine_buffer(const m3_engine* engine, const char* name, void** ptr, size_t* bytes);
int m3_engine_num_kernels(const m3_engine* engine);
/* What a stage of the prepared shape launches, for measurement (bench.py prices each stage against the roofline):
 * `kernel` = the device kernel the stage's dispatcher picks for this shape, `launches` = kernel launches of the stage,
 * `alg_bytes` / `flops` = algorithmic HBM bytes (operands read once, results written once) and FLOPs (2 per MAC) of one
 * run with every padded row live; `per_row` = 1 when both scale with the live rows of a packed ragged batch;
 * alg_bytes < 0: data-dependent (the grouped expert FFN: touched experts x weight bytes, priced by the caller from
 * the routing taps). */
typedef struct m3_stage_info {
  const char* kernel;
  int32_t launches;
  int32_t per_row;
  double alg_bytes;
  double flops;
} m3_stage_info;
int m3_engine_stage_info(const m3_engine* engine, int index, m3_stage_info* info);

#ifdef __cplusplus
}
#endif
#endif /* M3ASR_H_ */
