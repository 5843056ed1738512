// Whole-encoder engine: the native runtime that stands where the TensorRT engine stood.
//
// Reference: builder.py:36-98 builds a TensorRT plan from model.encoder(network_helper, feat, feat_len)
// and infer.py:38-103 runs it with execute_v2.  The network being executed is
//   Net.forward            trainer_3m_fix/model/conformer_fmoe_localComm_catEmbed_domain_acc_hier.py:198-234
//   embed encoder          trainer_3m_fix/model/conformer_embed_domain_acc.py:149-181
//   FmoeConformerLayer     trainer_3m_fix/layer/fmoe_transformer.py:72-170
//   ConformerEncoderLayer  trainer_3m_fix/layer/transformer.py:179-275
// Here the same network is an ordered list of fused kernel stages over caller-owned buffers, captured
// once per (shape, buffers) into a hipGraph and replayed.  The residual stream x (S x D) is updated in
// place by GEMM / combine epilogues; LayerNorms ride in GEMM prologues except where their output is a
// tensor of its own (MoE input, block output).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/m3asr.h"
#include "common.h"
#include "kernels.h"

namespace m3 {
struct MoeWorkspace {
  int32_t* mapping; int32_t* acc; int32_t* pos; float* slab; size_t bytes;
};
MoeWorkspace carve_moe_workspace(void* base, int S, int E, int D, int F);
}  // namespace m3

using namespace m3;

namespace {

struct Norm { const float* g = nullptr; const float* b = nullptr; };
struct Lin { const float* w = nullptr; const float* b = nullptr; const float* wsum = nullptr; const float* wbeta = nullptr; };

struct BlockW {
  Norm n_ffm, n_mha, n_conv, n_ff, n_final, n_cnn;
  Lin mac1, mac2, qkv, pos, out, pw1, pw2, ff1, ff2;
  const float* pos_u = nullptr; const float* pos_v = nullptr;
  const float* dw_w = nullptr; const float* dw_b = nullptr;
  const float* left_fill = nullptr;   // causal conv module: GLU(pointwise_conv1.bias), what a zero-padded frame is at the depthwise conv's input
  Lin router;                       // w: [E_total][D + De]  (unfused route path)
  Lin router_x;                     // w: [E][D] x-half with norm_ff folded (+ wsum, bias)  (fused route path)
  const float *ew1 = nullptr, *eb1 = nullptr, *ew2 = nullptr, *eb2 = nullptr;
  const float *es1 = nullptr, *es2 = nullptr;   // fp8 experts: per-row scales [E][F], [E][D]
  float h_scale = 0.f;                          // fp8 arithmetic: static scale of the hidden activations (0 = weight-only)
};

struct SubW { const float* c1w; const float* c1b; const float* c2w; const float* c2b; Lin out; };

struct Stage {
  std::string name;
  std::function<int(hipStream_t)> run;
  m3_stage_info info;   // kernel label + algorithmic bytes / FLOPs of the stage (m3_engine_stage_info)
  // what horizontal fusion pairs up (fuse_independent_pairs): 1 = a plain launch_gemm_f32 stage (`gemm`), 2 = the conv module's
  // depthwise conv + LayerNorm + SiLU (`dw`)
  int fuse_kind = 0;
  GemmParams gemm;
  DwArgs dw;
  AttArgs att;          // 3 = the fp32 rel-pos attention core
};

m3_stage_info stage_info(const char* kernel, int launches, double bytes, double flops, bool per_row = true) {
  m3_stage_info i;
  i.kernel = kernel; i.launches = launches; i.per_row = per_row ? 1 : 0; i.alg_bytes = bytes; i.flops = flops;
  return i;
}

struct Buf { void* ptr; size_t bytes; };

}  // namespace

struct m3_engine {
  m3_engine_config cfg;
  std::unordered_map<std::string, m3_weight_entry> table;
  std::vector<std::string> names;  // keeps c_str storage alive
  SubW sub_e, sub_m;
  std::vector<BlockW> eblocks, mblocks;
  Norm e_after, m_after;
  Lin out_linear;
  const float* pe = nullptr;
  int64_t pe_rows = 0;
  const float* pos_all = nullptr;   // [(embed_blocks + num_blocks) * D][D]: every block's linear_pos weight
  const float* cmvn_mean = nullptr; const float* cmvn_istd = nullptr;   // optional global CMVN (fused into conv1)
  const float* output_bias = nullptr;                                     // optional [V] added to the output (-log prior)
  const float* router_e_all = nullptr;   // [num_blocks * E][De]: embed half of every layer's router (fused route path)

  // state of the bound shape (buffers + stage list + captured graph); up to cfg.shape_cache more are parked, so a server
  // that alternates between a few (B, T) buckets with static I/O buffers replays graphs instead of re-capturing them
  struct Bound {
    int B = 0, T = 0, Tp = 0, S = 0;
    const float* feat = nullptr; const int32_t* feat_len = nullptr; float* logits = nullptr;
    void* ws = nullptr; size_t ws_bytes = 0;
    float* splitk_ws = nullptr; size_t splitk_bytes = 0;   // partial tiles of the split-K front-end GEMMs (inside ws)
    bool a16 = false;   // activations that only feed GEMMs are kept as bf16 (h1, ctx, dw, c1, c2) + a bf16 copy of x
    bool dma = false;   // a16 and the block GEMMs run on the LDS-DMA kernel: every kernel that writes xb also leaves its row statistics
    bool packed = false;   // ragged batch: the blocks run on the packed valid rows (cfg.packed_rows)
    bool xn_skipped = false;   // the router kernel was told not to write the fp32 MoE input rows (M3_ROUTER_SKIP_XN=1)
    // (build-time scratch of the stage list) the next conv1 stage also forms the subsampled lengths; a second LayerNorm
    // for the next norm_final stage
    bool lens_in_conv1 = false;
    const float *tail_ln_g = nullptr, *tail_ln_b = nullptr; float tail_ln_eps = 0.f; float* tail_ln_out = nullptr;
    int ep_cap = 0;        // expert parallel: rows per wire chunk this binding was built for (m3_engine_set_ep_capacity)
    // chunk-by-chunk (streaming) binding: T = 4 c + 3 input frames -> the c frames of one chunk; attention reads / extends the
    // K / V history and the causal depthwise conv its K-1 frame cache, both in the caller-owned state (m3_engine_forward_chunk)
    void* sstate = nullptr; int s_hist = 0, s_maxf = 0;
    // fork_embed: stages [fork_first, fork_mid) = the embed encoder (side branch of the captured graph), [fork_mid, join_at) =
    // what the main encoder does before it needs the embedding; -1 = one linear chain
    int fork_first = -1, fork_mid = -1, join_at = -1;
    // fold_pos_proj: linear_pos(pe[:T']) of every block, computed once per T'.  ENGINE-owned device memory (shared by all
    // bindings of the same T', freed with the last of them): a caller that reuses one workspace for several shapes, as a
    // TensorRT execution context does, must not be able to overwrite it between two forwards of a revived binding
    std::shared_ptr<float> pfold;
    std::vector<Stage> stages;
    std::unordered_map<std::string, Buf> buffers;
    int n_kernels = 0;
    hipGraphExec_t graph_exec = nullptr;
    bool graph_valid = false;
    uint64_t last_use = 0;
    bool matches(int b, int t, const float* f, const int32_t* fl, const float* lg, const void* w, size_t wb, int cap,
                 const void* st = nullptr, int hist = 0, int maxf = 0) const {
      return !stages.empty() && B == b && T == t && feat == f && feat_len == fl && logits == lg && ws == w && ws_bytes == wb &&
             ep_cap == cap && sstate == st && s_hist == hist && s_maxf == maxf;
    }
  };
  Bound cur;
  std::vector<Bound> parked;
  uint64_t use_clock = 0;
  int n_captures = 0;                                      // graphs captured so far (observability / tests)
  int ep_capacity = 0;                                     // rows per wire chunk agreed by the ranks for the NEXT bindings (0 = own rows)
  hipStream_t side = nullptr;                              // second capture stream: the embed branch of forked graphs
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::unordered_map<int, std::weak_ptr<float>> pfold_by_tp;   // T' -> folded positional projection still in use
};

namespace {

// rows up to which SoftmaxTopK + ScatterMapping run as ONE single-work-group launch (beyond: row-parallel top-1 + index kernel).
// M3_GATE_INDEX_MAX_ROWS overrides (read once).  A/B at configs[2] (1984 padded rows): 2048 -> 315 instead of 333 launches but
// 3.856 vs 3.794 ms per forward (one work-group walks 1984 x 32 logits): the default stays 512.
int gate_index_max_rows() {
  static const int v = [] { const char* e = getenv("M3_GATE_INDEX_MAX_ROWS"); return e ? atoi(e) : 512; }();
  return v;
}

// M3_SELF_ROUTE=0 puts the single-work-group SoftmaxTopK + ScatterMapping launch back in front of the B = 1 expert launch (read once)
bool self_route_enabled() {
  static const bool v = [] { const char* e = getenv("M3_SELF_ROUTE"); return !(e && atoi(e) == 0); }();
  return v;
}

// dtype: what the engine will read the tensor as (GEMM weights follow cfg.weight_dtype, everything else is fp32)
bool lookup(const m3_engine* e, const std::string& name, int64_t numel, const float** out, int dtype = M3_F32) {
  auto it = e->table.find(name);
  if (it == e->table.end()) {
    set_error("engine: weight '%s' missing from the plan", name.c_str());
    return false;
  }
  if (numel >= 0 && it->second.numel != numel) {
    set_error("engine: weight '%s' has %lld elements, expected %lld", name.c_str(), (long long)it->second.numel,
              (long long)numel);
    return false;
  }
  if (it->second.dtype != dtype) {
    set_error("engine: weight '%s' has dtype %d, the engine (weight_dtype=%d) expects %d", name.c_str(),
              (int)it->second.dtype, (int)e->cfg.weight_dtype, dtype);
    return false;
  }
  *out = (const float*)it->second.data;
  return true;
}

#define GET(dst, name, numel) \
  do { if (!lookup(e, (name), (numel), &(dst))) return false; } while (0)
// dense GEMM weights: fp32, or bf16 in both 16-bit and fp8 modes; expert weights follow weight_dtype itself
#define GETW(dst, name, numel) \
  do { if (!lookup(e, (name), (numel), &(dst), e->cfg.weight_dtype == M3_F32 ? M3_F32 : M3_BF16)) return false; } while (0)
#define GETE(dst, name, numel) \
  do { if (!lookup(e, (name), (numel), &(dst), e->cfg.weight_dtype)) return false; } while (0)

bool load_norm(const m3_engine* e, const std::string& p, int d, Norm* n) {
  GET(n->g, p + "weight", d);
  GET(n->b, p + "bias", d);
  return true;
}
bool load_lin(const m3_engine* e, const std::string& p, int64_t n_out, int64_t n_in, bool bias, Lin* l) {
  GETW(l->w, p + "weight", n_out * n_in);
  if (bias) GET(l->b, p + "bias", n_out);
  return true;
}
// Linear with a LayerNorm folded in (plan.py fold_layernorm): weight = W*gamma, bias = b + W.beta, wsum = rowsum(W*gamma)
bool load_lin_ln(const m3_engine* e, const std::string& p, int64_t n_out, int64_t n_in, bool wbeta, Lin* l,
                 bool fp32_only = false) {
  if (fp32_only) {
    GET(l->w, p + "ln.weight", n_out * n_in);
  } else {
    GETW(l->w, p + "ln.weight", n_out * n_in);
  }
  GET(l->b, p + "ln.bias", n_out);
  GET(l->wsum, p + "ln.wsum", n_out);
  if (wbeta) GET(l->wbeta, p + "ln.wbeta", n_out);
  return true;
}

bool load_block(const m3_engine* e, const std::string& p, int D, int F, int K, bool cnn_ln, bool moe, int De, BlockW* b, bool causal) {
  const m3_engine_config& c = e->cfg;
  // norm_ff_macaron / norm_mha / norm_conv (and norm_ff of dense blocks) have no tensor of their own: folded into weights
  if (!load_norm(e, p + "norm_final.", D, &b->n_final)) return false;
  if (moe && !load_norm(e, p + "norm_ff.", D, &b->n_ff)) return false;
  if (!load_lin_ln(e, p + "feed_forward_macaron.w_1.", F, D, false, &b->mac1) ||
      !load_lin(e, p + "feed_forward_macaron.w_2.", D, F, true, &b->mac2) ||
      !load_lin_ln(e, p + "self_attn.qkv.", 3 * D, D, false, &b->qkv) ||
      !load_lin(e, p + "self_attn.linear_out.", D, D, true, &b->out) ||
      !load_lin_ln(e, p + "conv_module.pointwise_conv1.", 2 * D, D, true, &b->pw1) ||
      !load_lin(e, p + "conv_module.pointwise_conv2.", D, D, true, &b->pw2))
    return false;
  GET(b->pos_u, p + "self_attn.pos_bias_u", D);
  GET(b->pos_v, p + "self_attn.pos_bias_v", D);
  GET(b->dw_w, p + "conv_module.depthwise_conv.weight_kc", (int64_t)K * D);
  GET(b->dw_b, p + "conv_module.depthwise_conv.bias", D);
  if (cnn_ln && !load_norm(e, p + "conv_module.norm.", D, &b->n_cnn)) return false;
  if (causal) GET(b->left_fill, p + "conv_module.left_fill", D);
  if (!moe) {
    if (!load_lin_ln(e, p + "feed_forward.w_1.", F, D, false, &b->ff1) ||
        !load_lin(e, p + "feed_forward.w_2.", D, F, true, &b->ff2))
      return false;
  } else {
    const int world = c.ep_world_size > 0 ? c.ep_world_size : 1;
    const int64_t Etot = (int64_t)c.num_experts * world;
    GET(b->router.w, p + "feed_forward.router_weights_t", Etot * (D + De));
    if (c.router_with_bias) GET(b->router.b, p + "feed_forward.router_bias", Etot);
    if (c.fuse_route && !load_lin_ln(e, p + "feed_forward.router_x.", Etot, D, false, &b->router_x, true)) return false;
    const int64_t E = c.num_experts;
    GETE(b->ew1, p + "feed_forward.experts.w_1.weight", E * F * D);
    if (c.weight_dtype == M3_FP8) {
      GET(b->es1, p + "feed_forward.experts.w_1.scale", E * F);
      GET(b->es2, p + "feed_forward.experts.w_2.scale", E * D);
      if (c.fp8_activations) {          // the calibrated scale of H: one fp32 number per layer, read once at set-up
        const float* hs = nullptr;
        GET(hs, p + "feed_forward.experts.h_scale", 1);
        if (hipMemcpy(&b->h_scale, hs, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess || !(b->h_scale > 0.f)) {
          set_error("engine: '%sfeed_forward.experts.h_scale' must hold one positive number", p.c_str());
          return false;
        }
      }
    }
    GET(b->eb1, p + "feed_forward.experts.w_1.bias", E * F);
    GETE(b->ew2, p + "feed_forward.experts.w_2.weight_sliced", E * D * F);
    GET(b->eb2, p + "feed_forward.experts.w_2.bias", E * D);
  }
  return true;
}

bool load_sub(const m3_engine* e, const std::string& p, int D, int idim, SubW* s) {
  const int F2 = ((idim - 1) / 2 - 1) / 2;
  GET(s->c1w, p + "conv.0.weight_9c", 9 * (int64_t)D);
  GET(s->c1b, p + "conv.0.bias", D);
  GETW(s->c2w, p + "conv.2.weight_ohwi", 9 * (int64_t)D * D);
  GET(s->c2b, p + "conv.2.bias", D);
  return load_lin(e, p + "out.0.", D, (int64_t)D * F2, true, &s->out);
}
#undef GET
#undef GETW
#undef GETE

inline int sub_len(int t) { return ((t - 3) / 2 + 1 - 3) / 2 + 1; }

struct Carver {
  char* base; size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <typename T> T* take(size_t n) {
    T* p = (T*)(base ? base + off : nullptr);
    off += align_up(n * sizeof(T), 256);
    return p;
  }
};

// Buffer plan of one bound shape (identical code computes the size and the addresses).
struct Plan {
  int32_t* lens;
  float *c1, *c2, *x, *emb, *h1, *qkv, *pbuf, *ctx, *glu, *dw, *xn, *rl, *eall;
  void* xb;                             // bf16 copy of the residual stream (16-bit modes, long batches)
  int32_t* moe_fs;                      // fp8 plans: the F split the fused expert kernel chose on the device (slab count for the combine)
  unsigned char* xq; float* xq_scale;   // fp8 plans: the MoE input rows as e4m3 [S][D] + per-row scales (router kernel -> fused fp8 expert kernel)
  float* xstats;                        // [S][kXbStatParts][2]: row statistics of xb for the folded-LayerNorm GEMMs (gemm_bf16_dma.hip)
  int32_t* gate_idx; float* gate_val;   // [n_moe][S]
  void* moe_ws; size_t moe_ws_bytes;
  float* splitk; size_t splitk_bytes;   // split-K partials of conv2 / subsampling Linear (fp32 plans, short inputs)
  float* taps;                          // [n_blocks_total][S][D] when debug_taps
  float* pfold;                         // [n_blocks_total][Tp][D] when fold_pos_proj
  // packed ragged batches: row plan, padded staging of the subsamplers' output, packed logits
  int32_t *row0, *pad_of; float* xpad; void* xbpad; float* lpk;
  // expert parallel (ep_world_size > 1): send-side index over GLOBAL expert ids, the two wire buffers [world][1 + C][D]
  // and the receive-side gate; the MoE workspace is then sized for the world * (1 + C) rows a rank can receive
  int32_t *ep_acc, *ep_mapping, *ep_pos, *ep_map_send, *ep_gate_recv, *ep_overflow; float *wire_a, *wire_b; int ep_cap, ep_rows;
  // fork_embed: the embed encoder runs on its own graph branch beside the main subsampler and block 0 up to its router
  // (conformer_fmoe_..._hier.py:206-215: embed is needed first by blocks.0's router), so it owns a second set of scratch
  bool fork;
  bool hfuse;                           // embed chain and main prefix interleaved in one stream (separate scratch, as with fork)
  float *e_c1, *e_c2, *e_x, *e_h1, *e_qkv, *e_ctx, *e_glu, *e_dw, *e_xpad, *e_splitk, *e_xstats; void *e_xb, *e_xbpad;
  size_t bytes;
};

// Only on request.  Measured at configs[1] (profiles/r03_ab_headline.txt): one forward alone 2.39 -> 2.33 ms (the ~13 launches
// of the main encoder's start overlap the embed encoder), but four execution contexts x two branches are eight concurrently
// active queues, past the four the part runs truly concurrently: 207 k -> 69 k frames/s.
bool use_fork_embed(const m3_engine_config& c, int B, int S) { return c.fork_embed > 0 && !c.debug_taps; }
// the blocks run on packed rows: B > 1 (or forced), no per-block taps (they are read as (B, T', D)), staged route
bool use_packed_rows(const m3_engine_config& c, int B) {
  if (c.packed_rows < 0 || (c.packed_rows == 0 && B <= 1)) return false;
  return !c.debug_taps && c.fuse_route == 0 && B <= 1024;   // (expert-parallel ranks too: rows past the live count never travel)
}

// horizontal fusion of the embed encoder with the main encoder's independent prefix (fuse_independent_gemm_pairs): fp32 plans
// whose block GEMMs are 16-row-tile launches (S <= 128 rows).  Needs the embed chain's scratch apart from the main chain's.
bool use_hfuse(const m3_engine_config& c, int B, int S) {
  static const int on = [] { const char* ev = getenv("M3_HFUSE"); return ev ? atoi(ev) : 1; }();
  return on && !c.debug_taps && c.weight_dtype == M3_F32 && c.embed_blocks > 0 && c.num_blocks > 0 && S <= 128 && c.fork_embed <= 0 &&
         !use_packed_rows(c, B);
}

Plan make_plan(const m3_engine_config& c, void* base, int B, int T, int ep_capacity = 0) {
  Plan p;
  Carver cv(base);
  const int Tp = sub_len(T), S = B * Tp;
  const int T1 = (T - 3) / 2 + 1, F1 = (c.input_dim - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  const int D = c.attention_dim > c.embed_dim ? c.attention_dim : c.embed_dim;
  const int F = c.hidden_units > c.embed_linear_units ? c.hidden_units : c.embed_linear_units;
  const int world = c.ep_world_size > 0 ? c.ep_world_size : 1;
  const int Etot = c.num_experts * world;
  p.lens = cv.take<int32_t>(B);
  p.c1 = cv.take<float>((size_t)B * T1 * F1 * D);
  p.c2 = cv.take<float>((size_t)S * F2 * D);
  p.x = cv.take<float>((size_t)S * D);
  p.emb = cv.take<float>((size_t)S * D);
  p.xb = cv.take<uint16_t>((size_t)S * D);
  p.xstats = cv.take<float>((size_t)S * 2 * kXbStatParts);
  p.h1 = cv.take<float>((size_t)S * F);
  p.qkv = cv.take<float>((size_t)S * 3 * D);
  p.pbuf = cv.take<float>((size_t)Tp * D * (c.num_blocks + c.embed_blocks));
  p.ctx = cv.take<float>((size_t)S * D);
  p.glu = cv.take<float>((size_t)S * D);
  p.dw = cv.take<float>((size_t)S * D);
  p.xn = cv.take<float>((size_t)S * D);
  p.xq = nullptr; p.xq_scale = nullptr; p.moe_fs = nullptr;
  if (c.weight_dtype == M3_FP8) p.moe_fs = cv.take<int32_t>(64);
  if (c.weight_dtype == M3_FP8 && D == 512) {
    p.xq = cv.take<unsigned char>((size_t)S * D);
    p.xq_scale = cv.take<float>((size_t)S);
  }
  p.rl = cv.take<float>((size_t)S * Etot);
  p.eall = cv.take<float>((size_t)S * Etot * c.num_blocks);
  p.gate_idx = cv.take<int32_t>((size_t)c.num_blocks * S);
  p.gate_val = cv.take<float>((size_t)c.num_blocks * S);
  // rows per wire chunk: what the ranks agreed on (m3_engine_set_ep_capacity).  0 = this rank's own row count (no row can
  // be dropped); a smaller agreed capacity is a BOUNDED wire: a chunk that needs more rows reports it in "ep.overflow" and the
  // driver repeats the forward with a larger one (m3asr/ep.py)
  const bool ep = world > 1 || c.ep_stages > 0;
  p.ep_cap = ep ? (ep_capacity > 0 ? ep_capacity : S) : 0;
  p.ep_rows = ep ? world * (p.ep_cap + 1) : 0;
  p.moe_ws_bytes = carve_moe_workspace(nullptr, ep ? p.ep_rows : S, c.num_experts, c.attention_dim, c.hidden_units).bytes;
  p.moe_ws = cv.take<char>(p.moe_ws_bytes * (size_t)(c.debug_taps ? c.num_blocks : 1));
  p.ep_acc = p.ep_mapping = p.ep_pos = p.ep_map_send = p.ep_gate_recv = p.ep_overflow = nullptr; p.wire_a = p.wire_b = nullptr;
  if (ep) {
    p.ep_overflow = cv.take<int32_t>(64);
    p.ep_acc = cv.take<int32_t>((size_t)Etot + 1);
    p.ep_mapping = cv.take<int32_t>(S);
    p.ep_pos = cv.take<int32_t>(S);
    p.ep_map_send = cv.take<int32_t>(S);
    p.ep_gate_recv = cv.take<int32_t>(p.ep_rows);
    p.wire_a = cv.take<float>((size_t)p.ep_rows * c.attention_dim);
    p.wire_b = cv.take<float>((size_t)p.ep_rows * c.attention_dim);
  }
  {
    size_t n1 = 0, n2 = 0;
    if (c.weight_dtype == M3_F32) {
      GemmParams g;   // conv2 as implicit GEMM
      g.mode = GEMM_A_CONV3X3S2; g.lda = 4; g.conv_C = D; g.M = S * F2; g.N = D; g.K = 9 * D; g.ldy = D;
      gemm_f32_splitk_plan(g, &n1);
      GemmParams l;   // Linear(C*F2 -> D)
      l.lda = F2 * D; l.M = S; l.N = D; l.K = F2 * D; l.ldy = D;
      gemm_f32_splitk_plan(l, &n2);
    }
    p.splitk_bytes = n1 > n2 ? n1 : n2;
    p.splitk = p.splitk_bytes ? cv.take<float>(p.splitk_bytes / sizeof(float)) : nullptr;
  }
  p.taps = c.debug_taps ? cv.take<float>((size_t)(c.num_blocks + c.embed_blocks) * S * D) : nullptr;
  p.pfold = nullptr;
  p.row0 = p.pad_of = nullptr; p.xpad = p.lpk = nullptr; p.xbpad = nullptr;
  if (use_packed_rows(c, B)) {
    p.row0 = cv.take<int32_t>(B + 1);
    p.pad_of = cv.take<int32_t>(S);
    p.xpad = cv.take<float>((size_t)S * D);
    p.xbpad = cv.take<uint16_t>((size_t)S * D);
    p.lpk = cv.take<float>((size_t)S * c.output_dim);
  }
  p.fork = use_fork_embed(c, B, S);
  p.hfuse = !p.fork && use_hfuse(c, B, S);
  p.e_c1 = p.c1; p.e_c2 = p.c2; p.e_x = p.x; p.e_h1 = p.h1; p.e_qkv = p.qkv; p.e_ctx = p.ctx; p.e_glu = p.glu; p.e_dw = p.dw;
  p.e_xb = p.xb; p.e_xpad = p.xpad; p.e_xbpad = p.xbpad; p.e_splitk = p.splitk; p.e_xstats = p.xstats;
  if (p.fork || p.hfuse) {
    p.e_c1 = cv.take<float>((size_t)B * T1 * F1 * D);
    p.e_c2 = cv.take<float>((size_t)S * F2 * D);
    p.e_x = cv.take<float>((size_t)S * D);
    p.e_xb = cv.take<uint16_t>((size_t)S * D);
    p.e_xstats = cv.take<float>((size_t)S * 2 * kXbStatParts);
    p.e_h1 = cv.take<float>((size_t)S * F);
    p.e_qkv = cv.take<float>((size_t)S * 3 * D);
    p.e_ctx = cv.take<float>((size_t)S * D);
    p.e_glu = cv.take<float>((size_t)S * D);
    p.e_dw = cv.take<float>((size_t)S * D);
    if (p.splitk_bytes) p.e_splitk = cv.take<float>(p.splitk_bytes / sizeof(float));
    if (use_packed_rows(c, B)) {      // (not `if (p.xpad)`: the sizing pass carves from a null base)
      p.e_xpad = cv.take<float>((size_t)S * D);
      p.e_xbpad = cv.take<uint16_t>((size_t)S * D);
    }
  }
  p.bytes = cv.off;
  return p;
}

// Layout of the caller-owned streaming state (identical code computes the size and the addresses): the device-side chunk
// counter, then per block (embed blocks first) the K | V history [B][hist][2 D] and the depthwise conv's ping-pong cache
// [2][B][K-1][D] (post-GLU frames; the reference caches the module's INPUT and re-runs pointwise_conv1 + GLU on it,
// convolution.py:118-123 -- the same numbers, since both are per-frame operations).
struct StreamState {
  int32_t* step = nullptr;
  std::vector<float*> kv, conv;
  size_t bytes = 0;
};
StreamState carve_stream_state(const m3_engine_config& c, void* base, int B, int hist) {
  Carver cv(base);
  StreamState st;
  st.step = cv.take<int32_t>(64);
  const int nb = c.embed_blocks + c.num_blocks, D = c.attention_dim, K = c.cnn_module_kernel;
  for (int i = 0; i < nb; ++i) st.kv.push_back(cv.take<float>((size_t)B * hist * 2 * D));
  for (int i = 0; i < nb; ++i) st.conv.push_back(cv.take<float>((size_t)2 * B * (K - 1) * D));
  st.bytes = cv.off;
  return st;
}

// the same plan with the embed branch's scratch under the usual names (what the embed encoder's stages are built from)
Plan embed_view(const Plan& pl) {
  Plan q = pl;
  q.c1 = pl.e_c1; q.c2 = pl.e_c2; q.x = pl.e_x; q.h1 = pl.e_h1; q.qkv = pl.e_qkv; q.ctx = pl.e_ctx; q.glu = pl.e_glu; q.dw = pl.e_dw;
  q.xb = pl.e_xb; q.xpad = pl.e_xpad; q.xbpad = pl.e_xbpad; q.splitk = pl.e_splitk; q.xstats = pl.e_xstats;
  return q;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
static void add_stage(m3_engine* e, const std::string& name, int kernels, std::function<int(hipStream_t)> fn,
                      m3_stage_info info = stage_info("", 0, 0.0, 0.0)) {
  info.launches = kernels;
  e->cur.stages.push_back(Stage{name, std::move(fn), info});
  e->cur.n_kernels += kernels;
}

// algorithmic traffic of one GEMM: weights once, A rows once, result once (+ the residual it adds, + side outputs)
static m3_stage_info gemm_info(const GemmParams& p, bool splitk) {
  const bool glu = p.act == ACT_GLU;
  const double M = p.M, N = p.N, K = p.K, Nout = glu ? N / 2 : N;
  const double wsz = p.w_bf16 ? 2 : 4, asz = p.a_bf16 ? 2 : 4, ysz = p.y_bf16 ? 2 : 4;
  double a_bytes = M * K * asz;
  if (p.mode == GEMM_A_CONV3X3S2)   // implicit conv: the input tensor is read once, not 9 times
    a_bytes = (double)(p.M / (p.conv_T2 * p.conv_F2)) * p.conv_T1 * p.conv_F1 * p.conv_C * asz;
  double bytes = N * K * wsz + a_bytes + M * Nout * ysz;
  if (p.resid) bytes += M * Nout * 4;
  if (p.Yb) bytes += M * Nout * 2;
  if (p.ln_out) bytes += M * (K - p.K1) * 4;
  return stage_info(gemm_kernel_label(p, splitk), 1, bytes, 2.0 * M * N * K, p.mode != GEMM_A_CONV3X3S2);
}

// fp32_weights: the router GEMMs keep fp32 weights in every mode (a flipped top-1 is a discrete error)
static void add_gemm(m3_engine* e, const std::string& name, GemmParams p, bool fp32_weights = false) {
  p.w_bf16 = (!fp32_weights && e->cfg.weight_dtype != M3_F32) ? 1 : 0;
  size_t need = 0;
  if (gemm_f32_splitk_plan(p, &need) >= 2 && e->cur.splitk_ws != nullptr && need <= e->cur.splitk_bytes) {
    float* ws = e->cur.splitk_ws; const size_t wsb = e->cur.splitk_bytes;
    add_stage(e, name, 2, [p, ws, wsb](hipStream_t s) { return launch_gemm_f32_splitk(p, ws, wsb, s); }, gemm_info(p, true));
    return;
  }
  add_stage(e, name, 1, [p](hipStream_t s) { return launch_gemm_f32(p, s); }, gemm_info(p, false));
  e->cur.stages.back().fuse_kind = 1;
  e->cur.stages.back().gemm = p;
}

// Horizontal fusion (B = 1-sized fp32 plans): the embed encoder and the main encoder's prefix -- its subsampling and block 0 up to
// the router, the first stage that reads the embedding -- are two independent chains.  Where a stage of each is a skinny fp32
// GEMM of the same instantiation, the two become ONE launch (gemm_f32_dual_kernel): a launch saved is ~6 us of a forward that is
// 275 dependent launches.  Each chain keeps its own order; only the first embed stage (it also derives the output lengths every
// later kernel reads) is guaranteed to stay in front of every main-prefix stage.  Arithmetic per problem is unchanged: results
// are bit-identical.  M3_HFUSE=0: off.
static bool stages_fusable(const Stage& a, const Stage& b) {
  if (a.fuse_kind == 0 || a.fuse_kind != b.fuse_kind) return false;
  if (a.fuse_kind == 1) return gemm_f32_dual_fusable(a.gemm, b.gemm);
  if (a.fuse_kind == 2) return dwconv_dual_fusable(a.dw, b.dw);
  if (a.fuse_kind == 3) return relpos_attention_dual_fusable(a.att, b.att);
  return false;
}
static Stage fused_stage(const Stage& a, const Stage& b) {
  Stage d;
  d.name = a.name + "+" + b.name;
  const char* label = "";
  if (a.fuse_kind == 1) {
    const GemmParams pa = a.gemm, pb = b.gemm;
    d.run = [pa, pb](hipStream_t s) { return launch_gemm_f32_dual(pa, pb, s); };
    label = "gemm_f32_dual_kernel";
  } else if (a.fuse_kind == 2) {
    const DwArgs pa = a.dw, pb = b.dw;
    d.run = [pa, pb](hipStream_t s) { return launch_dwconv_ln_silu_dual(pa, pb, s); };
    label = "dwconv_ln_silu_dual_kernel";
  } else {
    const AttArgs pa = a.att, pb = b.att;
    d.run = [pa, pb](hipStream_t s) { return launch_relpos_attention_dual(pa, pb, s); };
    label = "relpos_attention_dual_kernel";
  }
  d.info = stage_info(label, 1, a.info.alg_bytes + b.info.alg_bytes, a.info.flops + b.info.flops);
  return d;
}
static void fuse_independent_pairs(m3_engine* e, int first, int mid, int join) {
  if (first < 0 || mid <= first + 1 || join <= mid) return;
  std::vector<Stage>& st = e->cur.stages;
  std::vector<Stage> out(st.begin(), st.begin() + first);
  std::vector<Stage> pending;                         // main-prefix stages waiting for the next pair they precede
  int i = first, saved = 0;
  for (int m = mid; m < join; ++m) {
    int partner = -1;
    if (st[m].fuse_kind)
      for (int k = std::max(i, first + 1); k < mid; ++k)
        if (stages_fusable(st[k], st[m])) { partner = k; break; }
    if (partner < 0) { pending.push_back(st[m]); continue; }
    for (; i < partner; ++i) out.push_back(st[i]);
    for (Stage& q : pending) out.push_back(q);
    pending.clear();
    out.push_back(fused_stage(st[partner], st[m]));
    i = partner + 1;
    ++saved;
  }
  for (; i < mid; ++i) out.push_back(st[i]);
  for (Stage& q : pending) out.push_back(q);
  for (int k = join; k < (int)st.size(); ++k) out.push_back(st[k]);
  st.swap(out);
  e->cur.n_kernels -= saved;
}

static void build_subsample(m3_engine* e, const std::string& pfx, const SubW& w, int D, const Plan& pl, float* xout) {
  const m3_engine_config& c = e->cfg;
  const int B = e->cur.B, T = e->cur.T;
  const int T1 = (T - 3) / 2 + 1, F1 = (c.input_dim - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1, T2 = (T1 - 3) / 2 + 1;
  const float* feat = e->cur.feat;
  float* c1 = pl.c1; float* c2 = pl.c2;
  const int idim = c.input_dim;
  const float* cm = e->cmvn_mean; const float* ci = e->cmvn_istd;
  const bool a16 = e->cur.a16;    // c1, c2 only feed GEMMs: kept as bf16; the Linear also writes the bf16 copy of x
  // (the forward's first conv1 also forms the subsampled lengths when no stage in front of it needs them: see "lens" below)
  const int32_t* flen = e->cur.lens_in_conv1 ? e->cur.feat_len : nullptr;
  int32_t* lens_out = pl.lens;
  e->cur.lens_in_conv1 = false;
  add_stage(e, pfx + "conv1", 1, [=](hipStream_t s) { return launch_conv1_relu(feat, w.c1w, w.c1b, cm, ci, B, T, idim, D, c1, s, 1, a16, flen, lens_out); },
            stage_info("conv1_relu_kernel", 1, (double)B * T * idim * 4 + (double)B * T1 * F1 * D * (a16 ? 2 : 4), 18.0 * B * T1 * F1 * D, false));
  GemmParams g;
  g.a_bf16 = a16; g.y_bf16 = a16;
  g.mode = GEMM_A_CONV3X3S2; g.A = c1; g.lda = 4;
  g.conv_T1 = T1; g.conv_F1 = F1; g.conv_T2 = T2; g.conv_F2 = F2; g.conv_C = D;
  g.W = w.c2w; g.bias = w.c2b; g.Y = c2; g.ldy = D; g.M = B * T2 * F2; g.N = D; g.K = 9 * D; g.act = ACT_RELU;
  if (e->cur.packed) g.conv_len = pl.lens;   // tiles of padded frames only: skipped (never gathered into the packed rows)
  add_gemm(e, pfx + "conv2", g);
  // Linear(C*F2 -> D) on the (f, c)-ordered flatten, with the positional-encoding scale sqrt(D)
  // (rel_positional_encoding_kernel.cu:62-69) folded into the epilogue.
  // packed ragged batch: the subsampler works on the padded (B, T) input; its rows are packed right after it
  const bool packed = e->cur.packed;
  GemmParams l;
  l.A = c2; l.lda = F2 * D; l.W = w.out.w; l.bias = w.out.b; l.Y = packed ? pl.xpad : xout; l.ldy = D;
  l.M = B * T2; l.N = D; l.K = F2 * D; l.alpha = sqrtf((float)D);
  l.a_bf16 = a16;
  if (a16) { l.Yb = packed ? pl.xbpad : pl.xb; l.ldyb = D; }
  add_gemm(e, pfx + "linear", l);
  if (packed) {
    const float* xpad = pl.xpad; const void* xbpad = pl.xbpad; void* xb = pl.xb; const int32_t* pad_of = pl.pad_of;
    const int S = B * T2;
    add_stage(e, pfx + "pack", a16 ? 2 : 1, [=](hipStream_t s) {
      if (int rc = launch_local_gather(xpad, pad_of, S, D * 4, xout, s)) return rc;
      return a16 ? launch_local_gather(xbpad, pad_of, S, D * 2, xb, s) : 0;
    }, stage_info("row_permute_kernel", 1, (double)S * D * (a16 ? 12 : 8) + 4.0 * S, 0.0));
  }
  if (e->cur.dma) {   // the first block's folded-LayerNorm GEMM takes the row statistics of xb from its producer: here a pass of its own
    const void* xbv = pl.xb; float* xs = pl.xstats;
    const int S = B * T2;
    add_stage(e, pfx + "row_stats", 1, [=](hipStream_t s) { return launch_row_stats_bf16(xbv, S, D, xs, s); },
              stage_info("row_stats_bf16_kernel", 1, (double)S * D * 2 + 32.0 * S, 3.0 * S * D));
  }
}

static void build_block(m3_engine* e, const std::string& pfx, const BlockW& w, int D, int F, int H, int K, bool cnn_ln,
                        bool moe, int layer, int tap_index, const Plan& pl, bool causal) {
  // every GEMM of a block is row-wise over the S rows of the batch: packed batches pass the live-row count
  auto add_gemm = [&](m3_engine* e_, const std::string& name, GemmParams g, bool fp32_weights = false) {
    if (e_->cur.packed) g.m_dev = pl.row0 + e_->cur.B;
    ::add_gemm(e_, name, g, fp32_weights);
  };
  const m3_engine_config& c = e->cfg;
  const int B = e->cur.B, Tp = e->cur.Tp, S = e->cur.S;
  float* x = pl.x;
  const int32_t* lens = pl.lens;
  const float eps = 1e-12f;  // all block LayerNorms (fmoe_transformer.py:54-65)

  // 16-bit modes, long batches: GEMM A operands come as bf16 -- the copy xb of the residual stream (written by every
  // kernel that writes x) and bf16 h1 / ctx / dw -- because these GEMMs are bound by the traffic of their fp32 A operand
  const bool a16 = e->cur.a16;
  void* xb = pl.xb;
  const bool dma = e->cur.dma;
  float* xstats = pl.xstats;
  auto from_xb = [&](GemmParams& g) {
    if (a16) { g.A = (const float*)xb; g.a_bf16 = 1; }
    if (dma) { g.ln_stats = xstats; g.ln_stat_parts = kXbStatParts; }
  };
  auto also_xb = [&](GemmParams& g) {
    if (a16) { g.Yb = xb; g.ldyb = D; }
    if (dma) g.Yb_stats = xstats;
  };
  // packed ragged batch: rows [0, P) are the valid frames of all utterances back to back, P = row0[B] on the device;
  // row-wise kernels skip the tiles beyond P, attention and the depthwise conv find their utterance through row0 / pad_of
  const bool packed = e->cur.packed;
  const int32_t* row0 = packed ? pl.row0 : nullptr;
  const int32_t* pad_of = packed ? pl.pad_of : nullptr;
  const int32_t* pdev = packed ? pl.row0 + B : nullptr;
  // "frame t of utterance b is padding" for the gate: padded layout (lens, T'), packed layout one run of P rows
  const int32_t* live_len = packed ? pdev : lens;
  const int live_rpb = packed ? S : Tp;
  {  // x += 0.5 * FFN_macaron(LN(x))
    GemmParams g;
    g.A = x; g.lda = D; g.W = w.mac1.w; g.bias = w.mac1.b; g.Y = pl.h1; g.ldy = F; g.M = S; g.N = F; g.K = D;
    g.ln_wsum = w.mac1.wsum; g.ln_eps = eps; g.act = ACT_SILU;   // norm_ff_macaron is folded into w_1 (plan.py)
    from_xb(g); g.y_bf16 = a16;
    add_gemm(e, pfx + "ffn_macaron.w1", g);
    GemmParams h;
    h.A = pl.h1; h.lda = F; h.W = w.mac2.w; h.bias = w.mac2.b; h.Y = x; h.ldy = D; h.M = S; h.N = D; h.K = F;
    h.alpha = 0.5f; h.resid = x; h.ldr = D;
    h.a_bf16 = a16; also_xb(h);
    add_gemm(e, pfx + "ffn_macaron.w2", h);
  }
  {  // x += MHA(LN(x))
    GemmParams g;
    g.A = x; g.lda = D; g.W = w.qkv.w; g.bias = w.qkv.b; g.Y = pl.qkv; g.ldy = 3 * D; g.M = S; g.N = 3 * D; g.K = D;
    g.ln_wsum = w.qkv.wsum; g.ln_eps = eps;                  // norm_mha is folded into the qkv weight
    from_xb(g);
    // 16-bit modes, long batches, T' <= 128: q | k | v are written as bf16 and the attention core runs on bf16 MFMAs with
    // K / P / V of a head staged once per (utterance, head) (attention.hip, second kernel)
    const bool att16 = a16 && relpos_attention_bf16_supports(Tp, D / H);
    g.y_bf16 = att16;
    add_gemm(e, pfx + "att.qkv", g);
    // p = linear_pos(pos_emb) of all blocks comes from ONE GEMM per forward ("pos_all" stage):
    // block i's slice is columns [i*D, (i+1)*D) of pbuf [T'][n_blocks*D]
    const int ldp = (c.num_blocks + c.embed_blocks) * D;
    const float* pmat = pl.pbuf + (size_t)tap_index * D;
    const float* qkv = pl.qkv; float* ctx = pl.ctx;
    const float* pu = w.pos_u; const float* pv = w.pos_v;
    const int dk = D / H;
    const float scale = 1.f / sqrtf((float)dk);
    const int chunk = c.static_chunk_size, left_chunks = c.num_left_chunks;   // static chunk mask (0 = full context)
    if (e->cur.sstate != nullptr) {   // chunk-by-chunk: keys = K / V history + this chunk, positions absolute, history appended in place
      const StreamState st = carve_stream_state(c, e->cur.sstate, B, e->cur.s_hist);
      float* hist = st.kv[tap_index]; const int cap = e->cur.s_hist; const int32_t* step = st.step;
      add_stage(e, pfx + "att.core", 1, [=](hipStream_t s) {
        return launch_relpos_attention_stream(qkv, 3 * D, hist, cap, pmat, ldp, pu, pv, lens, step, B, Tp, H, dk, scale, ctx, D, left_chunks, s);
      }, stage_info("relpos_attention_stream_kernel", 1, (double)S * D * 24 + (double)Tp * D * 4, 6.0 * Tp * D * S));
    } else
    {
    add_stage(e, pfx + "att.core", 1, [=](hipStream_t s) {
      if (att16) return launch_relpos_attention_bf16(qkv, 3 * D, pmat, ldp, pu, pv, lens, B, Tp, H, dk, scale, ctx, D, s, row0, chunk, left_chunks);
      return launch_relpos_attention(qkv, 3 * D, pmat, ldp, pu, pv, lens, B, Tp, H, dk, scale, ctx, D, s, a16, row0, chunk, left_chunks);
    }, stage_info(att16 ? "relpos_attention_bf16_kernel" : "relpos_attention_kernel", 1,
                  (double)S * D * (att16 ? 8 : (12 + (a16 ? 2 : 4))) + (double)Tp * D * 4, 6.0 * Tp * D * S));
    if (!att16) {      // (the fp32 core: a candidate for sharing a launch with the other encoder's, fuse_independent_pairs)
      AttArgs aa;
      aa.qkv = qkv; aa.ldq = 3 * D; aa.pmat = pmat; aa.ldp = ldp; aa.pos_u = pu; aa.pos_v = pv; aa.row_len = lens; aa.B = B; aa.T = Tp; aa.H = H;
      aa.dk = dk; aa.scale = scale; aa.out = ctx; aa.ldo = D; aa.out_bf16 = a16; aa.row0 = row0; aa.chunk = chunk; aa.left_chunks = left_chunks;
      e->cur.stages.back().fuse_kind = 3;
      e->cur.stages.back().att = aa;
    }
    }
    GemmParams o;
    o.A = pl.ctx; o.lda = D; o.W = w.out.w; o.bias = w.out.b; o.Y = x; o.ldy = D; o.M = S; o.N = D; o.K = D;
    o.resid = x; o.ldr = D;
    o.a_bf16 = a16; also_xb(o);
    add_gemm(e, pfx + "att.out", o);
  }
  {  // x += ConvModule(LN(x))
    GemmParams g;
    g.A = x; g.lda = D; g.W = w.pw1.w; g.bias = w.pw1.b; g.Y = pl.glu; g.ldy = D; g.M = S; g.N = 2 * D; g.K = D;
    g.ln_wsum = w.pw1.wsum; g.ln_wbeta = w.pw1.wbeta; g.ln_eps = eps; g.act = ACT_GLU;   // norm_conv folded into pw1
    // padded frames enter the conv module as zeros (convolution.py:101-104); packed: the rows >= P are "padding", and
    // row P thereby receives the constant a zeroed frame produces -- the depthwise conv reads it for taps len <= t < T'
    g.row_len = live_len; g.rows_per_batch = live_rpb; g.mask_in = 1;
    from_xb(g);
    add_gemm(e, pfx + "conv.pw1_glu", g);
    const float* glu = pl.glu; float* dw = pl.dw;
    const float* dww = w.dw_w; const float* dwb = w.dw_b;
    const float* ng = cnn_ln ? w.n_cnn.g : nullptr; const float* nb = cnn_ln ? w.n_cnn.b : nullptr;
    const float* lfill = causal ? w.left_fill : nullptr;   // causal conv module (convolution.py:43-49,118-123)
    if (e->cur.sstate != nullptr) {
      const StreamState st = carve_stream_state(c, e->cur.sstate, B, e->cur.s_hist);
      float* cpair = st.conv[tap_index]; const int32_t* step = st.step;
      add_stage(e, pfx + "conv.dw_ln_silu", 1, [=](hipStream_t s) {
        return launch_dwconv_ln_silu_stream(glu, dww, dwb, ng, nb, 1e-5f, B, Tp, D, K, dw, cpair, step, lens, s, a16);
      }, stage_info("dwconv_ln_silu_kernel", 1, (double)S * D * 8 + (double)K * D * 4 + 8.0 * B * (K - 1) * D, 2.0 * K * D * S));
    } else
    {
      DwArgs da;
      da.z = glu; da.w_kc = dww; da.bias = dwb; da.gamma = ng; da.beta = nb; da.eps = 1e-5f; da.B = B; da.T = Tp; da.D = D; da.K = K;
      da.out = dw; da.out_bf16 = a16; da.pad_of = pad_of; da.row0 = row0; da.row_len = lens; da.causal_left_fill = lfill;
      add_stage(e, pfx + "conv.dw_ln_silu", 1, [da](hipStream_t s) { return launch_dwconv_ln_silu_args(da, s); },
                stage_info("dwconv_ln_silu_kernel", 1, (double)S * D * (4 + (a16 ? 2 : 4)) + (double)K * D * 4, 2.0 * K * D * S));
      e->cur.stages.back().fuse_kind = 2;
      e->cur.stages.back().dw = da;
    }
    GemmParams h;
    h.A = pl.dw; h.lda = D; h.W = w.pw2.w; h.bias = w.pw2.b; h.Y = x; h.ldy = D; h.M = S; h.N = D; h.K = D;
    h.row_len = live_len; h.rows_per_batch = live_rpb; h.mask_out = 1; h.resid = x; h.ldr = D;
    h.a_bf16 = a16; also_xb(h);
    add_gemm(e, pfx + "conv.pw2", h);
  }
  if (!moe) {  // x = LN_final(x + 0.5 * FFN(LN(x)))
    GemmParams g;
    g.A = x; g.lda = D; g.W = w.ff1.w; g.bias = w.ff1.b; g.Y = pl.h1; g.ldy = F; g.M = S; g.N = F; g.K = D;
    g.ln_wsum = w.ff1.wsum; g.ln_eps = eps; g.act = ACT_SILU;   // norm_ff is folded into w_1
    from_xb(g); g.y_bf16 = a16;
    add_gemm(e, pfx + "ffn.w1", g);
    GemmParams h;
    h.A = pl.h1; h.lda = F; h.W = w.ff2.w; h.bias = w.ff2.b; h.Y = x; h.ldy = D; h.M = S; h.N = D; h.K = F;
    h.alpha = 0.5f; h.resid = x; h.ldr = D;
    h.a_bf16 = a16;                                   // (x is rewritten by norm_final below: no bf16 copy here)
    add_gemm(e, pfx + "ffn.w2", h);
    const float* fg = w.n_final.g; const float* fb = w.n_final.b;
    void* xbo = a16 ? xb : nullptr;
    float* xso = dma ? xstats : nullptr;
    // (the embed encoder's last block: after_norm rides in the same launch, conformer_embed_domain_acc.py:171-181)
    const float* g2 = e->cur.tail_ln_g; const float* b2 = e->cur.tail_ln_b; float* y2 = e->cur.tail_ln_out;
    const float eps2 = e->cur.tail_ln_eps;
    e->cur.tail_ln_g = nullptr;
    add_stage(e, pfx + "norm_final", 1, [=](hipStream_t s) { return launch_layernorm(x, fg, fb, eps, x, S, D, s, xbo, xso, g2, b2, eps2, y2); },
              stage_info("layernorm_kernel", 1, (double)S * D * (a16 ? 10 : 8) + (g2 ? 4.0 * S * D : 0.0), (g2 ? 16.0 : 8.0) * S * D));
  } else {  // x = LN_final(x + 0.5 * gate * Expert_g(LN(x)))     (positionwise_feed_forward.py:209-265)
    const int world = c.ep_world_size > 0 ? c.ep_world_size : 1;
    const int Etot = c.num_experts * world, E = c.num_experts, De = c.embed_dim;
    float* xn = pl.xn; float* rl = pl.rl;
    int32_t* gidx = pl.gate_idx + (size_t)layer * S;
    float* gval = pl.gate_val + (size_t)layer * S;
    const float* ng = w.n_ff.g; const float* nb = w.n_ff.b;
    void* mws = (char*)pl.moe_ws + (c.debug_taps ? (size_t)layer * pl.moe_ws_bytes : 0);
    const MoeWorkspace mw = carve_moe_workspace(mws, S, E, D, F);
    const float *ew1 = w.ew1, *eb1 = w.eb1, *ew2 = w.ew2, *eb2 = w.eb2;
    const float* fg = w.n_final.g; const float* fb = w.n_final.b;
    const float* gv = c.keep_expert_output ? nullptr : gval;
    // S <= 256 rows, all experts local, fp32, staged route: the expert launch routes for itself (M3_SELF_ROUTE=0: index launch as before)
    const bool self_route = self_route_enabled() && c.fuse_route == 0 && world == 1 && c.ep_stages <= 0 && c.weight_dtype == M3_F32 &&
                            expert_ffn_f32_self_routing(S, Etot) && !expert_ffn_f32_tiled(S, E, D, F);
    const bool fused_route = c.fuse_route == 1 && world == 1 && S <= 256 && (E == 16 || E == 32 || E == 64);
    // fuse_route = 2 ("split route"): the embed half of every layer's router product comes from one GEMM per forward
    // ("router_e_all"), the x half is a K = D GEMM with norm_ff folded in (output-side LayerNorm, the embed half added as
    // the epilogue residual) instead of the K = De + D GEMM with a LayerNorm prologue, and the expert kernel applies
    // norm_ff while it gathers rows, so xn is never materialised
    const bool split_route = c.fuse_route == 2 && world == 1 && S < 1024 && (Etot == 8 || Etot == 16 || Etot == 32 || Etot == 64);
    bool router_gate = false;   // the dedicated router kernel also did SoftmaxTopK (no moe_top1 stage)
    bool use_xq = false;        // ... and left the rows quantised for the fused fp8 expert kernel
    bool split_self = false;    // split route whose expert launch routes for itself (slabs hold ORIGINAL rows, b2 inside)
    int32_t* fs_dev = nullptr;  // the fused fp8 kernel's device-side F split (slab count), when it is allowed to choose
    if (split_route) {
      GemmParams r;
      r.A = x; r.lda = D; r.W = w.router_x.w; r.bias = w.router_x.b; r.ln_wsum = w.router_x.wsum; r.ln_eps = eps;
      r.Y = rl; r.ldy = Etot; r.M = S; r.N = Etot; r.K = D;
      r.resid = pl.eall + (size_t)layer * E; r.ldr = c.num_blocks * E;
      add_gemm(e, pfx + "moe_router", r, true);
      // (the expert launch routes for itself here too: no index launch; M3_SELF_ROUTE=0 puts it back)
      split_self = self_route_enabled() && c.weight_dtype == M3_F32 && expert_ffn_f32_self_routing(S, Etot) && !expert_ffn_f32_tiled(S, E, D, F);
      if (split_self) {
        add_stage(e, pfx + "moe_local.expert", 1, [=](hipStream_t s) {
          return launch_expert_route_ffn_f32(x, D, rl, live_len, live_rpb, S, E, D, F, ew1, eb1, ew2, 1, eb2, mw.slab, gidx, gval,
                                             mw.mapping, mw.acc, mw.pos, s, ng, nb, eps);
        }, stage_info("expert_ffn_f32_kernel", 1, -1.0, 4.0 * D * F * S));
      } else {
      add_stage(e, pfx + "moe_gate_index", 1, [=](hipStream_t s) {
        return launch_moe_gate_index(rl, Etot, live_len, live_rpb, S, gidx, gval, mw.mapping, mw.acc, mw.pos, s);
      }, stage_info("moe_index_kernel", 1, (double)S * (Etot * 4 + 16) + 4.0 * (E + 1), 0.0));
      add_stage(e, pfx + "moe_local.expert", 1, [=](hipStream_t s) {
        return launch_expert_ffn_f32(x, D, mw.pos, mw.acc, S, E, D, F, ew1, eb1, ew2, 1, mw.slab, ng, nb, eps, s);
      }, stage_info("expert_ffn_f32_kernel", 1, -1.0, 4.0 * D * F * S));
      }
    } else if (fused_route) {
      // router (x half, norm_ff folded; embed half precomputed for all layers by "router_e_all") + SoftmaxTopK +
      // ScatterMapping in ONE launch; the expert kernel applies norm_ff itself while it gathers rows
      const float* wx = w.router_x.w; const float* wsum = w.router_x.wsum; const float* rb = w.router_x.b;
      const float* eall = pl.eall + (size_t)layer * E;
      const int ld_e = c.num_blocks * E;
      add_stage(e, pfx + "moe_route", 1, [=](hipStream_t s) {
        return launch_moe_route(x, D, D, wx, wsum, rb, eall, ld_e, eps, lens, Tp, S, E, gidx, gval, mw.mapping, mw.acc,
                                mw.pos, s);
      }, stage_info("moe_route_kernel", 1, (double)E * D * 4 + (double)S * (D + E) * 4 + 16.0 * S, 2.0 * S * E * D));
      add_stage(e, pfx + "moe_local.expert", 1, [=](hipStream_t s) {
        return launch_expert_ffn_f32(x, D, mw.pos, mw.acc, S, E, D, F, ew1, eb1, ew2, 1, mw.slab, ng, nb, eps, s);
      }, stage_info("expert_ffn_f32_kernel", 1, -1.0, 4.0 * D * F * S));
    } else {
    GemmParams r;
    r.mode = GEMM_A_CONCAT2; r.A = pl.emb; r.lda = De; r.K1 = De; r.A2 = x; r.lda2 = D;
    r.W = w.router.w; r.bias = w.router.b; r.Y = rl; r.ldy = Etot; r.M = S; r.N = Etot; r.K = De + D;
    // LayerNorm(norm_ff) rides in the router GEMM: applied to the x half of cat([embed, x]) and written
    // out once as xn, the expert FFN's input
    r.ln_gamma = ng; r.ln_beta = nb; r.ln_eps = eps; r.ln_on_a2 = 1; r.ln_out = xn; r.ld_ln_out = D;
    // from 2048 rows on (A/B at the three BASELINE shapes, one device: configs[4]-share 46.0 -> 29.1 us per layer, forward
    // 8.83 -> 8.52 ms; configs[2] 14.1 vs 14.9 us and B = 1 +1.5 us per layer: there the 16-column work-groups of gemm.hip
    // spread the 128-KB weight pull over more CUs).  M3_ROUTER_MIN_ROWS overrides (read once).
    static const int router_min_rows = [] { const char* ev = getenv("M3_ROUTER_MIN_ROWS"); return ev ? atoi(ev) : 2048; }();
    router_gate = moe_router_supports(De, D, Etot) && S >= router_min_rows && moe_router_fuses_top1(Etot) && S > gate_index_max_rows();
    if (moe_router_supports(De, D, Etot) && S >= router_min_rows) {
      // the dedicated kernel: one work-group per 16 rows and all experts, every activation byte read once (moe_router.hip)
      const float* emb = pl.emb; const float* rw = w.router.w; const float* rb = w.router.b;
      // fp8 arithmetic, all experts local, the fused expert kernel next: the rows ALSO leave the router kernel quantised (e4m3 + a
      // scale per row): the expert kernel reads 512 B per row instead of 2 KB.  M3_ROUTER_XQ=0: as before
      static const int xq_on = [] { const char* ev = getenv("M3_ROUTER_XQ"); return ev ? atoi(ev) : 1; }();
      use_xq = xq_on && pl.xq != nullptr && world == 1 && c.ep_stages <= 0 && c.weight_dtype == M3_FP8 && w.h_scale > 0.f &&
               expert_ffn_w8a8_fused(S, E, D, F);     // (the form launch_expert_ffn_w8a8 will take)
      unsigned char* xq = use_xq ? pl.xq : nullptr; float* xqs = use_xq ? pl.xq_scale : nullptr;
      // (the fp32 rows stay available as the "xn" buffer -- the calibration tools read them -- unless M3_ROUTER_SKIP_XN=1)
      static const int skip_xn = [] { const char* ev = getenv("M3_ROUTER_SKIP_XN"); return ev ? atoi(ev) : 0; }();
      float* xn_out = (use_xq && skip_xn && !c.debug_taps) ? nullptr : xn;
      if (xn_out == nullptr) e->cur.xn_skipped = true;   // ("xn" is then not offered as a buffer: a reader fails instead of reading stale rows)
      add_stage(e, pfx + "moe_router", 1, [=](hipStream_t s) {
        // (+ SoftmaxTopK in its tail when the row-parallel top-1 launch would follow: gate_idx / gate_value come from here)
        return launch_moe_router(emb, De, De, x, D, D, rw, rb, ng, nb, eps, xn_out, D, rl, Etot, S, Etot, pdev, s,
                                 router_gate ? gidx : nullptr, router_gate ? gval : nullptr, live_len, live_rpb, xq, xqs);
      }, stage_info("moe_router_kernel", 1, (double)Etot * (De + D) * 4 + (double)S * (De + 2 * D + Etot) * 4, 2.0 * S * Etot * (De + D)));
    } else {
      add_gemm(e, pfx + "moe_router", r, true);
    }
    const bool ep = world > 1 || c.ep_stages > 0;
    if (ep) {
      // ---- expert parallel (m3asr/ep.py drives the two all-to-alls between these stages; FastMoE semantics
      //      trainer_3m_fix/fmoe/functions.py:13-86,175-199): nothing returns to the host, the exchange has a fixed shape
      //      wire [world][1 + C][D]: chunk j = what goes to / came from rank j, header row = E_loc row counts ----
      const int cap = pl.ep_cap, R = pl.ep_rows;
      int32_t *g_acc = pl.ep_acc, *g_map = pl.ep_mapping, *g_pos = pl.ep_pos, *map_send = pl.ep_map_send, *gate_recv = pl.ep_gate_recv;
      int32_t* ep_overflow = pl.ep_overflow;
      float *wire_a = pl.wire_a, *wire_b = pl.wire_b;
      const MoeWorkspace rw = carve_moe_workspace(mws, R, E, D, F);     // receive side: R wire rows over the E local experts
      // top-1 + local index over GLOBAL expert ids (the same kernel choice by row count as with all experts local)
      const bool one_launch = S <= gate_index_max_rows() && (Etot == 8 || Etot == 16 || Etot == 32 || Etot == 64);
      if (one_launch) {
        add_stage(e, pfx + "moe_gate_index", 1, [=](hipStream_t s) {
          return launch_moe_gate_index(rl, Etot, live_len, live_rpb, S, gidx, gval, g_map, g_acc, g_pos, s);
        }, stage_info("moe_index_kernel", 1, (double)S * (Etot * 4 + 16) + 4.0 * (Etot + 1), 0.0));
      } else if (!router_gate) {
        add_stage(e, pfx + "moe_top1", 1, [=](hipStream_t s) {
          return launch_softmax_top1(rl, Etot, live_len, live_rpb, S, Etot, gidx, gval, s);
        }, stage_info("softmax_top1_kernel", 1, (double)S * (Etot * 4 + 8), 0.0));
      }
      // wire row of every token (+ count headers), rows scattered straight into the send wire
      add_stage(e, pfx + "moe_ep.send", one_launch ? 1 : 2, [=](hipStream_t s) {
        if (!one_launch)
          if (int rc = launch_moe_index(gidx, S, Etot, g_map, g_acc, g_pos, s)) return rc;
        return launch_ep_send_rows(gidx, g_map, g_acc, S, world, E, cap, map_send, xn, wire_a, D * 4, s, ep_overflow);
      }, stage_info("ep_send_rows_kernel", one_launch ? 1 : 2, (double)S * D * 8 + 24.0 * S, 0.0));
      if (world == 1) add_stage(e, pfx + "moe_ep.exchange1", 0, [=](hipStream_t s) {   // one rank: the all-to-all is a copy
        M3_CHECK_HIP(hipMemcpyAsync(wire_b, wire_a, (size_t)R * D * 4, hipMemcpyDeviceToDevice, s));
        return 0;
      });
      // this rank's experts on everything it received (its own stable index puts the rows in FastMoE's receive order: by
      // local expert, then source rank, then wire order); results return to wire_a at the wire rows they came in on
      const bool e16 = c.weight_dtype != M3_F32, e8 = c.weight_dtype == M3_FP8;
      const float *es1 = w.es1, *es2 = w.es2;
      const float h_scale = w.h_scale;
      const int wmode = e8 ? (h_scale > 0.f ? 3 : 2) : (e16 ? 1 : 0);
      const int elaunches = e16 ? expert_ffn_w16_launches(wmode, R, E, D, F) : (expert_ffn_f32_tiled(R, E, D, F) ? 2 : 1);
      const float* erows = e16 ? expert_ffn_w16_rows(wmode, rw.slab, R, E, D, F) : expert_ffn_f32_rows(rw.slab, R, E, D, F);
      const int eslices = e16 ? expert_ffn_w16_slices(wmode, R, E, D, F) : expert_ffn_f32_slices(R, E, D, F);
      // bf16 experts in the tiled two-GEMM form: GEMM-2's epilogue adds b2 and puts every row straight back on its wire row
      // (no un-permuting combine launch; wire rows nobody sent keep stale bytes -- no rank ever reads them back)
      const bool scatter2 = wmode == 1 && expert_ffn_bf16_tiled(R, E, D, F);
      add_stage(e, pfx + "moe_ep.expert", (scatter2 ? 2 : 3) + elaunches, [=](hipStream_t s) {
        if (int rc = launch_ep_recv_gate(wire_b, world, E, cap, D * 4, gate_recv, s)) return rc;
        if (int rc = launch_moe_index(gate_recv, R, E, rw.mapping, rw.acc, rw.pos, s)) return rc;
        if (scatter2) return launch_expert_ffn_bf16w(wire_b, D, rw.pos, rw.acc, R, E, D, F, ew1, eb1, ew2, 1, rw.slab, s, eb2, wire_a);
        int rc;
        if (e8) rc = launch_expert_ffn_w8a8(wire_b, D, rw.pos, rw.acc, R, E, D, F, ew1, es1, eb1, ew2, es2, 1, h_scale, rw.slab, s);
        else if (e16) rc = launch_expert_ffn_bf16w(wire_b, D, rw.pos, rw.acc, R, E, D, F, ew1, eb1, ew2, 1, rw.slab, s);
        else rc = launch_expert_ffn_f32(wire_b, D, rw.pos, rw.acc, R, E, D, F, ew1, eb1, ew2, 1, rw.slab, nullptr, nullptr, 0.f, s);
        if (rc) return rc;
        return launch_moe_combine(erows, eslices, rw.mapping, gate_recv, nullptr, eb2, nullptr, 1.f, nullptr, nullptr, 0.f, wire_a, R, D, s);
      }, stage_info(e16 ? expert_ffn_w16_kernel(wmode, R, E, D, F)
                        : (expert_ffn_f32_tiled(R, E, D, F) ? "expert_gemm_f32_tiled_kernel" : "expert_ffn_f32_kernel"),
                    1, -1.0, 4.0 * D * F * S));
      if (world == 1) add_stage(e, pfx + "moe_ep.exchange2", 0, [=](hipStream_t s) {
        M3_CHECK_HIP(hipMemcpyAsync(wire_b, wire_a, (size_t)R * D * 4, hipMemcpyDeviceToDevice, s));
        return 0;
      });
      // local_gather + gate + residual + LayerNorm: token s reads its result at the wire row it was sent from
      add_stage(e, pfx + "moe_ep.combine", 1, [=](hipStream_t s) {
        return launch_moe_combine(wire_b, 1, map_send, nullptr, gv, nullptr, x, 0.5f, fg, fb, eps, x, S, D, s, a16 ? xb : nullptr,
                                  dma ? xstats : nullptr);
      }, stage_info("moe_combine_kernel", 1, (double)S * D * 12 + (a16 ? 2.0 * S * D : 0.0), (double)S * D * 11));
      e->cur.buffers["ep.wire_a"] = Buf{wire_a, (size_t)R * D * 4};
      e->cur.buffers["ep.wire_b"] = Buf{wire_b, (size_t)R * D * 4};
    } else {
    // "moe_local.*": index + grouped expert FFN + combine with all experts local
    if (self_route) {
      // short inputs, fp32: SoftmaxTopK + ScatterMapping happen inside the expert launch (every work-group derives its
      // expert's rows from the router logits; moe_expert.hip "self-routing form") -- no index launch
      add_stage(e, pfx + "moe_local.expert", 1, [=](hipStream_t s) {
        return launch_expert_route_ffn_f32(xn, D, rl, live_len, live_rpb, S, E, D, F, ew1, eb1, ew2, 1, eb2, mw.slab, gidx, gval,
                                           mw.mapping, mw.acc, mw.pos, s);
      }, stage_info("expert_ffn_f32_kernel", 1, -1.0, 4.0 * D * F * S));
    } else {
    if (S <= gate_index_max_rows() && (Etot == 8 || Etot == 16 || Etot == 32 || Etot == 64)) {
      // SoftmaxTopK plugin + ScatterMapping kernel of the reference in ONE launch (a single workgroup: right for a
      // few hundred rows; long batches take the row-parallel top-1 kernel + the index kernel below)
      add_stage(e, pfx + "moe_gate_index", 1, [=](hipStream_t s) {
        return launch_moe_gate_index(rl, Etot, live_len, live_rpb, S, gidx, gval, mw.mapping, mw.acc, mw.pos, s);
      }, stage_info("moe_index_kernel", 1, (double)S * (Etot * 4 + 16) + 4.0 * (E + 1), 0.0));
    } else {
      if (!router_gate)
      add_stage(e, pfx + "moe_top1", 1, [=](hipStream_t s) {
        return launch_softmax_top1(rl, Etot, live_len, live_rpb, S, Etot, gidx, gval, s);
      }, stage_info("softmax_top1_kernel", 1, (double)S * (Etot * 4 + 8), 0.0));
      add_stage(e, pfx + "moe_local.index", 1, [=](hipStream_t s) {
        return launch_moe_index(gidx, S, E, mw.mapping, mw.acc, mw.pos, s);
      }, stage_info("moe_index_kernel", 1, 12.0 * S + 4.0 * (E + 1), 0.0));
    }
    const bool e16 = c.weight_dtype != M3_F32, e8 = c.weight_dtype == M3_FP8;
    const float *es1 = w.es1, *es2 = w.es2;
    const float h_scale = w.h_scale;
    // all experts local, fused fp8 kernel, M3_FUSED8_ADAPT=1: the kernel may split F finer than the host's choice when the routing
    // leaves CUs without an item (the expert-parallel receive side always keeps the host's split: its result stays comparable bit
    // for bit with any other grouping of the same rows under the same split).  A latency / throughput trade, OFF by default:
    // measured at configs[4]'s share on one box, the launch 38.5 -> 27.8 us and one forward alone 6.70 -> 6.57 ms, but 4.58 -> 4.45 M
    // frames/s at four contexts (twice the partial-output slabs; the idle CUs were being used by the other contexts' kernels)
    static const int adapt_on = [] { const char* ev = getenv("M3_FUSED8_ADAPT"); return ev ? atoi(ev) : 0; }();
    fs_dev = (adapt_on && e8 && h_scale > 0.f && pl.moe_fs != nullptr && !c.debug_taps && expert_ffn_w8a8_fused(S, E, D, F)) ? pl.moe_fs : nullptr;
    const int wmode = e8 ? (h_scale > 0.f ? 3 : 2) : (e16 ? 1 : 0);
    const int elaunches = e16 ? expert_ffn_w16_launches(wmode, S, E, D, F) : (expert_ffn_f32_tiled(S, E, D, F) ? 2 : 1);
    add_stage(e, pfx + "moe_local.expert", elaunches, [=](hipStream_t s) {
      if (e8) return launch_expert_ffn_w8a8(xn, D, mw.pos, mw.acc, S, E, D, F, ew1, es1, eb1, ew2, es2, 1, h_scale, mw.slab, s,
                                            use_xq ? pl.xq : nullptr, use_xq ? pl.xq_scale : nullptr, fs_dev);
      if (e16) return launch_expert_ffn_bf16w(xn, D, mw.pos, mw.acc, S, E, D, F, ew1, eb1, ew2, 1, mw.slab, s);
      return launch_expert_ffn_f32(xn, D, mw.pos, mw.acc, S, E, D, F, ew1, eb1, ew2, 1, mw.slab, nullptr, nullptr, 0.f, s);
    }, stage_info(e16 ? expert_ffn_w16_kernel(wmode, S, E, D, F)
                      : (expert_ffn_f32_tiled(S, E, D, F) ? "expert_gemm_f32_tiled_kernel" : "expert_ffn_f32_kernel"),
                  1, -1.0, 4.0 * D * F * S));
    }
    }
    }
    if (!(world > 1 || c.ep_stages > 0)) {
    // long batches run the expert FFN as two grouped GEMMs whose result is ONE slab of sorted rows (never with fused_route: S <= 256)
    const bool e16c = c.weight_dtype != M3_F32;
    const int wmode_c = c.weight_dtype == M3_FP8 ? (w.h_scale > 0.f ? 3 : 2) : 1;
    const float* erows = (fused_route || split_route) ? mw.slab : (e16c ? expert_ffn_w16_rows(wmode_c, mw.slab, S, E, D, F) : expert_ffn_f32_rows(mw.slab, S, E, D, F));
    const int eslices = (fused_route || split_route) ? F / kExpertSlice : (e16c ? expert_ffn_w16_slices(wmode_c, S, E, D, F) : expert_ffn_f32_slices(S, E, D, F));
    // (self-routing expert launch: slabs hold ORIGINAL rows with b2 already in slice 0 -> no mapping, no b2 here)
    const int32_t* cmap = (self_route || split_self) ? nullptr : mw.mapping;
    const float* cb2 = (self_route || split_self) ? nullptr : eb2;
    add_stage(e, pfx + "moe_local.combine", 1, [=](hipStream_t s) {
      return launch_moe_combine(erows, fs_dev ? 4 : eslices, cmap, gidx, gv, cb2, x, 0.5f, fg, fb, eps, x, S, D, s, a16 ? xb : nullptr,
                                dma ? xstats : nullptr, fs_dev);
    }, stage_info("moe_combine_kernel", 1, (double)S * D * 4 * (eslices + 2) + (a16 ? 2.0 * S * D : 0.0), (double)S * D * (eslices + 10)));
    }
    const std::string b = pfx.substr(0, pfx.size() - 1);
    e->cur.buffers[b + ".gate_idx"] = Buf{gidx, (size_t)S * 4};
    e->cur.buffers[b + ".gate_value"] = Buf{gval, (size_t)S * 4};
    e->cur.buffers[b + ".mapping"] = Buf{mw.mapping, (size_t)S * 4};
    e->cur.buffers[b + ".acc_histogram"] = Buf{mw.acc, (size_t)(E + 1) * 4};
  }
  if (c.debug_taps) {
    float* tap = pl.taps + (size_t)tap_index * S * D;
    add_stage(e, pfx + "tap", 0, [=](hipStream_t s) {
      M3_CHECK_HIP(hipMemcpyAsync(tap, x, (size_t)S * D * sizeof(float), hipMemcpyDeviceToDevice, s));
      return 0;
    });
    e->cur.buffers[pfx.substr(0, pfx.size() - 1) + ".out"] = Buf{tap, (size_t)S * D * 4};
  }
}

extern "C" {

int m3_engine_output_frames(int T) { return T >= 7 ? sub_len(T) : 0; }

m3_engine* m3_engine_create(const m3_engine_config* config, const m3_weight_entry* table, int n_entries) {
  if (!config || !table || n_entries <= 0) {
    set_error("engine_create: null config / weight table");
    return nullptr;
  }
  m3_engine* e = new m3_engine;
  e->cfg = *config;
  const m3_engine_config& c = e->cfg;
  auto fail = [&](const char* msg) -> m3_engine* {
    if (msg) set_error("%s", msg);
    delete e;
    return nullptr;
  };
  if (c.attention_dim % c.attention_heads || c.embed_dim % c.embed_heads) return fail("engine_create: dim % heads != 0");
  if (c.attention_dim % 16 || c.embed_dim % 16 || c.hidden_units % 64 || c.embed_linear_units % 16)
    return fail("engine_create: dims must be multiples of 16 (hidden_units of 64)");
  if (c.embed_dim != c.attention_dim) return fail("engine_create: embed_dim != attention_dim is not supported");
  if (c.weight_dtype != M3_F32 && c.weight_dtype != M3_BF16 && c.weight_dtype != M3_FP8)
    return fail("engine_create: weight_dtype must be f32, bf16 or fp8");
  if (c.weight_dtype == M3_FP8 && (c.attention_dim % 64 || c.hidden_units % 64))
    return fail("engine_create: fp8 experts need attention_dim and hidden_units that are multiples of 64");
  if (c.fp8_activations && c.weight_dtype != M3_FP8) return fail("engine_create: fp8_activations needs weight_dtype fp8");
  if (c.ep_stages > 0 && c.fuse_route) return fail("engine_create: ep_stages needs the staged route (fuse_route = 0)");
  if (c.ep_world_size > 1 && (c.ep_rank < 0 || c.ep_rank >= c.ep_world_size)) return fail("engine_create: ep_rank outside [0, ep_world_size)");
  if (c.weight_dtype != M3_F32) {
    if (c.fuse_route) return fail("engine_create: fuse_route is fp32-only");
    if (c.attention_dim % 32 || c.hidden_units % 64 || c.embed_linear_units % 32)
      return fail("engine_create: bf16 weights need dims that are multiples of 32");
  }
  e->names.reserve(n_entries);
  for (int i = 0; i < n_entries; ++i) {
    if (!table[i].name || !table[i].data) return fail("engine_create: null weight entry");
    e->names.emplace_back(table[i].name);
    m3_weight_entry w = table[i];
    w.name = nullptr;
    e->table[e->names.back()] = w;
  }
  const int D = c.attention_dim, De = c.embed_dim, K = c.cnn_module_kernel;
  if (!load_sub(e, "embed.subsampling.", De, c.input_dim, &e->sub_e) || !load_sub(e, "subsampling.", D, c.input_dim, &e->sub_m))
    return fail(nullptr);
  if (!load_norm(e, "embed.after_norm.", De, &e->e_after) ||
      !load_lin_ln(e, "out_linear.", c.output_dim, D, false, &e->out_linear))
    return fail(nullptr);
  {   // optional front / back end tensors
    auto m = e->table.find("cmvn.mean"), v = e->table.find("cmvn.istd"), ob = e->table.find("output_bias");
    if ((m == e->table.end()) != (v == e->table.end())) return fail("engine_create: cmvn.mean and cmvn.istd go together");
    if (m != e->table.end()) {
      if (m->second.numel != c.input_dim || v->second.numel != c.input_dim) return fail("engine_create: cmvn vectors must have input_dim entries");
      e->cmvn_mean = (const float*)m->second.data;
      e->cmvn_istd = (const float*)v->second.data;
    }
    if (ob != e->table.end()) {
      if (ob->second.numel != c.output_dim) return fail("engine_create: output_bias must have output_dim entries");
      e->output_bias = (const float*)ob->second.data;
    }
  }
  {
    auto it = e->table.find("pe");
    if (it == e->table.end() || it->second.numel % D) return fail("engine_create: positional table 'pe' missing");
    e->pe = (const float*)it->second.data;
    e->pe_rows = it->second.numel / D;
  }
  if (!lookup(e, "pos_all.weight", (int64_t)(c.embed_blocks + c.num_blocks) * D * D, &e->pos_all,
              c.weight_dtype == M3_F32 ? M3_F32 : M3_BF16))
    return fail(nullptr);
  if (c.fuse_route && !lookup(e, "router_e_all.weight", (int64_t)c.num_blocks * c.num_experts * De, &e->router_e_all))
    return fail(nullptr);
  e->eblocks.resize(c.embed_blocks);
  for (int i = 0; i < c.embed_blocks; ++i)
    if (!load_block(e, "embed.blocks." + std::to_string(i) + ".", De, c.embed_linear_units, K, c.embed_cnn_layer_norm,
                    false, De, &e->eblocks[i], c.embed_causal > 0))
      return fail(nullptr);
  e->mblocks.resize(c.num_blocks);
  for (int i = 0; i < c.num_blocks; ++i)
    if (!load_block(e, "blocks." + std::to_string(i) + ".", D, c.hidden_units, K, c.cnn_layer_norm, true, De,
                    &e->mblocks[i], c.causal > 0))
      return fail(nullptr);
  return e;
}

void m3_engine_destroy(m3_engine* engine) {
  if (!engine) return;
  if (engine->ev_fork) (void)hipEventDestroy(engine->ev_fork);
  if (engine->ev_join) (void)hipEventDestroy(engine->ev_join);
  if (engine->side) (void)hipStreamDestroy(engine->side);
  if (engine->cur.graph_exec) (void)hipGraphExecDestroy(engine->cur.graph_exec);
  for (auto& b : engine->parked)
    if (b.graph_exec) (void)hipGraphExecDestroy(b.graph_exec);
  delete engine;
}

size_t m3_engine_workspace_size(const m3_engine* engine, int B, int T) {
  if (!engine || B <= 0 || T < 7) return 0;
  return make_plan(engine->cfg, nullptr, B, T, engine->ep_capacity).bytes;
}

static int prepare_impl(m3_engine* e, const float* feat, const int32_t* feat_len, int B, int T, float* logits,
                        void* workspace, size_t workspace_bytes, void* sstate, int s_hist, int s_maxf) {
  M3_REQUIRE(e && feat && feat_len && logits && workspace, "engine_prepare: null argument");
  M3_REQUIRE(B > 0 && T >= 7, "engine_prepare: need B > 0 and T >= 7 frames (got B=%d T=%d)", B, T);
  const m3_engine_config& c = e->cfg;
  const int Tp = sub_len(T);
  M3_REQUIRE(Tp < e->pe_rows, "engine_prepare: T'=%d exceeds the positional table (%lld rows)", Tp,
             (long long)e->pe_rows);  // rel_positional_encoding_plugin.cpp:139-142
  if (int rc = init_expert_ffn_kernels()) return rc;
  if (int rc = init_expert_ffn_bf16_kernels()) return rc;
  if (int rc = init_expert_ffn_w8_kernels()) return rc;
  if (int rc = init_gemm_bf16_tiled_kernels()) return rc;
  if (int rc = init_expert_ffn_f32_tiled_kernels()) return rc;
  if (int rc = init_expert_ffn_fused_fp8_kernels()) return rc;
  if (int rc = init_gemm_f32_tiled_kernels()) return rc;
  if (int rc = init_gemm_bf16_dma_kernels()) return rc;
  if (int rc = init_expert_gemm_g256_kernels()) return rc;
  if (int rc = init_moe_router_kernels()) return rc;
  if (int rc = init_relpos_attention_bf16_kernels()) return rc;
  Plan pl = make_plan(c, workspace, B, T, e->ep_capacity);
  M3_REQUIRE(workspace_bytes >= pl.bytes, "engine_prepare: workspace %zu bytes < required %zu", workspace_bytes, pl.bytes);
  // ---- shape cache: park the current binding, revive a parked one with the same (shape, buffers) ----
  if (e->cur.matches(B, T, feat, feat_len, logits, workspace, workspace_bytes, e->ep_capacity, sstate, s_hist, s_maxf)) {
    e->cur.last_use = ++e->use_clock;
    return (int)e->cur.stages.size();
  }
  const int capacity = c.shape_cache > 0 ? c.shape_cache : (c.shape_cache < 0 ? 0 : 7);
  if (!e->cur.stages.empty()) {
    if (capacity > 0) {
      if ((int)e->parked.size() >= capacity) {           // evict the least recently used binding
        size_t lru = 0;
        for (size_t i = 1; i < e->parked.size(); ++i)
          if (e->parked[i].last_use < e->parked[lru].last_use) lru = i;
        if (e->parked[lru].graph_exec) (void)hipGraphExecDestroy(e->parked[lru].graph_exec);
        e->parked.erase(e->parked.begin() + lru);
      }
      e->parked.push_back(std::move(e->cur));
    } else if (e->cur.graph_exec) {
      (void)hipGraphExecDestroy(e->cur.graph_exec);
    }
    e->cur = m3_engine::Bound();
  }
  for (size_t i = 0; i < e->parked.size(); ++i)
    if (e->parked[i].matches(B, T, feat, feat_len, logits, workspace, workspace_bytes, e->ep_capacity, sstate, s_hist, s_maxf)) {
      e->cur = std::move(e->parked[i]);
      e->parked.erase(e->parked.begin() + i);
      e->cur.last_use = ++e->use_clock;
      return (int)e->cur.stages.size();
    }
  e->cur.last_use = ++e->use_clock;
  e->cur.B = B; e->cur.T = T; e->cur.Tp = Tp; e->cur.S = B * Tp;
  e->cur.feat = feat; e->cur.feat_len = feat_len; e->cur.logits = logits; e->cur.ws = workspace; e->cur.ws_bytes = workspace_bytes;
  e->cur.ep_cap = e->ep_capacity;
  e->cur.sstate = sstate; e->cur.s_hist = s_hist; e->cur.s_maxf = s_maxf;
  const bool streaming = sstate != nullptr;
  e->cur.stages.clear(); e->cur.buffers.clear(); e->cur.n_kernels = 0; e->cur.graph_valid = false; e->cur.xn_skipped = false;
  e->cur.splitk_ws = pl.splitk; e->cur.splitk_bytes = pl.splitk_bytes;
  {
    // bf16 activation operands need every GEMM that reads or rewrites them on the LDS-tiled kernel: the narrowest ones
    // are the D x D projections (the expert-parallel driver's combine op writes the bf16 copy too: m3_moe_combine_bf16)
    GemmParams t;
    t.M = B * Tp; t.N = c.attention_dim; t.K = c.attention_dim; t.w_bf16 = 1;
    e->cur.a16 = c.weight_dtype != M3_F32 && c.bf16_activations >= 0 && !c.debug_taps && c.embed_dim == c.attention_dim &&
                 (c.embed_linear_units % 128) == 0 && (c.hidden_units % 128) == 0 && gemm_bf16w_uses_tiled(t);
  }
  {
    GemmParams t;     // the narrowest block GEMM as the LDS-DMA kernel would see it
    t.M = B * Tp; t.N = c.attention_dim; t.K = c.attention_dim; t.lda = c.attention_dim; t.w_bf16 = 1; t.a_bf16 = 1;
    e->cur.dma = e->cur.a16 && c.attention_dim == 128 * kXbStatParts && gemm_bf16w_uses_dma(t);
  }
  e->cur.packed = use_packed_rows(c, B);
  if (streaming) {   // a chunk is a few rows per utterance: padded layout, fp32 activations (the 16-bit modes keep their bf16 weights)
    e->cur.a16 = e->cur.dma = e->cur.packed = false;   // (the plan's packed-row buffers stay carved, unused)
  }
  if (int rc = init_gemm_f32_splitk_kernels()) return rc;
  const int S = e->cur.S, D = c.attention_dim, De = c.embed_dim;

  if (pl.ep_overflow != nullptr) {   // bounded expert-parallel wire: the overflow report of this forward starts at 0
    int32_t* ovf = pl.ep_overflow;
    add_stage(e, "ep.reset", 0, [=](hipStream_t s) {
      M3_CHECK_HIP(hipMemsetAsync(ovf, 0, sizeof(int32_t), s));
      return 0;
    });
    e->cur.buffers["ep.overflow"] = Buf{ovf, sizeof(int32_t)};
  }
  // valid lengths after the two stride-2 convs (MaskConv2dSample x2, subsampling.py:119-137)
  {
    // No launch of their own: the packed layout's row plan forms them on its way; otherwise the forward's first kernel (the
    // embed subsampler's conv1) does.  Exceptions that keep the separate "lens" stage: a forked capture (the main branch reads
    // the lengths while the embed branch, whose conv1 would write them, runs beside it) and a positional projection that is
    // not folded (its GEMM would sit in front of conv1 -- harmless, but the stage order of round 1 is kept for it).
    int32_t* lens = pl.lens;
    e->cur.lens_in_conv1 = false;
    if (e->cur.packed) {   // row plan of the packed layout: first row of every utterance, packed -> padded row map
      int32_t* row0 = pl.row0; int32_t* pad_of = pl.pad_of;
      add_stage(e, "pack_plan", 1, [=](hipStream_t s) { return launch_pack_plan(lens, B, Tp, row0, pad_of, s, feat_len); },
                stage_info("pack_plan_kernel", 1, 12.0 * B + 4.0 * B * Tp, 0.0, false));
    } else if (pl.fork || (!c.fold_pos_proj && !streaming)) {
      add_stage(e, "lens", 1, [=](hipStream_t s) { return launch_subsample_lens(feat_len, B, lens, s); },
                stage_info("subsample_lens_kernel", 1, 8.0 * B, 0.0, false));
    } else {
      e->cur.lens_in_conv1 = true;
    }
  }
  // ---- p = linear_pos(pe[:T']) for all blocks at once (attention.py:345; input-independent, so with
  //      fold_pos_proj it is computed once per bound shape instead of once per forward) ----
  {
    const int nb = c.num_blocks + c.embed_blocks;
    GemmParams pp;
    // (streaming: keys carry their ABSOLUTE position, rel_positional_encoding_kernel.cu:108-111 pe[offset : offset + T]: one
    //  table over the s_maxf positions a stream can reach, always folded)
    const int Tpos = streaming ? s_maxf : Tp;
    pp.A = e->pe; pp.lda = D; pp.W = e->pos_all; pp.Y = pl.pbuf; pp.ldy = nb * D; pp.M = Tpos; pp.N = nb * D; pp.K = D;
    if (c.fold_pos_proj || streaming) {
      for (auto it = e->pfold_by_tp.begin(); it != e->pfold_by_tp.end();)      // tables no binding holds any more
        it = it->second.expired() ? e->pfold_by_tp.erase(it) : std::next(it);
      std::shared_ptr<float> pf = e->pfold_by_tp.count(Tpos) ? e->pfold_by_tp[Tpos].lock() : nullptr;
      if (!pf) {
        float* dev = nullptr;
        M3_CHECK_HIP(hipMalloc((void**)&dev, (size_t)Tpos * nb * D * sizeof(float)));
        pf = std::shared_ptr<float>(dev, [](float* q) { (void)hipFree(q); });
        pp.Y = dev;
        pp.w_bf16 = c.weight_dtype != M3_F32;
        if (int rc = launch_gemm_f32(pp, nullptr)) return rc;
        M3_CHECK_HIP(hipStreamSynchronize(nullptr));
        e->pfold_by_tp[Tpos] = pf;
      }
      e->cur.pfold = pf;
      pl.pbuf = pf.get();
    } else {
      add_gemm(e, "pos_all", pp);
    }
  }
  // ---- embed encoder (conformer_embed_domain_acc.py:149-181) ----
  const Plan ple = embed_view(pl);
  const int hf_first = (int)e->cur.stages.size();       // (horizontal fusion: the embed chain starts here ...)
  if (pl.fork) e->cur.fork_first = (int)e->cur.stages.size();
  e->cur.splitk_ws = ple.splitk;
  build_subsample(e, "embed.subsample.", e->sub_e, De, ple, ple.x);
  for (int i = 0; i < c.embed_blocks; ++i) {
    if (i + 1 == c.embed_blocks) {   // after_norm joins the last block's norm_final launch
      e->cur.tail_ln_g = e->e_after.g; e->cur.tail_ln_b = e->e_after.b; e->cur.tail_ln_eps = 1e-12f; e->cur.tail_ln_out = pl.emb;
    }
    build_block(e, "embed.blocks." + std::to_string(i) + ".", e->eblocks[i], De, c.embed_linear_units, c.embed_heads,
                c.cnn_module_kernel, c.embed_cnn_layer_norm, false, i, i, ple, c.embed_causal > 0);
  }
  e->cur.splitk_ws = pl.splitk;
  if (c.embed_blocks == 0) {
    float* x = ple.x; float* emb = pl.emb;
    const float* g = e->e_after.g; const float* b = e->e_after.b;
    add_stage(e, "embed.after_norm", 1, [=](hipStream_t s) { return launch_layernorm(x, g, b, 1e-12f, emb, S, De, s); },
              stage_info("layernorm_kernel", 1, 8.0 * S * De, 8.0 * S * De));
  }
  // embed half of every layer's router product in one GEMM: emb does not change across the main blocks
  if ((c.fuse_route == 2 && c.ep_world_size <= 1 && S < 1024) ||
      (c.fuse_route == 1 && (c.ep_world_size <= 1) && S <= 256 &&
       (c.num_experts == 16 || c.num_experts == 32 || c.num_experts == 64))) {
    GemmParams g;
    g.A = pl.emb; g.lda = De; g.W = e->router_e_all; g.Y = pl.eall; g.ldy = c.num_blocks * c.num_experts;
    g.M = S; g.N = c.num_blocks * c.num_experts; g.K = De;
    add_gemm(e, "router_e_all", g, true);
  }
  // ---- main MoE encoder (conformer_fmoe_localComm_catEmbed_domain_acc_hier.py:198-234) ----
  if (pl.fork) e->cur.fork_mid = (int)e->cur.stages.size();
  const int hf_mid = (int)e->cur.stages.size();         // (... and the main encoder's independent prefix here)
  build_subsample(e, "subsample.", e->sub_m, D, pl, pl.x);
  for (int i = 0; i < c.num_blocks; ++i)
    build_block(e, "blocks." + std::to_string(i) + ".", e->mblocks[i], D, c.hidden_units, c.attention_heads,
                c.cnn_module_kernel, c.cnn_layer_norm, true, i, c.embed_blocks + i, pl, c.causal > 0);
  {
    GemmParams g;
    g.A = pl.x; g.lda = D; g.W = e->out_linear.w; g.bias = e->out_linear.b; g.Y = logits; g.ldy = c.output_dim;
    g.M = S; g.N = c.output_dim; g.K = D;
    g.ln_wsum = e->out_linear.wsum; g.ln_eps = 1e-12f;       // after_norm is folded into out_linear
    if (e->cur.a16) { g.A = (const float*)pl.xb; g.a_bf16 = 1; }
    if (e->cur.dma) { g.ln_stats = pl.xstats; g.ln_stat_parts = kXbStatParts; }
    float* lout = logits;
    if (e->cur.packed) {   // packed rows -> packed logits; the (B, T', V) output is filled from them at the end
      lout = pl.lpk;
      g.Y = lout;
      g.m_dev = pl.row0 + B;
    }
    add_gemm(e, "logits", g);
    const float* ob = e->output_bias;
    const int V = c.output_dim;
    if (c.log_softmax_out) {
      add_stage(e, "log_softmax", 1, [=](hipStream_t s) { return launch_log_softmax_bias(lout, ob, lout, (size_t)S, V, s); },
                stage_info("log_softmax_bias_kernel", 1, 8.0 * S * V, 4.0 * S * V));
    }   // without log-softmax a prior is folded into out_linear's bias when the plan is packed (plan.py)
    if (e->cur.packed) {
      const int32_t* row0 = pl.row0;
      add_stage(e, "unpack", 1, [=](hipStream_t s) { return launch_unpack_rows(lout, row0, B, Tp, V, logits, s); },
                stage_info("unpack_rows_kernel", 1, 8.0 * S * V, 0.0, false));
    }
  }
  if (pl.fork) {     // the main branch joins the embed branch at the first stage that reads the embedding: blocks.0's router
    for (size_t i = 0; i < e->cur.stages.size(); ++i)
      if (e->cur.stages[i].name == "blocks.0.moe_router" || e->cur.stages[i].name == "blocks.0.moe_route") {
        e->cur.join_at = (int)i;
        break;
      }
    if (e->cur.join_at < 0 || c.num_blocks < 1) e->cur.fork_first = e->cur.fork_mid = e->cur.join_at = -1;
  }
  if (pl.hfuse && !streaming && hf_first >= 0) {
    int join = -1;
    for (size_t i = (size_t)hf_mid; i < e->cur.stages.size(); ++i) {
      const std::string& n = e->cur.stages[i].name;
      if (n == "blocks.0.moe_router" || n == "blocks.0.moe_route" || n == "blocks.0.moe_gate_index") { join = (int)i; break; }
    }
    if (join > 0) fuse_independent_pairs(e, hf_first, hf_mid, join);
  }
  if (streaming) {   // the chunk counter moves on the device: the same captured graph serves every chunk of the stream
    int32_t* step = carve_stream_state(c, sstate, B, s_hist).step;
    add_stage(e, "stream.advance", 1, [=](hipStream_t s) { return launch_advance_counter(step, 1, s); },
              stage_info("advance_counter_kernel", 1, 8.0, 0.0, false));
  }
  e->cur.buffers["x"] = Buf{pl.x, (size_t)S * D * 4};
  if (!e->cur.xn_skipped) e->cur.buffers["xn"] = Buf{pl.xn, (size_t)S * D * 4};
  if (pl.xq != nullptr) {
    e->cur.buffers["xq"] = Buf{pl.xq, (size_t)S * D};
    e->cur.buffers["xq_scale"] = Buf{pl.xq_scale, (size_t)S * 4};
  }
  e->cur.buffers["embed"] = Buf{pl.emb, (size_t)S * De * 4};
  if (e->cur.a16) e->cur.buffers["xb"] = Buf{pl.xb, (size_t)S * D * 2};
  e->cur.buffers["lens"] = Buf{pl.lens, (size_t)B * 4};
  if (e->cur.packed) e->cur.buffers["row0"] = Buf{pl.row0, (size_t)(B + 1) * 4};
  if (pl.ep_rows) e->cur.buffers["ep.gate_recv"] = Buf{pl.ep_gate_recv, (size_t)pl.ep_rows * 4};
  e->cur.buffers["router_logits"] = Buf{pl.rl, (size_t)S * c.num_experts * (c.ep_world_size > 0 ? c.ep_world_size : 1) * 4};

  return (int)e->cur.stages.size();
}

int m3_engine_prepare(m3_engine* e, const float* feat, const int32_t* feat_len, int B, int T, float* logits,
                      void* workspace, size_t workspace_bytes) {
  return prepare_impl(e, feat, feat_len, B, T, logits, workspace, workspace_bytes, nullptr, 0, 0);
}

int m3_engine_set_ep_capacity(m3_engine* engine, int rows_per_chunk) {
  M3_REQUIRE(engine != nullptr && rows_per_chunk >= 0, "engine_set_ep_capacity: bad argument");
  engine->ep_capacity = rows_per_chunk;
  return 0;
}

int m3_engine_num_stages(const m3_engine* engine) { return engine ? (int)engine->cur.stages.size() : 0; }
const char* m3_engine_stage_name(const m3_engine* engine, int index) {
  if (!engine || index < 0 || index >= (int)engine->cur.stages.size()) return nullptr;
  return engine->cur.stages[index].name.c_str();
}
int m3_engine_num_captures(const m3_engine* engine) { return engine ? engine->n_captures : 0; }
int m3_engine_stage_info(const m3_engine* engine, int index, m3_stage_info* info) {
  M3_REQUIRE(engine && info, "engine_stage_info: null argument");
  M3_REQUIRE(index >= 0 && index < (int)engine->cur.stages.size(), "engine_stage_info: no stage %d", index);
  *info = engine->cur.stages[index].info;
  return 0;
}
int m3_engine_num_kernels(const m3_engine* engine) { return engine ? engine->cur.n_kernels : 0; }

int m3_engine_run(m3_engine* engine, int first_stage, int last_stage, m3_stream stream) {
  M3_REQUIRE(engine && !engine->cur.stages.empty(), "engine_run: engine not prepared");
  M3_REQUIRE(first_stage >= 0 && last_stage <= (int)engine->cur.stages.size() && first_stage <= last_stage,
             "engine_run: bad stage range [%d,%d)", first_stage, last_stage);
  for (int i = first_stage; i < last_stage; ++i)
    if (int rc = engine->cur.stages[i].run((hipStream_t)stream)) return rc;
  return 0;
}

int m3_engine_buffer(const m3_engine* engine, const char* name, void** ptr, size_t* bytes) {
  M3_REQUIRE(engine && name && ptr && bytes, "engine_buffer: null argument");
  auto it = engine->cur.buffers.find(name);
  M3_REQUIRE(it != engine->cur.buffers.end(), "engine_buffer: no buffer named '%s' for the bound shape", name);
  *ptr = it->second.ptr;
  *bytes = it->second.bytes;
  return 0;
}

int m3_engine_forward(m3_engine* e, const float* feat, const int32_t* feat_len, int B, int T, float* logits,
                      void* workspace, size_t workspace_bytes, int use_graph, m3_stream stream_) {
  M3_REQUIRE(e != nullptr, "engine_forward: null engine");
  hipStream_t stream = (hipStream_t)stream_;
  if (!e->cur.matches(B, T, feat, feat_len, logits, workspace, workspace_bytes, e->ep_capacity)) {
    int rc = m3_engine_prepare(e, feat, feat_len, B, T, logits, workspace, workspace_bytes);
    if (rc < 0) return rc;
  }
  e->cur.last_use = ++e->use_clock;
  M3_REQUIRE(e->cfg.ep_world_size <= 1, "engine_forward: an expert-parallel engine (ep_world_size = %d) is run stage-wise, with the "
             "all-to-all between its moe_ep.* stages (m3asr/ep.py)", (int)e->cfg.ep_world_size);
  if (!use_graph) return m3_engine_run(e, 0, (int)e->cur.stages.size(), stream_);
  if (!e->cur.graph_valid) {
    M3_REQUIRE(stream != nullptr, "engine_forward: graph capture needs a non-default stream");
    if (e->cur.graph_exec) {
      (void)hipGraphExecDestroy(e->cur.graph_exec);
      e->cur.graph_exec = nullptr;
    }
    hipGraph_t graph = nullptr;
    const int n_st = (int)e->cur.stages.size();
    const bool fork = e->cur.fork_first >= 0 && e->cur.fork_mid > e->cur.fork_first && e->cur.join_at > e->cur.fork_mid;
    if (fork && e->side == nullptr) {
      M3_CHECK_HIP(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
      M3_CHECK_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
      M3_CHECK_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    }
    M3_CHECK_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    if (!fork) {
      rc = m3_engine_run(e, 0, n_st, stream_);
    } else {
      // two branches between "lens" and blocks.0's router: the embed encoder on the side stream (it joins the capture through
      // the fork event), the main subsampler + block 0 up to its router on the capture stream
      hipError_t he = hipSuccess;
      rc = m3_engine_run(e, 0, e->cur.fork_first, stream_);
      if (!rc && (he = hipEventRecord(e->ev_fork, stream)) != hipSuccess) rc = -1;
      if (!rc && (he = hipStreamWaitEvent(e->side, e->ev_fork, 0)) != hipSuccess) rc = -1;
      if (!rc) rc = m3_engine_run(e, e->cur.fork_first, e->cur.fork_mid, (m3_stream)e->side);
      if (!rc) rc = m3_engine_run(e, e->cur.fork_mid, e->cur.join_at, stream_);
      if (!rc && (he = hipEventRecord(e->ev_join, e->side)) != hipSuccess) rc = -1;
      if (!rc && (he = hipStreamWaitEvent(stream, e->ev_join, 0)) != hipSuccess) rc = -1;
      if (!rc) rc = m3_engine_run(e, e->cur.join_at, n_st, stream_);
      if (he != hipSuccess) set_error("engine_forward: forked capture failed: %s", hipGetErrorString(he));
    }
    hipError_t ce = hipStreamEndCapture(stream, &graph);
    if (rc) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    M3_CHECK_HIP(ce);
    M3_CHECK_HIP(hipGraphInstantiate(&e->cur.graph_exec, graph, nullptr, nullptr, 0));
    M3_CHECK_HIP(hipGraphDestroy(graph));
    e->cur.graph_valid = true;
    ++e->n_captures;
  }
  M3_CHECK_HIP(hipGraphLaunch(e->cur.graph_exec, stream));
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------------
// Chunk-by-chunk (streaming) execution: decoding-chunk semantics of trainer_3m_fix/model/encoder.py:100-140
// (decoding_chunk_size / num_decoding_left_chunks -> add_optional_chunk_mask) with the caches the reference's streaming plugins
// were written for (cat_split_cache_kernel.cu:30-107, att_stream_softmax_kernel.cu:136-191,
// rel_positional_encoding_kernel.cu:108-123).  Contract: with static_chunk_size = c, causal conv modules in both encoders and
// the same weights, the logits of chunk n equal rows [n c, (n + 1) c) of the full-utterance forward (m3_engine_forward) up to
// fp32 rounding of the GEMMs (their kernels are chosen by row count); the attention core and the conv are bit-exact.
static int stream_check(const m3_engine* e, const m3_stream_desc* d) {
  M3_REQUIRE(e != nullptr && d != nullptr, "engine stream: null argument");
  const m3_engine_config& c = e->cfg;
  M3_REQUIRE(c.static_chunk_size > 0, "engine stream: the engine was built without static_chunk_size (the chunk length in output frames)");
  M3_REQUIRE(c.causal > 0 && c.embed_causal > 0, "engine stream: both encoders need causal conv modules (a symmetric depthwise conv "
             "looks %d frames into the future)", (int)(c.cnn_module_kernel - 1) / 2);
  M3_REQUIRE(c.ep_world_size <= 1 && c.ep_stages <= 0 && c.fork_embed <= 0, "engine stream: expert-parallel stages / forked capture are not available chunk by chunk");
  M3_REQUIRE(d->B > 0 && d->max_frames >= c.static_chunk_size && d->max_frames < e->pe_rows,
             "engine stream: need B > 0 and chunk <= max_frames < %lld positions (got B=%d max_frames=%d)", (long long)e->pe_rows, d->B, d->max_frames);
  const int need = c.num_left_chunks < 0 ? d->max_frames : (c.num_left_chunks + 1) * c.static_chunk_size;
  M3_REQUIRE(d->history_frames >= need, "engine stream: history_frames=%d < %d (%s)", d->history_frames, need,
             c.num_left_chunks < 0 ? "all left chunks are visible: the history must hold the whole stream" : "(num_left_chunks + 1) chunks");
  return 0;
}

int m3_engine_chunk_input_frames(const m3_engine* engine) {
  // c output frames need input frames [4 j0, 4 (j0 + c - 1) + 6]: 4 c + 3 of them, advancing by 4 c per chunk (7-frame context of
  // the two stride-2 3x3 convs, subsampling.py:103-145)
  return engine && engine->cfg.static_chunk_size > 0 ? 4 * engine->cfg.static_chunk_size + 3 : 0;
}

size_t m3_engine_stream_state_size(const m3_engine* engine, const m3_stream_desc* desc) {
  if (stream_check(engine, desc)) return 0;
  return carve_stream_state(engine->cfg, nullptr, desc->B, desc->history_frames).bytes;
}

int m3_engine_stream_reset(m3_engine* e, const m3_stream_desc* desc, void* state, size_t state_bytes, m3_stream stream_) {
  if (int rc = stream_check(e, desc)) return rc;
  const m3_engine_config& c = e->cfg;
  const StreamState st = carve_stream_state(c, state, desc->B, desc->history_frames);
  M3_REQUIRE(state != nullptr && state_bytes >= st.bytes, "engine_stream_reset: state %zu bytes < required %zu", state_bytes, st.bytes);
  hipStream_t stream = (hipStream_t)stream_;
  M3_CHECK_HIP(hipMemsetAsync(st.step, 0, 64 * sizeof(int32_t), stream));
  // the K-1 frames left of frame 0 are what the conv module's zero padding becomes behind pointwise_conv1 + GLU
  const int K = c.cnn_module_kernel, D = c.attention_dim, nb = c.embed_blocks + c.num_blocks;
  for (int i = 0; i < nb; ++i) {
    const BlockW& w = i < c.embed_blocks ? e->eblocks[i] : e->mblocks[i - c.embed_blocks];
    if (int rc = launch_fill_rows(w.left_fill, D, st.conv[i], (size_t)2 * desc->B * (K - 1), stream)) return rc;
  }
  return 0;
}

int m3_engine_forward_chunk(m3_engine* e, const m3_stream_desc* desc, void* state, size_t state_bytes, const float* feat_chunk,
                            const int32_t* chunk_feat_len, float* logits, void* workspace, size_t workspace_bytes, int chunk_index,
                            int use_graph, m3_stream stream_) {
  if (int rc = stream_check(e, desc)) return rc;
  const m3_engine_config& c = e->cfg;
  const int C = c.static_chunk_size, T = 4 * C + 3, B = desc->B;
  const size_t need = carve_stream_state(c, nullptr, B, desc->history_frames).bytes;
  M3_REQUIRE(state != nullptr && state_bytes >= need, "engine_forward_chunk: state %zu bytes < required %zu", state_bytes, need);
  M3_REQUIRE(chunk_index >= 0 && (long)(chunk_index + 1) * C <= desc->max_frames,
             "engine_forward_chunk: chunk %d ends past max_frames=%d (the stream is longer than the state was sized for)", chunk_index, desc->max_frames);
  hipStream_t stream = (hipStream_t)stream_;
  if (!e->cur.matches(B, T, feat_chunk, chunk_feat_len, logits, workspace, workspace_bytes, e->ep_capacity, state, desc->history_frames, desc->max_frames)) {
    int rc = prepare_impl(e, feat_chunk, chunk_feat_len, B, T, logits, workspace, workspace_bytes, state, desc->history_frames, desc->max_frames);
    if (rc < 0) return rc;
  }
  e->cur.last_use = ++e->use_clock;
  if (!use_graph) return m3_engine_run(e, 0, (int)e->cur.stages.size(), stream_);
  if (!e->cur.graph_valid) {
    M3_REQUIRE(stream != nullptr, "engine_forward_chunk: graph capture needs a non-default stream");
    if (e->cur.graph_exec) { (void)hipGraphExecDestroy(e->cur.graph_exec); e->cur.graph_exec = nullptr; }
    hipGraph_t graph = nullptr;
    M3_CHECK_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    const int rc = m3_engine_run(e, 0, (int)e->cur.stages.size(), stream_);
    hipError_t ce = hipStreamEndCapture(stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    M3_CHECK_HIP(ce);
    M3_CHECK_HIP(hipGraphInstantiate(&e->cur.graph_exec, graph, nullptr, nullptr, 0));
    M3_CHECK_HIP(hipGraphDestroy(graph));
    e->cur.graph_valid = true;
    ++e->n_captures;
  }
  M3_CHECK_HIP(hipGraphLaunch(e->cur.graph_exec, stream));
  return 0;
}

}  // extern "C"
