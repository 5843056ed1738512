// Plugin registry + plugin objects behind the generic m3_plugin_* C ABI.
//
// Mirrors the reference's plugin library structure without TensorRT:
//   PluginCreatorRegistry / init_trt_plugin_plus      TRTAPI++/plugin/trt_plugin_plus.cpp:56-165
//   IPluginCreator::createPlugin(PluginFieldCollection) e.g. fmoe_expert_plugin.cpp:331-360
//   IPluginV2DynamicExt::{getOutputDimensions, getWorkspaceSize, enqueue, serialize, clone}
//                                                      e.g. fmoe_expert_plugin.cpp:189-304
// Plugin names, versions, attribute names and input/output orders are the reference's.
// Only LINEAR layout, fp32 data (data_type 0) and int32 indices are implemented in this round; the
// reference itself asserts on HALF in the FMoE plugin (fmoe_expert_plugin.cpp:264-266).
#include <math.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/m3asr.h"
#include "common.h"
#include "kernels.h"

namespace m3 {
int moe_expert_ffn(const float* x, const int32_t* gate_idx, const float* w1, const float* b1, const float* w2,
                   const float* b2, int S, int E, int D, int F, const float* gate_value, const float* resid,
                   float alpha, const float* ln_gamma, const float* ln_beta, float ln_eps, float* y, void* ws,
                   size_t ws_bytes, hipStream_t stream);
}

enum PluginKind {
  K_FMOE = 0, K_SOFTMAX_TOPK, K_ATT_MASKED_SOFTMAX, K_LAYER_NORM, K_MASKED_FILL, K_GLU, K_MASK_CONV2D_SAMPLE,
  K_REL_POS_ENC, K_DUMP_TENSOR, K_CAT_SPLIT_CACHE, K_ATT_STREAM_SOFTMAX, K_COUNT
};
static const char* const kPluginNames[K_COUNT] = {
    "FMoEExpertPluginDynamic",      "SoftmaxTopKPluginDynamic", "AttMaskedSoftmaxPluginDynamic",
    "LayerNormPluginDynamic",       "MaskedFillPluginDynamic",  "GluPluginDynamic",
    "MaskConv2dSamplePluginDynamic", "RelPositionalEncodingPluginDynamic", "DumpTensorPluginDynamic",
    // the streaming operators (built by the reference, cat_split_cache_plugin.h:26-27, att_stream_softmax_plugin.h:26-27)
    "CatSplitCachePluginDynamic",   "AttStreamSoftmaxPluginDynamic"};

// POD attribute block (what serialize() writes, reference: serialize.hpp:36-52)
struct PluginAttrs {
  int32_t kind;
  int32_t data_type;
  int32_t num_expert, idim, hidden_units, act_type;  // FMoE
  float scale;                                       // AttMaskedSoftmax / RelPositionalEncoding
  int32_t dim;                                       // LayerNorm / RelPositionalEncoding
  float eps;                                         // LayerNorm
  float fill;                                        // MaskedFill
  int32_t axis_dim;                                  // Glu
  int32_t left_padding, stride;                      // MaskConv2dSample
  int32_t max_len, streaming;                        // RelPositionalEncoding
  int32_t cache_len;                                 // AttStreamSoftmax (scale above; CatSplitCache uses axis_dim)
};

struct m3_plugin {
  PluginAttrs a;
};

static std::mutex g_registry_mutex;  // the reference guards registry insertions (trt_plugin_plus.cpp:64-66)

static int find_kind(const char* name, const char* version) {
  if (!name || !version || strcmp(version, "1") != 0) return -1;
  for (int k = 0; k < K_COUNT; ++k)
    if (strcmp(name, kPluginNames[k]) == 0) return k;
  return -1;
}

struct FieldReader {
  const m3_field* f;
  int n;
  bool get_i32(const char* name, int32_t* out) const {
    for (int i = 0; i < n; ++i)
      if (f[i].name && strcmp(f[i].name, name) == 0 && f[i].type == M3_FIELD_INT32 && f[i].length >= 1 && f[i].data) {
        *out = *(const int32_t*)f[i].data;
        return true;
      }
    return false;
  }
  bool get_f32(const char* name, float* out) const {
    for (int i = 0; i < n; ++i)
      if (f[i].name && strcmp(f[i].name, name) == 0 && f[i].type == M3_FIELD_FLOAT32 && f[i].length >= 1 && f[i].data) {
        *out = *(const float*)f[i].data;
        return true;
      }
    return false;
  }
};

static int64_t volume(const m3_tensor& t) {
  int64_t v = 1;
  for (int i = 0; i < t.ndim; ++i) v *= t.shape[i];
  return v;
}

extern "C" {

int m3_registry_lookup(const char* plugin_name, const char* plugin_version) {
  std::lock_guard<std::mutex> lock(g_registry_mutex);
  return find_kind(plugin_name, plugin_version) >= 0 ? 1 : 0;
}
int m3_registry_count(void) { return K_COUNT; }
const char* m3_registry_name(int index) { return (index >= 0 && index < K_COUNT) ? kPluginNames[index] : nullptr; }

m3_plugin* m3_plugin_create(const char* plugin_name, const char* plugin_version, const m3_field* fields, int n_fields) {
  const int kind = find_kind(plugin_name, plugin_version);
  if (kind < 0) {
    m3::set_error("plugin_create: no creator for (%s, %s)", plugin_name ? plugin_name : "(null)",
                  plugin_version ? plugin_version : "(null)");
    return nullptr;
  }
  FieldReader r{fields, n_fields};
  PluginAttrs a;
  memset(&a, 0, sizeof(a));
  a.kind = kind;
  a.act_type = 0;
  bool ok = true;
  const bool has_dtype = r.get_i32("data_type", &a.data_type);
  switch (kind) {
    case K_FMOE:
      ok = has_dtype && r.get_i32("num_expert", &a.num_expert) && r.get_i32("idim", &a.idim) &&
           r.get_i32("hidden_units", &a.hidden_units);
      r.get_i32("act_type", &a.act_type);
      ok = ok && a.num_expert > 0 && a.idim > 0 && a.hidden_units > 0;
      break;
    case K_SOFTMAX_TOPK: ok = has_dtype; break;
    case K_ATT_MASKED_SOFTMAX: ok = has_dtype && r.get_f32("scale", &a.scale); break;
    case K_LAYER_NORM: ok = has_dtype && r.get_i32("dim", &a.dim) && r.get_f32("eps", &a.eps) && a.dim > 0; break;
    case K_MASKED_FILL: ok = has_dtype && r.get_f32("fill", &a.fill); break;
    case K_GLU: ok = has_dtype && r.get_i32("axis_dim", &a.axis_dim); break;
    case K_MASK_CONV2D_SAMPLE:
      ok = r.get_i32("left_padding", &a.left_padding) && r.get_i32("stride", &a.stride) && a.stride > 0;
      break;
    case K_REL_POS_ENC:
      ok = has_dtype && r.get_f32("scale", &a.scale) && r.get_i32("max_len", &a.max_len) && r.get_i32("dim", &a.dim);
      r.get_i32("streaming", &a.streaming);
      break;
    case K_DUMP_TENSOR: break;
    case K_CAT_SPLIT_CACHE:      // cat_split_cache_plugin.cpp createPlugin: data_type, axis_dim
      ok = has_dtype && r.get_i32("axis_dim", &a.axis_dim);
      break;
    case K_ATT_STREAM_SOFTMAX:   // att_stream_softmax_plugin.cpp createPlugin: data_type, scale, cache_len
      ok = has_dtype && r.get_f32("scale", &a.scale) && r.get_i32("cache_len", &a.cache_len) && a.cache_len >= 0;
      break;
  }
  if (!ok) {
    m3::set_error("plugin_create(%s): missing or invalid attribute", plugin_name);
    return nullptr;
  }
  if (kind != K_MASK_CONV2D_SAMPLE && kind != K_DUMP_TENSOR && a.data_type != M3_F32) {
    m3::set_error("plugin_create(%s): data_type %d not implemented (fp32 only)", plugin_name, a.data_type);
    return nullptr;
  }
  m3_plugin* p = new m3_plugin;
  p->a = a;
  return p;
}

m3_plugin* m3_plugin_clone(const m3_plugin* plugin) {
  if (!plugin) return nullptr;
  m3_plugin* p = new m3_plugin;
  p->a = plugin->a;
  return p;
}
void m3_plugin_destroy(m3_plugin* plugin) { delete plugin; }
const char* m3_plugin_type(const m3_plugin* plugin) { return plugin ? kPluginNames[plugin->a.kind] : nullptr; }
int m3_plugin_num_outputs(const m3_plugin* plugin) {
  if (!plugin) return 0;
  return (plugin->a.kind == K_SOFTMAX_TOPK || plugin->a.kind == K_REL_POS_ENC || plugin->a.kind == K_CAT_SPLIT_CACHE) ? 2 : 1;
}

static int expect_inputs(const m3_plugin* p, int n_in) {
  static const int kNumInputs[K_COUNT] = {6, 2, 2, 3, 2, 1, 1, 2, 1, 2, 3};
  // RelPositionalEncoding with streaming = 1 takes the frame counter as a third input (rel_positional_encoding_plugin.cpp:93-95)
  const int want = kNumInputs[p->a.kind] + ((p->a.kind == K_REL_POS_ENC && p->a.streaming) ? 1 : 0);
  M3_REQUIRE(n_in == want, "%s: expected %d inputs, got %d", kPluginNames[p->a.kind], want, n_in);
  return 0;
}

int m3_plugin_output_dims(const m3_plugin* plugin, const m3_tensor* in, int n_in, m3_tensor* out, int n_out) {
  M3_REQUIRE(plugin && in && out, "plugin_output_dims: null argument");
  if (int rc = expect_inputs(plugin, n_in)) return rc;
  M3_REQUIRE(n_out == m3_plugin_num_outputs(plugin), "%s: expected %d outputs", kPluginNames[plugin->a.kind],
             m3_plugin_num_outputs(plugin));
  const PluginAttrs& a = plugin->a;
  auto copy_shape = [](m3_tensor& d, const m3_tensor& s) {
    d.ndim = s.ndim;
    d.dtype = s.dtype;
    for (int i = 0; i < 8; ++i) d.shape[i] = s.shape[i];
  };
  switch (a.kind) {
    case K_SOFTMAX_TOPK:  // value (B,T,1) T, idx (B,T,1) i32  (softmax_topk_plugin.cpp:59-72)
      M3_REQUIRE(in[0].ndim == 3, "SoftmaxTopK: logits must be (B,T,E)");
      copy_shape(out[0], in[0]); out[0].shape[2] = 1;
      copy_shape(out[1], in[0]); out[1].shape[2] = 1; out[1].dtype = M3_I32;
      break;
    case K_GLU: {
      const int ax = a.axis_dim < 0 ? a.axis_dim + in[0].ndim : a.axis_dim;
      M3_REQUIRE(ax >= 0 && ax < in[0].ndim && in[0].shape[ax] % 2 == 0, "Glu: bad axis %d", a.axis_dim);
      copy_shape(out[0], in[0]); out[0].shape[ax] = in[0].shape[ax] / 2;
      break;
    }
    case K_REL_POS_ENC:  // [x*scale, pe[:, :T]]  (rel_positional_encoding_plugin.cpp:59-92)
      M3_REQUIRE(in[0].ndim == 3 && in[1].ndim == 3, "RelPositionalEncoding: x (B,T,D), pe (1,max_len,D)");
      copy_shape(out[0], in[0]);
      copy_shape(out[1], in[1]); out[1].shape[1] = in[0].shape[1];
      break;
    case K_CAT_SPLIT_CACHE: {  // inputs (cache, x): output = cat along axis_dim, out_cache like cache  (cat_split_cache_plugin.cpp:152-165)
      const int ax = a.axis_dim < 0 ? in[1].ndim - 1 : a.axis_dim;
      M3_REQUIRE(in[0].ndim == in[1].ndim && in[1].ndim >= 3 && ax > 0 && ax < in[1].ndim,
                 "CatSplitCache: needs >= 3 dims and axis_dim in [1, ndim) (cat_split_cache_plugin.cpp:88-98), got ndim %d axis %d", in[1].ndim, a.axis_dim);
      for (int i = 0; i < in[1].ndim; ++i)
        M3_REQUIRE(i == ax || in[0].shape[i] == in[1].shape[i], "CatSplitCache: cache and input differ in dim %d", i);
      copy_shape(out[0], in[1]); out[0].shape[ax] = in[0].shape[ax] + in[1].shape[ax];
      copy_shape(out[1], in[0]);
      break;
    }
    default: copy_shape(out[0], in[0]); break;
  }
  return 0;
}

size_t m3_plugin_workspace_size(const m3_plugin* plugin, const m3_tensor* in, int n_in, const m3_tensor* out,
                                int n_out) {
  (void)out; (void)n_out;
  if (!plugin || !in || plugin->a.kind != K_FMOE || n_in < 1 || in[0].ndim != 3) return 0;
  const int S = (int)(in[0].shape[0] * in[0].shape[1]);
  return m3_moe_expert_workspace_size(S, plugin->a.num_expert, plugin->a.idim, plugin->a.hidden_units);
}

int m3_plugin_enqueue(m3_plugin* plugin, const m3_tensor* in, int n_in, m3_tensor* out, int n_out, void* workspace,
                      size_t workspace_bytes, m3_stream stream_) {
  M3_REQUIRE(plugin && in && out, "plugin_enqueue: null argument");
  if (int rc = expect_inputs(plugin, n_in)) return rc;
  M3_REQUIRE(n_out == m3_plugin_num_outputs(plugin), "%s: wrong output count", kPluginNames[plugin->a.kind]);
  hipStream_t stream = (hipStream_t)stream_;
  const PluginAttrs& a = plugin->a;
  for (int i = 0; i < n_in; ++i) M3_REQUIRE(in[i].data != nullptr, "%s: input %d is null", kPluginNames[a.kind], i);
  for (int i = 0; i < n_out; ++i) M3_REQUIRE(out[i].data != nullptr, "%s: output %d is null", kPluginNames[a.kind], i);
  switch (a.kind) {
    case K_FMOE: {
      M3_REQUIRE(in[0].ndim == 3 && in[0].shape[2] == a.idim, "FMoEExpert: x must be (B,T,%d)", a.idim);
      M3_REQUIRE(in[1].dtype == M3_I32, "FMoEExpert: gate_idx must be int32");
      M3_REQUIRE(volume(in[2]) == (int64_t)a.num_expert * a.hidden_units * a.idim, "FMoEExpert: w1 must be (E,F,D)");
      M3_REQUIRE(volume(in[4]) == (int64_t)a.num_expert * a.hidden_units * a.idim, "FMoEExpert: w2 must be (E,D,F)");
      const int S = (int)(in[0].shape[0] * in[0].shape[1]);
      return m3::moe_expert_ffn((const float*)in[0].data, (const int32_t*)in[1].data, (const float*)in[2].data,
                                (const float*)in[3].data, (const float*)in[4].data, (const float*)in[5].data, S,
                                a.num_expert, a.idim, a.hidden_units, nullptr, nullptr, 1.f, nullptr, nullptr, 0.f,
                                (float*)out[0].data, workspace, workspace_bytes, stream);
    }
    case K_SOFTMAX_TOPK: {
      M3_REQUIRE(in[0].ndim == 3, "SoftmaxTopK: logits must be (B,T,E)");
      const int B = (int)in[0].shape[0], T = (int)in[0].shape[1], E = (int)in[0].shape[2];
      return m3::launch_softmax_top1((const float*)in[0].data, E, (const int32_t*)in[1].data, T, B * T, E,
                                     (int32_t*)out[1].data, (float*)out[0].data, stream);
    }
    case K_ATT_MASKED_SOFTMAX: {
      M3_REQUIRE(in[0].ndim == 4, "AttMaskedSoftmax: scores must be (B,h,T1,T2)");
      return m3::launch_att_masked_softmax((const float*)in[0].data, (const int32_t*)in[1].data, (int)in[0].shape[0],
                                           (int)in[0].shape[1], (int)in[0].shape[2], (int)in[0].shape[3], a.scale,
                                           (float*)out[0].data, stream);
    }
    case K_LAYER_NORM: {
      M3_REQUIRE(in[0].ndim >= 1 && in[0].shape[in[0].ndim - 1] == a.dim, "LayerNorm: last dim must be %d", a.dim);
      return m3::launch_layernorm((const float*)in[0].data, (const float*)in[1].data, (const float*)in[2].data, a.eps,
                                  (float*)out[0].data, (int)(volume(in[0]) / a.dim), a.dim, stream);
    }
    case K_MASKED_FILL: {
      M3_REQUIRE(in[0].ndim == 3, "MaskedFill: x must be (B,C,T)");
      return m3::launch_masked_fill((const float*)in[0].data, (const int32_t*)in[1].data, (int)in[0].shape[0],
                                    (int)in[0].shape[1], (int)in[0].shape[2], a.fill, (float*)out[0].data, stream);
    }
    case K_GLU: {
      const int ax = a.axis_dim < 0 ? a.axis_dim + in[0].ndim : a.axis_dim;
      M3_REQUIRE(ax >= 0 && ax < in[0].ndim, "Glu: bad axis");
      int64_t outer = 1, inner = 1;
      for (int i = 0; i < ax; ++i) outer *= in[0].shape[i];
      for (int i = ax + 1; i < in[0].ndim; ++i) inner *= in[0].shape[i];
      return m3::launch_glu((const float*)in[0].data, (int)outer, (int)(in[0].shape[ax] / 2), (int)inner,
                            (float*)out[0].data, stream);
    }
    case K_MASK_CONV2D_SAMPLE:
      return m3::launch_mask_conv2d_sample((const int32_t*)in[0].data, (int)volume(in[0]), a.left_padding, a.stride,
                                           (int32_t*)out[0].data, stream);
    case K_REL_POS_ENC: {
      M3_REQUIRE(in[0].ndim == 3 && in[0].shape[2] == a.dim, "RelPositionalEncoding: x must be (B,T,%d)", a.dim);
      M3_REQUIRE(in[0].shape[1] < a.max_len, "RelPositionalEncoding: T'=%lld must be < max_len=%d",
                 (long long)in[0].shape[1], a.max_len);  // rel_positional_encoding_plugin.cpp:139-142
      // one launch for both outputs (the reference's kernel does the same, rel_positional_encoding_kernel.cu:62-69)
      // streaming = 1: pos_emb = pe[off : off + T], off = frame_num[0] (the kernel's stated contract,
      // rel_positional_encoding_kernel.cu:108-111); the offset is bounded by the table on the device side (max_offset)
      const int32_t* frame_num = a.streaming ? (const int32_t*)in[2].data : nullptr;
      if (a.streaming) M3_REQUIRE(in[2].dtype == M3_I32, "RelPositionalEncoding: frame_num must be int32");
      const int pe_len = (int)in[1].shape[1];
      return m3::launch_rel_positional_encoding((const float*)in[0].data, (const float*)in[1].data, pe_len, frame_num,
                                                a.streaming ? pe_len - (int)in[0].shape[1] : 0,
                                                a.scale, (int)in[0].shape[0], (int)in[0].shape[1], a.dim,
                                                (float*)out[0].data, (float*)out[1].data, nullptr, stream);
    }
    case K_CAT_SPLIT_CACHE: {   // cat_split_cache_plugin.cpp:113-150: batch = prod(dims before axis), rows = the rest
      const int ax = a.axis_dim < 0 ? in[1].ndim - 1 : a.axis_dim;
      M3_REQUIRE(in[0].ndim == in[1].ndim && ax > 0 && ax < in[1].ndim, "CatSplitCache: bad axis_dim %d", a.axis_dim);
      M3_REQUIRE(in[0].dtype == in[1].dtype && (in[1].dtype == M3_F32 || in[1].dtype == M3_I32), "CatSplitCache: 4-byte elements only");
      int64_t batch = 1, input_dim = 1, cache_dim = 1;
      for (int i = 0; i < ax; ++i) batch *= in[1].shape[i];
      for (int i = ax; i < in[1].ndim; ++i) { input_dim *= in[1].shape[i]; cache_dim *= in[0].shape[i]; }
      return m3::launch_cat_split_cache(in[0].data, in[1].data, (int)batch, (int)cache_dim, (int)input_dim, out[0].data, out[1].data, stream);
    }
    case K_ATT_STREAM_SOFTMAX: {   // att_stream_softmax_plugin.cpp:95-126: scores (B, h, T, ld), decode_frame_num [B], mask_idx [B]
      M3_REQUIRE(in[0].ndim == 4, "AttStreamSoftmax: scores must be (B,h,T,ld)");
      M3_REQUIRE(in[1].dtype == M3_I32 && in[2].dtype == M3_I32, "AttStreamSoftmax: decode_frame_num / mask_idx must be int32");
      return m3::launch_att_stream_softmax((const float*)in[0].data, (const int32_t*)in[1].data, (const int32_t*)in[2].data,
                                           (int)in[0].shape[0], (int)(in[0].shape[1] * in[0].shape[2]), (int)in[0].shape[3], a.cache_len,
                                           a.scale, (float*)out[0].data, stream);
    }
    case K_DUMP_TENSOR: {
      size_t es = (in[0].dtype == M3_F16 || in[0].dtype == M3_BF16) ? 2 : (in[0].dtype == M3_I8 ? 1 : 4);
      M3_CHECK_HIP(hipMemcpyAsync(out[0].data, in[0].data, (size_t)volume(in[0]) * es, hipMemcpyDeviceToDevice, stream));
      return 0;
    }
  }
  M3_REQUIRE(false, "plugin_enqueue: unknown plugin kind %d", a.kind);
}

size_t m3_plugin_serialization_size(const m3_plugin* plugin) { return plugin ? sizeof(PluginAttrs) : 0; }
int m3_plugin_serialize(const m3_plugin* plugin, void* buffer, size_t bytes) {
  M3_REQUIRE(plugin && buffer && bytes >= sizeof(PluginAttrs), "plugin_serialize: buffer too small");
  memcpy(buffer, &plugin->a, sizeof(PluginAttrs));
  return 0;
}
m3_plugin* m3_plugin_deserialize(const char* plugin_name, const char* plugin_version, const void* buffer, size_t bytes) {
  const int kind = find_kind(plugin_name, plugin_version);
  if (kind < 0 || !buffer || bytes < sizeof(PluginAttrs)) {
    m3::set_error("plugin_deserialize: bad arguments");
    return nullptr;
  }
  m3_plugin* p = new m3_plugin;
  memcpy(&p->a, buffer, sizeof(PluginAttrs));
  if (p->a.kind != kind) {
    m3::set_error("plugin_deserialize: blob is a %s, not a %s", kPluginNames[p->a.kind % K_COUNT], plugin_name);
    delete p;
    return nullptr;
  }
  return p;
}

}  // extern "C"
