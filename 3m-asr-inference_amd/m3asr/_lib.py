"""ctypes binding of libm3asr_hip.so (the C ABI declared in include/m3asr.h).

This is the only way the Python host reaches the device: there is NO CPU / torch fallback.  If the
shared library is missing the import raises, and every wrapper raises RuntimeError with
m3_last_error() on a non-zero status (the reference's Python raises RuntimeError on a null
layer/plugin, TRTAPI++/python/trt_helper/base_network_helper.py:38-55).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# M3ASR_LIB: development override (kernel experiments built next to the tree); the product loads the in-tree library
LIB_PATH = os.environ.get("M3ASR_LIB") or os.path.join(_HERE, "libm3asr_hip.so")


class M3Error(RuntimeError):
    pass


class Tensor(C.Structure):  # m3_tensor
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 8)]


class Field(C.Structure):  # m3_field
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("type", C.c_int32), ("length", C.c_int32)]


class LinearDesc(C.Structure):  # m3_linear_desc
    _fields_ = [("a", C.c_void_p), ("lda", C.c_int32),
                ("a2", C.c_void_p), ("lda2", C.c_int32), ("k1", C.c_int32),
                ("w", C.c_void_p), ("bias", C.c_void_p),
                ("y", C.c_void_p), ("ldy", C.c_int32),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
                ("ln_wsum", C.c_void_p), ("ln_wbeta", C.c_void_p),
                ("len", C.c_void_p), ("rows_per_batch", C.c_int32), ("mask_in", C.c_int32), ("mask_out", C.c_int32),
                ("act", C.c_int32), ("alpha", C.c_float),
                ("resid", C.c_void_p), ("ldr", C.c_int32), ("weight_dtype", C.c_int32),
                ("a_dtype", C.c_int32), ("y_dtype", C.c_int32), ("y_copy_bf16", C.c_void_p), ("ld_copy", C.c_int32),
                ("y_copy_stats", C.c_void_p), ("ln_stats", C.c_void_p), ("ln_stat_parts", C.c_int32)]


class EngineConfig(C.Structure):  # m3_engine_config
    _fields_ = [(n, C.c_int32) for n in (
        "input_dim", "output_dim", "attention_dim", "attention_heads", "num_blocks",
        "embed_dim", "embed_heads", "embed_linear_units", "embed_blocks",
        "num_experts", "hidden_units", "cnn_module_kernel", "cnn_layer_norm", "embed_cnn_layer_norm",
        "router_with_bias", "keep_expert_output", "ep_world_size", "ep_rank", "fold_pos_proj", "debug_taps", "log_softmax_out", "fuse_route",
        "shape_cache", "bf16_activations", "weight_dtype", "packed_rows", "fp8_activations", "ep_stages", "fork_embed", "static_chunk_size", "num_left_chunks", "causal", "embed_causal")]


class StreamDesc(C.Structure):  # m3_stream_desc
    _fields_ = [("B", C.c_int32), ("history_frames", C.c_int32), ("max_frames", C.c_int32)]


class WeightEntry(C.Structure):  # m3_weight_entry
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64), ("dtype", C.c_int32)]


FIELD_FLOAT32, FIELD_INT32 = 1, 5
F32, F16, I8, I32, BF16, FP8 = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_RELU, ACT_SILU, ACT_GLU, ACT_SIGMOID, ACT_LOG = 0, 1, 2, 3, 4, 5
OP_SUM, OP_PROD = 0, 1

_vp, _i, _f, _sz, _i64, _cp = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64, C.c_char_p
_P = C.POINTER

# name -> (restype, argtypes).  Must list every symbol include/m3asr.h declares
# (tests/test_abi.py checks this table against the header and the .so).
class StageInfo(C.Structure):        # m3_stage_info
    _fields_ = [("kernel", C.c_char_p), ("launches", C.c_int32), ("per_row", C.c_int32),
                ("alg_bytes", C.c_double), ("flops", C.c_double)]


SIGNATURES = {
    "m3_abi_version": (_i, []),
    "m3_last_error": (_cp, []),
    "m3_registry_lookup": (_i, [_cp, _cp]),
    "m3_registry_count": (_i, []),
    "m3_registry_name": (_cp, [_i]),
    "m3_plugin_create": (_vp, [_cp, _cp, _P(Field), _i]),
    "m3_plugin_clone": (_vp, [_vp]),
    "m3_plugin_destroy": (None, [_vp]),
    "m3_plugin_type": (_cp, [_vp]),
    "m3_plugin_num_outputs": (_i, [_vp]),
    "m3_plugin_output_dims": (_i, [_vp, _P(Tensor), _i, _P(Tensor), _i]),
    "m3_plugin_workspace_size": (_sz, [_vp, _P(Tensor), _i, _P(Tensor), _i]),
    "m3_plugin_enqueue": (_i, [_vp, _P(Tensor), _i, _P(Tensor), _i, _vp, _sz, _vp]),
    "m3_plugin_serialization_size": (_sz, [_vp]),
    "m3_plugin_serialize": (_i, [_vp, _vp, _sz]),
    "m3_plugin_deserialize": (_vp, [_cp, _cp, _vp, _sz]),
    "m3_moe_scatter_mapping": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "m3_moe_local_scatter": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "m3_moe_local_gather": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "m3_moe_expert_slice": (_i, []),
    "m3_moe_expert_workspace_size": (_sz, [_i, _i, _i, _i]),
    "m3_moe_expert_ffn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _f, _vp,
                               _vp, _sz, _vp]),
    "m3_moe_expert_ffn_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _f, _vp,
                                    _vp, _sz, _vp]),
    "m3_moe_expert_ffn_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _f, _vp,
                                   _vp, _sz, _vp]),
    "m3_moe_expert_ffn_fp8a8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _f, _vp,
                                     _vp, _sz, _vp]),
    "m3_moe_expert_ffn_fp8a8_active": (_i, [_i, _i, _i, _i]),
    "m3_moe_expert_ffn_fp8a8_xq": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp,
                                        _f, _vp, _vp, _sz, _vp]),
    "m3_quantize_rows_e4m3": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "m3_moe_combine": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _i, _i, _vp]),
    "m3_moe_combine_bf16": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _i, _i, _vp]),
    "m3_softmax_top1": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "m3_moe_router": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i, _i, _i, _vp]),
    "m3_linear": (_i, [_P(LinearDesc), _vp]),
    "m3_linear_workspace_size": (_sz, [_P(LinearDesc)]),
    "m3_linear_ws": (_i, [_P(LinearDesc), _vp, _sz, _vp]),
    "m3_conv2d_3x3s2_first": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "m3_conv2d_3x3s2": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "m3_layer_norm": (_i, [_vp, _vp, _vp, _f, _vp, _i, _i, _vp]),
    "m3_relpos_attention": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _vp]),
    "m3_relpos_attention_bf16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp, _i, _vp]),
    "m3_relpos_attention_chunk": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp, _i, _vp]),
    "m3_dwconv_ln_silu": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp]),
    "m3_subsample_conv1": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "m3_subsample_conv1_cmvn": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "m3_cmvn": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "m3_log_softmax_bias": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "m3_subsample_conv2": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "m3_ctc_greedy": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "m3_ctc_topk": (_i, [_vp, _sz, _i, _i, _vp, _vp, _vp]),
    "m3_ctc_prefix_beam_search": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "m3_cat_split_cache": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "m3_att_stream_softmax": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "m3_rel_positional_encoding": (_i, [_vp, _vp, _i, _vp, _i, _f, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "m3_att_masked_softmax": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "m3_masked_fill": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "m3_glu": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "m3_mask_conv2d_sample": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "m3_scale": (_i, [_vp, _f, _vp, _sz, _vp]),
    "m3_unary": (_i, [_vp, _vp, _sz, _i, _vp]),
    "m3_binary": (_i, [_vp, _vp, _vp, _P(_i64), _P(_i64), _P(_i64), _i, _i, _vp]),
    "m3_permute": (_i, [_vp, _vp, _P(_i64), _P(_i64), _i, _vp]),
    "m3_concat_last": (_i, [_vp, _i, _vp, _i, _vp, _sz, _vp]),
    "m3_softmax": (_i, [_vp, _vp, _sz, _i, _vp]),
    "m3_batched_matmul": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _vp]),
    "m3_pad2d": (_i, [_vp, _sz, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "m3_depthwise_conv1d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "m3_engine_create": (_vp, [_P(EngineConfig), _P(WeightEntry), _i]),
    "m3_engine_destroy": (None, [_vp]),
    "m3_engine_output_frames": (_i, [_i]),
    "m3_engine_workspace_size": (_sz, [_vp, _i, _i]),
    "m3_engine_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _i, _vp]),
    "m3_engine_prepare": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz]),
    "m3_engine_chunk_input_frames": (_i, [_vp]),
    "m3_engine_stream_state_size": (_sz, [_vp, _P(StreamDesc)]),
    "m3_engine_stream_reset": (_i, [_vp, _P(StreamDesc), _vp, _sz, _vp]),
    "m3_engine_forward_chunk": (_i, [_vp, _P(StreamDesc), _vp, _sz, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp]),
    "m3_engine_set_ep_capacity": (_i, [_vp, _i]),
    "m3_engine_num_captures": (_i, [_vp]),
    "m3_engine_num_stages": (_i, [_vp]),
    "m3_engine_stage_name": (_cp, [_vp, _i]),
    "m3_engine_run": (_i, [_vp, _i, _i, _vp]),
    "m3_engine_buffer": (_i, [_vp, _cp, _P(_vp), _P(_sz)]),
    "m3_engine_num_kernels": (_i, [_vp]),
    "m3_engine_stage_info": (_i, [_vp, _i, _vp]),
    "m3_ep_send_map": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "m3_ep_recv_gate": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises M3Error if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise M3Error("libm3asr_hip.so not found at %s -- build it with `make -C 3m-asr-inference_amd` "
                      "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/.so out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().m3_last_error().decode()


def check(status, what):
    if status != 0:
        raise M3Error("%s failed (status %d): %s" % (what, status, last_error()))
