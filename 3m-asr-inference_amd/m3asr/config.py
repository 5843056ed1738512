"""Encoder configuration for the 3M-ASR Conformer-MoE hot path.

The reference ships no yaml (SURVEY.md §8 "Model dimensions"); the defaults below are the
values derived there.  Field names follow the reference's constructor arguments
(trainer_3m_fix/model/conformer_fmoe_localComm_catEmbed_domain_acc_hier.py:31-60) so that a
reference yaml's ``model_conf.encoder_conf`` maps onto this class 1:1 via ``from_reference_conf``.
"""
from dataclasses import dataclass, field, asdict
import json


@dataclass
class EncoderConfig:
    input_dim: int = 40            # builder.py:124
    output_dim: int = 1434         # builder.sh:8 (prior file name) - weak evidence, configurable
    attention_dim: int = 512       # README.md:221
    attention_heads: int = 8       # assumed; class default 4
    num_blocks: int = 18           # README.md:5
    cnn_module_kernel: int = 15
    cnn_module_norm: str = "layer_norm"
    # embed encoder (conformer_embed_domain_acc.py), defaults ...hier.py:69-95
    embed_heads: int = 4
    embed_dim: int = 512
    embed_linear_units: int = 1024
    embed_blocks: int = 6
    embed_cnn_module_norm: str = "layer_norm"
    # moe_conf (...hier.py:98-113)
    num_experts: int = 32
    hidden_units: int = 1024
    router_with_bias: bool = False
    keep_expert_output: bool = False
    # expert parallelism (reference: moe_conf rank/world_size, inference default 1)
    ep_world_size: int = 1
    ep_rank: int = 0
    max_len: int = 5000            # positional_encoding.py:31
    # storage of the GEMM weights: "f32" (exact fp32 MFMA) or "bf16" (bf16 MFMA, fp32 accumulate; activations stay
    # fp32) = the reference's --fp16 / plugin_data_type 1 (builder.py:160, builder_helper.py:47-57); "fp8" = bf16 dense
    # weights + e4m3 expert weights with per-row scales, dequantised to bf16 at the MFMA input (the --int8 slot of the reference)
    weight_dtype: str = "f32"
    # weight_dtype "fp8" only: fp8 ARITHMETIC (e4m3 x e4m3 MFMA) in the grouped expert FFN of long batches -- rows quantised
    # with a per-row dynamic scale, H with the calibrated per-layer scale "blocks.N.feed_forward.experts.h_scale" of the
    # plan (m3asr/calibrate.py) -- the reference's --int8 slot (builder.py:39-49, builder_helper.py:109-123)
    fp8_activations: bool = False
    log_softmax_out: bool = False  # output log_softmax(logits) (+ output_bias) instead of raw logits (builder.py:77-88)
    # static chunk mask of the streaming encoders (model/conformer.py:40,142-175 -> utils/mask.py:127-134): with
    # static_chunk_size > 0 every attention (embed and main encoder) sees, for query frame i, the keys of its own chunk and
    # of num_decoding_left_chunks chunks to the left (< 0: all of them) -- besides the padding mask.  0 = full context.
    static_chunk_size: int = 0
    num_decoding_left_chunks: int = -1
    # causal ConvolutionModule (layer/convolution.py:43-49,118-123; constructor argument `causal` of the main encoder,
    # ...domain_acc_hier.py:55,179, and embed_conf['causal'] of the embed encoder, conformer_embed_domain_acc.py:51,127):
    # lorder = kernel - 1 frames padded on the left in front of pointwise_conv1, depthwise conv without padding.
    # Together with static_chunk_size > 0 this is the configuration that can be decoded chunk by chunk (Engine.streaming).
    causal: bool = False
    embed_causal: bool = False

    def fp8_label(self):
        """What the fp8 mode of this config computes in (for reports: a weight-only mode must not read as fp8 MFMA)."""
        if self.fp8_activations:
            return ("fp8 arithmetic (e4m3 weights x e4m3 activations, v_mfma_f32_32x32x16_fp8_fp8) in the grouped expert FFN where "
                    "the fused fp8 kernel applies, e4m3 weight-only (W8A16) elsewhere")
        return "fp8 e4m3 weight-only (W8A16: dequantised to bf16 at the MFMA input, bf16 MFMA)"

    @property
    def d_k(self):
        return self.attention_dim // self.attention_heads

    @property
    def sub_freq(self):
        """Frequency bins left after the two stride-2 3x3 convs (subsampling.py:94)."""
        return ((self.input_dim - 1) // 2 - 1) // 2

    def to_json(self):
        return json.dumps(asdict(self), sort_keys=True)

    @staticmethod
    def from_json(s):
        return EncoderConfig(**json.loads(s))

    @staticmethod
    def tiny(**kw):
        """Small full-structure model used for fixtures (SURVEY §8c fixture plan (i))."""
        base = dict(output_dim=16, attention_dim=32, attention_heads=2, num_blocks=2,
                    embed_heads=2, embed_dim=32, embed_linear_units=64, embed_blocks=2,
                    num_experts=4, hidden_units=64)
        base.update(kw)
        return EncoderConfig(**base)

    @staticmethod
    def from_reference_conf(input_dim, output_dim, encoder_conf):
        """Map a reference yaml ``model_conf['encoder_conf']`` dict onto EncoderConfig."""
        ec = dict(encoder_conf or {})
        emb = dict(ec.get("embed_conf") or {})
        moe = dict(ec.get("moe_conf") or {})
        return EncoderConfig(
            input_dim=input_dim, output_dim=output_dim,
            attention_dim=ec.get("attention_dim", 256),
            attention_heads=ec.get("attention_heads", 4),
            num_blocks=ec.get("num_blocks", 6),
            cnn_module_kernel=ec.get("cnn_module_kernel", 15),
            cnn_module_norm=ec.get("cnn_module_norm", "batch_norm"),
            embed_heads=emb.get("attention_heads", 4),
            embed_dim=emb.get("attention_dim", 512),
            embed_linear_units=emb.get("linear_units", 1024),
            embed_blocks=emb.get("num_blocks", 6),
            embed_cnn_module_norm=emb.get("cnn_module_norm", "batch_norm"),
            num_experts=moe.get("num_experts", 4),
            hidden_units=moe.get("hidden_units", 1024),
            router_with_bias=moe.get("router_with_bias", False),
            keep_expert_output=moe.get("keep_expert_output", False),
            ep_world_size=moe.get("world_size", 1), ep_rank=moe.get("rank", 0),
            static_chunk_size=ec.get("static_chunk_size", 0),
            causal=bool(ec.get("causal", False)), embed_causal=bool(emb.get("causal", False)),
        )


def subsampled_len(t):
    """T -> T' of Conv2dSubsampling4: two (l-3)//2+1 steps
    (mask_conv2d_sample_kernel.cu:34-35 with left_padding=2, stride=2)."""
    return ((t - 3) // 2 + 1 - 3) // 2 + 1
