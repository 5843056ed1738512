#!/usr/bin/env python3
"""Build an encoder plan -- same command line as the reference's builder.py (:150-168):

    python3 builder.py -c config.yaml -m checkpoint.pt -o encoder.plan [-prior prior.txt] [-cmvn cmvn] [-f] [-i]

Flow (reference builder.py:100-147, build_trt :36-98): yaml config -> ``model.<nnet_proto>.Net`` -> load_state_dict ->
declare inputs ``feat (-1,-1,idim) f32`` / ``feat_len (1,-1) i32`` + profiles -> ``model.encoder(network_helper, feat,
feat_len)`` -> optional ``+ (-log prior)`` -> markOutput -> build_engine(plan).  The emission runs op-by-op on the GPU on an
opt-shape dummy batch, and build_engine checks the fused engine against it before writing the plan.
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))

import numpy as np
import torch
import yaml

import trt_helper
from trt_helper import trt


def read_prior(prior_file, minimum_prior=None):
    """Prior probabilities with zeros smoothed to the smallest non-zero value (reference builder.py:16-27)."""
    prior = np.loadtxt(prior_file)[1:]
    prior[prior == 0] = prior[prior != 0].min()
    prior = prior / prior.sum()
    return np.maximum(prior, minimum_prior) if minimum_prior is not None else prior


class ConformerConfig(trt_helper.HelperConfig):
    pass


def build_trt(model, args, input_dim, plan_name, prior=None, profile=None):
    logger = trt_helper.init_trt_plugin(trt.Logger.VERBOSE if args.verbose else trt.Logger.INFO, "libm3asr_hip.so")
    cfg = ConformerConfig()
    cfg.max_workspace_size = 8
    if args.fp16:
        cfg.use_fp16, cfg.plugin_data_type = True, trt.DataType.HALF
    if args.fp8:
        cfg.use_fp8 = True                          # e4m3 expert weights + bf16 dense weights (W8A16), no calibration needed
    calibrator = None
    if args.int8:
        # the reference's 8-bit slot (builder.py:43-47 there: plugin_data_type, use_int8 and an AsrCalibrator over lists of
        # .npy feature files, behind an `assert 0`): fp8 arithmetic with the activation scales calibrated on those files
        cfg.use_int8 = True
        calibrator = trt_helper.AsrCalibrator(args.calib_feat_list, args.calib_feat_len_list, args.calib_cache, args.calib_batches)
    builder_helper = trt_helper.BuilderHelper(cfg, logger, calibrator)
    nh = builder_helper.get_network_helper()
    feat = nh.addInput(name="feat", dtype=trt.float32, shape=(-1, -1, input_dim))
    feat_len = nh.addInput(name="feat_len", dtype=trt.int32, shape=(1, -1))
    (min_b, opt_b, max_b), (min_t, opt_t, max_t) = profile or ((1, 4, 6), (1, 500, 6100))   # reference :58-64
    builder_helper.add_profile("feat", (min_b, min_t, input_dim), (opt_b, opt_t, input_dim), (max_b, max_t, input_dim))
    builder_helper.add_profile("feat_len", (1, min_b), (1, opt_b), (1, max_b))
    if args.cmvn_file:                              # global CMVN (the reference accepts the flag but never applies it)
        from m3asr import ops
        from m3asr.plan import read_cmvn_stats
        mean, istd = read_cmvn_stats(args.cmvn_file)
        builder_helper.cmvn = (mean, istd)
        raw = feat.resolve() if hasattr(feat, "resolve") else feat
        nh._bound["feat_raw"] = raw
        feat = nh._bound["feat"] = ops.cmvn(raw, None, mean.to(raw.device), istd.to(raw.device))
    if args.log_softmax:
        model.encoder.cfg.log_softmax_out = True
    res = model.encoder(nh, feat, feat_len)
    if args.log_softmax:                            # score = log_softmax(output)   (reference :77-81, commented out there)
        res = nh.addLog(nh.addSoftmax(res, dim=-1))
    if prior is not None:                           # score = score - log(prior)    (reference :83-88)
        torch_prior = torch.from_numpy(-np.log(prior)).float().view(1, 1, -1)
        builder_helper.output_bias = torch_prior
        res = nh.addAdd(res, nh.addConstant(torch_prior))
    nh.markOutput(res)
    engine = builder_helper.build_engine(plan_name)
    print("=======================bindings shape=====================")
    for i in range(engine.num_bindings):
        print("idx:%d, name: %s, is_input: %s, shape:%s" % (i, engine.get_binding_name(i), engine.binding_is_input(i),
                                                            engine.get_binding_shape(i)))
    print("=======================bindings shape=====================")
    return engine


def main(args):
    with open(args.config, "r") as f:
        configs = yaml.load(f, Loader=yaml.SafeLoader)
    configs["input_dim"] = 40                                           # reference builder.py:124
    nnet_module = importlib.import_module("model." + configs.get("nnet_proto"))
    input_dim, output_dim = configs["input_dim"], configs["output_dim"]
    model = nnet_module.Net(input_dim, output_dim, **configs["model_conf"])
    # checkpoints are tensors-only state_dicts: never unpickle arbitrary objects
    param_dict = torch.load(args.load_path, map_location="cpu", weights_only=True)
    model.load_state_dict(param_dict)
    print("Loading model from {}".format(args.load_path))
    print("model parameter size: {}".format(sum(p.numel() for p in model.parameters())))
    prior = read_prior(args.prior_file) if args.prior_file else None
    profile = None
    if args.opt_shape:
        b, t = (int(v) for v in args.opt_shape.split("x"))
        profile = ((1, b, max(b, 6)), (1, t, max(t, 6100)))
    build_trt(model, args, input_dim, args.output, prior, profile)


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="3M-ASR encoder plan builder (MI355X)")
    p.add_argument("-m", "--load_path", required=True, help="The PyTorch checkpoint file path.")
    p.add_argument("-o", "--output", required=True, help="The plan file to write")
    p.add_argument("-c", "--config", required=True, help="config file")
    p.add_argument("-prior", "--prior_file", required=False, help="prior file")
    p.add_argument("-cmvn", "--cmvn_file", required=False, help="global CMVN stats (Kaldi text matrix or 2xD .npy); fused into the first conv")
    p.add_argument("--log-softmax", dest="log_softmax", action="store_true",
                   help="output log_softmax(logits) (- log prior) instead of raw logits (reference builder.py:77-81)")
    p.add_argument("-f", "--fp16", action="store_true")
    p.add_argument("-i", "--int8", action="store_true")
    p.add_argument("--fp8", action="store_true", help="expert weights as fp8 e4m3 with per-row scales, dense weights bf16")
    p.add_argument("--calib-feat-list", default="np_inputs/np_feat.list", help="--int8: text file, one .npy feature batch per line")
    p.add_argument("--calib-feat-len-list", default="np_inputs/np_feat_len.list", help="--int8: matching .npy length files")
    p.add_argument("--calib-cache", default="conformer.int8.cache", help="--int8: calibration cache (JSON of the scales)")
    p.add_argument("--calib-batches", type=int, default=10)
    p.add_argument("-t", "--strict", action="store_true")
    p.add_argument("-w", "--workspace-size", default=1000, type=int)
    p.add_argument("-tcf", "--timing-cache-file", required=False)
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--opt-shape", default=None, help="BxT of the dummy batch the emission runs on (default 4x500)")
    main(p.parse_args())
