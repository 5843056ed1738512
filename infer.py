#!/usr/bin/env python3
"""Run an encoder plan -- same command line as the reference's infer.py (:130-137):

    python3 infer.py -p encoder.plan -i feat.npy [-o compare.npy]

feat.npy is (B,T,idim) float32; feat_len = feat.shape[1] for every utterance, as in the reference (infer.py:111-113).
Prints ``time=...ms`` for one forward after a warm-up and the output's shape / sum (reference :81-103)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "3m-asr-inference_amd"))

import numpy as np

import trt_helper
from trt_helper import trt


def main(args):
    logger = trt_helper.init_trt_plugin(trt.Logger.INFO, "libm3asr_hip.so")
    feat = np.load(args.input_file).astype(np.float32)
    feat_len = np.full((1, feat.shape[0]), feat.shape[1], dtype=np.int32)
    helper = trt_helper.InferHelper(args.plan_name, logger)
    base = [np.load(args.compare_output_file)] if args.compare_output_file else None
    outputs = helper.infer([feat, feat_len], base)
    for o in outputs:
        print("outputs.shape:" + str(o.shape))
        print("outputs.sum:" + str(o.sum()))
        print(o)
    if base is not None:
        print("compare_output=%s, dtype=%s, shape=%s" % (args.compare_output_file, base[0].dtype, base[0].shape))
        print("output.sum:" + str(base[0].sum()))


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="3M-ASR encoder inference (MI355X)")
    p.add_argument("-p", "--plan_name", required=True, help="The plan file path.")
    p.add_argument("-i", "--input_file", required=True, help="The input feat.npy file path.")
    p.add_argument("-o", "--compare_output_file", required=False, help="The compare output .npy file path.")
    main(p.parse_args())
