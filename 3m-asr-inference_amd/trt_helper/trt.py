"""The slice of the ``tensorrt`` Python namespace that the reference's model / driver code touches
(``import tensorrt as trt`` in builder.py:13, infer.py:8 and every trainer_3m_fix/layer/*.py), backed by
libm3asr_hip.so instead of TensorRT.  Reference-side model code binds to it with
``from trt_helper import trt``."""
import numpy as np
import torch

ITensor = torch.Tensor          # tensors flowing through network_helper are device torch tensors
float32 = torch.float32
int32 = torch.int32
float16 = torch.float16


class DataType:                 # values = HelperConfig.plugin_data_type (builder_helper.py:47-57)
    FLOAT, HALF, INT8, INT32 = 0, 1, 2, 3


class PluginFieldType:          # values of nvinfer1::PluginFieldType
    FLOAT16, FLOAT32, FLOAT64, INT8, INT16, INT32, CHAR, DIMS = range(8)


class PluginField:
    def __init__(self, name, data, type=PluginFieldType.FLOAT32):
        self.name, self.data, self.type = name, np.ascontiguousarray(data), type


class PluginFieldCollection(list):
    pass


class Logger:
    INTERNAL_ERROR, ERROR, WARNING, INFO, VERBOSE = range(5)

    def __init__(self, min_severity=INFO):
        self.min_severity = min_severity

    def log(self, severity, msg):
        if severity <= self.min_severity:
            print(msg)


class Permutation(tuple):
    pass


class Dims(tuple):
    pass


def volume(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n
