"""Host-side mirror of the reference's graph-builder package (TRTAPI++/python/trt_helper/__init__.py:30-34):
same public names, MI355X back end.  ``from trt_helper import trt`` gives the slice of the tensorrt namespace the
model code uses (PluginField, PluginFieldCollection, ...)."""
from . import trt
from .builder_helper import init_trt_plugin, HelperConfig, BuilderHelper
from .network_helper import NetworkHelper, PluginRegistry, PluginCreator
from .infer_helper import InferHelper
from m3asr.calibrate import AsrCalibrator

__all__ = ["trt", "init_trt_plugin", "HelperConfig", "BuilderHelper", "NetworkHelper", "InferHelper",
           "PluginRegistry", "PluginCreator", "AsrCalibrator"]
