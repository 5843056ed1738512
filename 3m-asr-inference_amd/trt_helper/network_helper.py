"""NetworkHelper: the reference's graph-builder API (TRTAPI++/python/trt_helper/{trt,tensor,torch,}_network_helper.py),
executed EAGERLY on the MI355X through libm3asr_hip.so.

The reference emits a TensorRT network; here every ``add*`` call runs its HIP kernel(s) immediately on the current
stream and returns the result tensor, so the reference's L1 model code (``forward(network_helper, ...)``) works
unchanged as an op-by-op executor.  The subset implemented is exactly what the hot-path model code uses
(SURVEY.md §8b); anything else raises ``RuntimeError("... not support!")`` like the reference's own ~90 stubs
(torch_network_helper.py).  Layouts are the reference's (NCHW convs, (B,C,1,T) conv1d, (B,h,T,dk) attention).
"""
import ctypes as C

import numpy as np
import torch

from m3asr import _lib, ops
from m3asr._lib import check
from . import trt


class _Layer:
    """What ``network.add_*`` returns in TensorRT: named, with get_output(i)."""

    def __init__(self, outputs):
        self.outputs = list(outputs)
        self.name = ""

    @property
    def num_outputs(self):
        return len(self.outputs)

    def get_output(self, i):
        return self.outputs[i]


class _ShuffleLayer(_Layer):
    """network.add_shuffle(x): attributes are set after creation (subsampling.py:29-36), evaluated on get_output."""

    def __init__(self, helper, x):
        super().__init__([None])
        self._helper, self._x = helper, x
        self.first_transpose = self.reshape_dims = self.second_transpose = None

    def get_output(self, i):
        if self.outputs[0] is None:
            self.outputs[0] = self._helper._shuffle(self._x, self.first_transpose, self.reshape_dims, self.second_transpose)
        return self.outputs[0]


class InputRef:
    """Network input declared before its data exists (the reference calls addInput before add_profile,
    builder.py:55-71).  Resolved to the bound device tensor by the first op that consumes it."""

    def __init__(self, helper, name, dtype, shape):
        self._helper, self.name, self.dtype, self.decl_shape = helper, name, dtype, tuple(shape)

    def resolve(self):
        h = self._helper
        if self.name not in h._bound and getattr(h, "_builder", None) is not None:
            h._builder._materialise()
        if self.name not in h._bound:
            raise RuntimeError("input '%s' has no data: bind_input() or add_profile() first" % self.name)
        t = h._bound[self.name]
        if len(self.decl_shape) != t.dim() or any(s != -1 and s != d for s, d in zip(self.decl_shape, t.shape)):
            raise RuntimeError("input '%s': bound tensor %s does not match declared shape %s" % (
                self.name, tuple(t.shape), self.decl_shape))
        h.inputs[self.name] = t
        return t

    @property
    def shape(self):
        return self.resolve().shape


class Plugin:
    """IPluginV2DynamicExt stand-in: owns an m3_plugin handle."""

    def __init__(self, handle, name):
        self.handle, self.plugin_type = handle, name

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            _lib.load().m3_plugin_destroy(h)


class PluginCreator:
    def __init__(self, name, version):
        self.name, self.plugin_version = name, version

    def create_plugin(self, name, pfc):
        """-> Plugin, or None on bad/missing attributes (reference creators return nullptr,
        fmoe_expert_plugin.cpp:356-359; the model code then raises RuntimeError)."""
        lib = _lib.load()
        arr = (_lib.Field * max(len(pfc), 1))()
        keep = []
        for i, f in enumerate(pfc):
            data = np.ascontiguousarray(f.data)
            keep.append(data)
            arr[i].name = f.name.encode()
            arr[i].data = data.ctypes.data
            arr[i].type = f.type
            arr[i].length = data.size
        h = lib.m3_plugin_create(self.name.encode(), self.plugin_version.encode(), arr, len(pfc))
        return Plugin(h, self.name) if h else None


class PluginRegistry:
    """trt.get_plugin_registry() stand-in (registry lives in the C library, plugins.hip)."""

    def get_plugin_creator(self, name, version, namespace=""):
        if _lib.load().m3_registry_lookup(name.encode(), version.encode()):
            return PluginCreator(name, version)
        return None


_DT = {torch.float32: _lib.F32, torch.int32: _lib.I32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16,
       torch.int8: _lib.I8}
_TORCH_DT = {v: k for k, v in _DT.items()}


def _desc(t):
    d = _lib.Tensor()
    d.data = t.data_ptr() if t is not None else None
    d.dtype, d.ndim = _DT[t.dtype], t.dim()
    for i, s in enumerate(t.shape):
        d.shape[i] = s
    return d


class _Network:
    """The part of trt.INetworkDefinition the model code calls directly."""

    def __init__(self, helper):
        self._h = helper
        self.num_layers = 0

    def add_plugin_v2(self, inputs, plugin):
        lib = _lib.load()
        if plugin is None or not plugin.handle:
            raise RuntimeError("add_plugin_v2: null plugin")
        ins = [self._h._dev(t) for t in inputs]
        n_in, n_out = len(ins), lib.m3_plugin_num_outputs(plugin.handle)
        in_d = (_lib.Tensor * n_in)(*[_desc(t) for t in ins])
        out_d = (_lib.Tensor * n_out)()
        check(lib.m3_plugin_output_dims(plugin.handle, in_d, n_in, out_d, n_out), "m3_plugin_output_dims")
        outs = []
        for i in range(n_out):
            shape = [out_d[i].shape[k] for k in range(out_d[i].ndim)]
            t = torch.empty(shape, dtype=_TORCH_DT[out_d[i].dtype], device=ins[0].device)
            out_d[i].data = t.data_ptr()
            outs.append(t)
        ws_bytes = lib.m3_plugin_workspace_size(plugin.handle, in_d, n_in, out_d, n_out)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=ins[0].device)
        check(lib.m3_plugin_enqueue(plugin.handle, in_d, n_in, out_d, n_out, ws.data_ptr(), ws_bytes,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "m3_plugin_enqueue(%s)" % plugin.plugin_type)
        self.num_layers += 1
        return _Layer(outs)

    def add_shuffle(self, x):
        self.num_layers += 1
        return _ShuffleLayer(self._h, x)

    def add_padding(self, x, pre_padding, post_padding):
        """trt.INetworkDefinition.add_padding (IPaddingLayer: zeros on the last two dims): the causal ConvolutionModule pads
        lorder = K - 1 frames on the left of its (B,C,1,T) input (convolution.py:118-123)."""
        self.num_layers += 1
        return _Layer([ops.pad2d(self._h._dev(x).contiguous(), tuple(pre_padding), tuple(post_padding))])

    def mark_output(self, x):
        self._h.outputs.append(x)


class NetworkHelper:
    def __init__(self, network=None, plugin_registry=None, config=None, logger=None, device="cuda:0"):
        self.device = torch.device(device)
        self.network = network if network is not None else _Network(self)
        self.plugin_registry = plugin_registry if plugin_registry is not None else PluginRegistry()
        self.config = config
        self.logger = logger if logger is not None else trt.Logger(trt.Logger.WARNING)
        self.input_num = 0
        self.inputs, self.outputs = {}, []
        self._bound = {}
        self._const_cache = {}

    # ---- plumbing -------------------------------------------------------------------------
    def bind_input(self, name, tensor):
        """Eager execution needs data: bind a device tensor to a network input before emission."""
        self._bound[name] = tensor.to(self.device).contiguous()

    def _dev(self, t):
        """Weights arrive as CPU tensors/parameters (addConstant / layer.weight): upload once, keyed by storage."""
        if t is None:
            return None
        if isinstance(t, InputRef):
            return t.resolve()
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)
        t = t.detach()
        if t.is_cuda:
            return t.contiguous()
        key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)
        hit = self._const_cache.get(key)
        if hit is None:
            hit = t.to(self.device).contiguous()
            self._const_cache[key] = hit
        return hit

    def set_layer_name(self, layer, name):
        if not layer:
            raise RuntimeError("Could not name")
        layer.name = str(self.network.num_layers) + "_" + name
        for i in range(layer.num_outputs):
            self.logger.log(trt.Logger.VERBOSE, "[Network] %s, output[%d] shape= %s" % (
                layer.name, i, tuple(layer.get_output(i).shape)))

    # ---- trt_network_helper.py -------------------------------------------------------------
    def addInput(self, name, dtype, shape):
        if name is None:
            name = "input" + str(self.input_num)
        self.input_num += 1
        ref = InputRef(self, name, dtype, shape)
        if getattr(self, "_builder", None) is not None:
            self._builder.declare_input(name, dtype, shape)
        return ref.resolve() if name in self._bound else ref

    def markOutput(self, x):
        self.network.mark_output(x)

    def addConstant(self, x, layer_name=None):
        return self._dev(x)

    def _shuffle(self, x, first, reshape, second):
        """TensorRT IShuffleLayer: transpose -> reshape (0 copies that dim, -1 infers) -> transpose."""
        x = self._dev(x)
        if first is not None:
            x = ops.permute_copy(x, tuple(first)) if tuple(first) != tuple(range(x.dim())) else x
        if reshape is not None:
            dims = [x.shape[i] if d == 0 else d for i, d in enumerate(reshape)]
            x = x.reshape(dims)
        if second is not None and tuple(second) != tuple(range(x.dim())):
            x = ops.permute_copy(x, tuple(second))
        return x

    def addShuffle(self, x, first_transpose, reshape_dims, second_transpose, layer_name=None):
        return self._shuffle(x, first_transpose, reshape_dims, second_transpose)

    # ---- tensor_network_helper.py -------------------------------------------------------------
    def addCat(self, inputs, dim=0, layer_name=None, precision=None):
        assert len(inputs) > 1
        if dim == -1:
            dim = inputs[0].dim() - 1
        if len(inputs) != 2 or dim != inputs[0].dim() - 1:
            raise RuntimeError("torch.cat on dim %d with %d inputs not support!" % (dim, len(inputs)))
        return ops.concat_last(inputs[0], inputs[1])

    def addMatMul(self, a, b, layer_name=None, precision=None):
        """Same rank: a @ b; otherwise a @ b^T (tensor_network_helper.py:268-282)."""
        return ops.batched_matmul(a, b, transpose_b=(a.dim() != b.dim()))

    def addAdd(self, a, b, layer_name=None, precision=None):
        return ops.binary(a, self._dev(b), _lib.OP_SUM)

    def addProd(self, a, b, layer_name=None, precision=None):
        return ops.binary(a, self._dev(b), _lib.OP_PROD)

    def addScale(self, x, scale, layer_name=None, precision=None):
        if x.dim() < 3:
            raise RuntimeError("input_len < 3 not support now! ")
        return ops.scale(x, scale)

    # ---- torch_network_helper.py -------------------------------------------------------------
    def addLinear(self, layer, x, layer_name=None, precision=None):
        w = self._dev(layer.weight)
        b = self._dev(layer.bias) if layer.bias is not None else None
        y = ops.linear(x.reshape(-1, x.shape[-1]), w, b)
        return y.view(tuple(x.shape[:-1]) + (w.shape[0],))

    def addLayerNorm(self, layer, x, layer_name=None, precision=None):
        return ops.layer_norm(x, self._dev(layer.weight), self._dev(layer.bias), layer.eps)

    def addConv2d(self, layer, x, layer_name=None, precision=None):
        """NCHW in/out, the convolution alone (the ReLU that follows it in Conv2dSubsampling4 is a separate addReLU in the
        emission, as in the reference; the fused engine uses the conv + ReLU kernels).  Implemented for what
        Conv2dSubsampling4 emits: 3x3, stride 2, no padding, groups 1."""
        if tuple(layer.kernel_size) != (3, 3) or tuple(layer.stride) != (2, 2) or tuple(layer.padding) != (0, 0) \
                or layer.groups != 1 or tuple(layer.dilation) != (1, 1):
            raise RuntimeError("nn.Conv2d other than 3x3/stride 2/no padding not support!")
        w, b = layer.weight.detach(), self._dev(layer.bias)
        O, I = w.shape[0], w.shape[1]
        if I == 1:
            key = ("c1", w.data_ptr())
            wp = self._const_cache.get(key)
            if wp is None:
                wp = self._const_cache[key] = w.reshape(O, 9).t().contiguous().to(self.device)
            y = ops.subsample_conv1(x.reshape(x.shape[0], x.shape[2], x.shape[3]).contiguous(), wp, b, act=_lib.ACT_NONE)
        else:
            key = ("c2", w.data_ptr())
            wp = self._const_cache.get(key)
            if wp is None:
                wp = self._const_cache[key] = w.permute(0, 2, 3, 1).contiguous().to(self.device)
            y = ops.subsample_conv2(ops.permute_copy(x, (0, 2, 3, 1)), wp, b, act=_lib.ACT_NONE)
        return ops.permute_copy(y, (0, 3, 1, 2))

    def addConv1d(self, layer, x, layer_name=None, precision=None):
        """x is (B,C,1,T) (torch_network_helper.py:199-225).  Pointwise (k=1) and depthwise (groups=C) only."""
        B, Cin, one, T = x.shape
        k = layer.kernel_size[0]
        w, b = layer.weight.detach(), (self._dev(layer.bias) if layer.bias is not None else None)
        if k == 1 and layer.groups == 1:
            rows = ops.permute_copy(x.reshape(B, Cin, T), (0, 2, 1)).reshape(B * T, Cin)
            y = ops.linear(rows, self._dev(w.reshape(w.shape[0], Cin)), b)
            return ops.permute_copy(y.view(B, T, -1), (0, 2, 1)).reshape(B, -1, 1, T)
        if layer.groups == Cin and w.shape[1] == 1 and layer.stride[0] == 1 and layer.dilation[0] == 1:
            y = ops.depthwise_conv1d(x.reshape(B, Cin, T).contiguous(), self._dev(w), b, layer.padding[0])
            return y.reshape(B, Cin, 1, y.shape[-1])       # (the causal module convolves its own left padding away: padding 0)
        raise RuntimeError("nn.Conv1d with kernel %d / groups %d not support!" % (k, layer.groups))

    def addGLU(self, x, axis_dim=-1, layer_name=None, precision=None):
        return ops.glu(x, axis_dim)

    def addReLU(self, x, layer_name=None, precision=None):
        return ops.unary(x, _lib.ACT_RELU)

    def addSiLU(self, x, layer_name=None, precision=None):
        return ops.unary(x, _lib.ACT_SILU)

    def addSigmoid(self, x, layer_name=None, precision=None):
        return ops.unary(x, _lib.ACT_SIGMOID)

    def addLog(self, x, layer_name=None, precision=None):
        return ops.unary(x, _lib.ACT_LOG)

    def addSoftmax(self, x, dim=-1, layer_name=None, precision=None):
        if dim not in (-1, x.dim() - 1):
            raise RuntimeError("softmax on dim %d not support!" % dim)
        return ops.softmax_lastdim(x)

    # ---- network_helper.py (plugin conveniences) ----------------------------------------------
    def addCatSplitCache(self, cache, x, dim, layer_name=None, precision=None):
        """plugin CatSplitCache (TRTAPI++/python/trt_helper/network_helper.py:80-105): [cat([cache, x], dim), new cache = the
        last cache.shape[dim] entries of it along dim]."""
        plg_creator = self.plugin_registry.get_plugin_creator("CatSplitCachePluginDynamic", "1", "")
        if not plg_creator:
            raise RuntimeError("Could not find CatSplitCachePluginDynamic")
        data_type = trt.PluginField("data_type", np.array([getattr(self.config, "plugin_data_type", 0)], dtype=np.int32), trt.PluginFieldType.INT32)
        axis = trt.PluginField("axis_dim", np.array([dim], dtype=np.int32), trt.PluginFieldType.INT32)
        plugin = plg_creator.create_plugin("CatSplitCachePluginDynamic", trt.PluginFieldCollection([data_type, axis]))
        if not plugin:
            raise RuntimeError("Could not create_plugin CatSplitCachePluginDynamic")
        layer = self.network.add_plugin_v2([cache, x], plugin)
        self.set_layer_name(layer, "CatSplitCachePlugin" if layer_name is None else "CatSplitCachePlugin." + layer_name)
        return [layer.get_output(0), layer.get_output(1)]


    def addDumpTensor(self, x, layer_name=None):
        """DumpTensor plugin (dump_tensor_plugin.cpp:79-130): identity + print."""
        t = x.detach().float().cpu()
        print("[DumpTensor] %s shape=%s sum=%.6f" % (layer_name or "", tuple(t.shape), float(t.sum())))
        return x

    def __getattr__(self, name):
        if name.startswith("add"):
            def _unsupported(*a, **k):
                raise RuntimeError(name[3:] + " not support!")
            return _unsupported
        raise AttributeError(name)
