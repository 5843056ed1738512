"""Per-operator tests of the network_helper.add* surface, one torch.nn / torch op each -- the shape of the reference's own
(unshipped) unit tests, TRTAPI++/python/run_test.sh:3-37 (`tests/test_nn_linear.py`, `test_nn_conv1d.py`, `test_nn_conv2d.py`,
`test_nn_layer_norm.py`, `test_nn_softmax.py`, `test_nn_relu.py`, `test_nn_sigmoid.py`, `test_nn_silu.py`, `test_torch_matmul.py` ...)
with their comparison rule `torch.allclose(base, out, rtol=1e-05, atol=1e-03)` (infer_helper.py:93).  Every op goes through
the C ABI of libm3asr_hip.so; the expected value is the torch op on the CPU."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import trt_helper
from trt_helper import trt

RTOL, ATOL = 1e-5, 1e-3          # the reference's InferHelper tolerance


@pytest.fixture(scope="module")
def nh():
    logger = trt_helper.init_trt_plugin(trt.Logger.INFO, "libm3asr_hip.so")
    return trt_helper.NetworkHelper(None, None, trt_helper.HelperConfig(), logger)


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def same(got, want):
    assert tuple(got.shape) == tuple(want.shape), (got.shape, want.shape)
    assert torch.allclose(got.cpu(), want, rtol=RTOL, atol=ATOL), float((got.cpu() - want).abs().max())


def test_nn_linear(nh):
    torch.manual_seed(0)
    for bias in (True, False):
        layer = nn.Linear(80, 48, bias=bias)
        x = rnd(2, 7, 80)
        same(nh.addLinear(layer, x.cuda()), layer(x).detach())


def test_nn_layer_norm(nh):
    torch.manual_seed(1)
    layer = nn.LayerNorm(96, eps=1e-12)
    with torch.no_grad():
        layer.weight.uniform_(0.5, 1.5)
        layer.bias.uniform_(-0.2, 0.2)
    x = rnd(3, 5, 96) * 3 + 1
    same(nh.addLayerNorm(layer, x.cuda()), layer(x).detach())


def test_nn_conv2d_subsampling_shapes(nh):
    torch.manual_seed(2)
    c1, c2 = nn.Conv2d(1, 32, 3, 2), nn.Conv2d(32, 32, 3, 2)
    x = rnd(2, 1, 61, 40)
    y1 = nh.addConv2d(c1, x.cuda())
    same(y1, c1(x).detach())
    same(nh.addConv2d(c2, y1), c2(c1(x)).detach())
    with pytest.raises(RuntimeError):
        nh.addConv2d(nn.Conv2d(1, 8, 5, 1), x.cuda())              # "... not support!" like the reference's stubs


def test_nn_conv1d_pointwise_and_depthwise(nh):
    torch.manual_seed(3)
    x = rnd(2, 64, 1, 37)                                           # (B, C, 1, T) as torch_network_helper.py:199-225
    pw = nn.Conv1d(64, 128, 1)
    same(nh.addConv1d(pw, x.cuda()), pw(x.squeeze(2)).unsqueeze(2).detach())
    dw = nn.Conv1d(64, 64, 15, padding=7, groups=64)
    same(nh.addConv1d(dw, x.cuda()), dw(x.squeeze(2)).unsqueeze(2).detach())
    with pytest.raises(RuntimeError):
        nh.addConv1d(nn.Conv1d(64, 64, 3, padding=1), x.cuda())


@pytest.mark.parametrize("name,fn", [("relu", F.relu), ("silu", F.silu), ("sigmoid", torch.sigmoid)])
def test_nn_activations(nh, name, fn):
    x = rnd(2, 9, 33, seed=4) * 3
    op = {"relu": nh.addReLU, "silu": nh.addSiLU, "sigmoid": nh.addSigmoid}[name]
    same(op(x.cuda()), fn(x))


def test_nn_log_and_softmax_and_log_softmax(nh):
    x = rnd(3, 11, 50, seed=5)
    sm = nh.addSoftmax(x.cuda(), dim=-1)
    same(sm, F.softmax(x, -1))
    same(nh.addLog(sm), F.log_softmax(x, -1))                        # builder.py:77-81 composes log(softmax(x))
    with pytest.raises(RuntimeError):
        nh.addSoftmax(x.cuda(), dim=0)


def test_nn_glu(nh):
    x = rnd(2, 13, 64, seed=6)
    same(nh.addGLU(x.cuda(), -1), F.glu(x, -1))
    xc = rnd(2, 64, 1, 13, seed=7)                                    # conv module layout: GLU over channels
    same(nh.addGLU(xc.cuda(), 1), F.glu(xc, 1))


def test_torch_matmul_add_prod_scale_cat(nh):
    a, b = rnd(2, 4, 9, 16, seed=8), rnd(2, 4, 16, 9, seed=9)
    same(nh.addMatMul(a.cuda(), b.cuda()), torch.matmul(a, b))
    w = rnd(20, 16, seed=10)
    same(nh.addMatMul(a.cuda(), w.cuda()), torch.matmul(a, w.t()))   # rank mismatch: a @ b^T (tensor_network_helper.py:268-282)
    x, y = rnd(2, 5, 8, seed=11), rnd(2, 5, 8, seed=12)
    same(nh.addAdd(x.cuda(), y.cuda()), x + y)
    same(nh.addProd(x.cuda(), y.cuda()), x * y)
    same(nh.addAdd(x.cuda(), rnd(1, 1, 8, seed=13)), x + rnd(1, 1, 8, seed=13))       # broadcast constant
    same(nh.addScale(x.cuda(), 0.125), x * 0.125)
    same(nh.addCat([x.cuda(), y.cuda()], dim=-1), torch.cat([x, y], -1))
    with pytest.raises(RuntimeError):
        nh.addCat([x.cuda(), y.cuda()], dim=0)


def test_shuffle_transpose_reshape(nh):
    x = rnd(2, 7, 4, 16, seed=14)
    got = nh.addShuffle(x.cuda(), (0, 2, 1, 3), (2, 4, 7 * 16), None)
    same(got, x.permute(0, 2, 1, 3).reshape(2, 4, 7 * 16))
    got = nh.addShuffle(x.cuda(), None, (2, 7, 64), (0, 2, 1))
    same(got, x.reshape(2, 7, 64).permute(0, 2, 1))
