"""bf16-storage weight gradient (csrc/wgrad_bf16.hip, C ABI: w2l_conv_wgrad_bf16) against torch autograd in float64 over the SAME
bf16-rounded operands: products of bf16 values are exact in fp32, what is left is the fp32 accumulation over K = pixels in
another order.  Tolerance 2e-4 of the gradient tensor's L-inf scale (the fp32 weight-gradient kernels' tolerance)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_conv_gpu import SIGS
from wav2lip_amd import _lib, bf16
from wav2lip_amd._lib import ACT_NONE, ConvGeom, check, ptr

pytestmark = pytest.mark.gpu


def _pair(v):
    return tuple(v) if isinstance(v, tuple) else (v, v)


def _rb(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _nhwc(x, cs=None, off=0):
    N, Cn, H, W = x.shape
    cs = cs or bf16.round8(Cn)
    buf = torch.zeros((N, H, W, cs), dtype=torch.bfloat16)
    buf[..., off:off + Cn] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    return buf


def _check(cuda, transposed, cin, cout, k, stride, pad, outpad, N, H, W, seed=0, x_extra=0, dz_extra=0):
    torch.manual_seed(seed)
    kh, kw = _pair(k)
    s, p, op = _pair(stride), _pair(pad), _pair(outpad)
    x = torch.randn(N, cin, H, W)
    wshape = (cin, cout, kh, kw) if transposed else (cout, cin, kh, kw)
    w = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(_rb(x), w, None, s, p, op) if transposed else F.conv2d(_rb(x), w, None, s, p)
    dz = torch.randn(y.shape)
    y.backward(_rb(dz))
    ref = w.grad
    g = ConvGeom(int(transposed), cin, cout, kh, kw, s[0], s[1], p[0], p[1], op[0], op[1], ACT_NONE)
    # operands may live in channel slices of wider buffers (the generator's concat buffers): x_extra / dz_extra more channels
    xb = _nhwc(x, bf16.round8(cin) + x_extra).to(cuda)
    dzb = _nhwc(dz, bf16.round8(cout) + dz_extra).to(cuda)
    if x_extra:
        xb[..., bf16.round8(cin):] = 5.0
    if dz_extra:
        dzb[..., bf16.round8(cout):] = -3.0
    dw = torch.full(wshape, 9.0, device=cuda)
    lib = _lib.load()
    check(lib.w2l_conv_wgrad_bf16(C.byref(g), _lib.current_stream(), N, H, W, ptr(xb), xb.shape[-1], ptr(dzb), dzb.shape[-1],
                                  ptr(dw)), "conv_wgrad_bf16")
    torch.cuda.synchronize()
    err = (dw.double().cpu() - ref).abs()
    S = float(ref.abs().max())
    assert float(err.max()) <= 2e-4 * S, "max err %.3e vs scale %.3e (ratio %.2e)" % (float(err.max()), S, float(err.max()) / S)


@pytest.mark.parametrize("idx", range(len(SIGS)))
def test_weight_gradient_signature(idx, cuda):
    """all 58 layer signatures of the hot path, N = 3"""
    kind, k, stride, pad, cin, cout, H, W, residual, outpad = SIGS[idx]
    _check(cuda, kind == "t", cin, cout, k, stride, pad, outpad, 3, H, W, seed=idx)


def test_ragged_boxes_batches_and_channel_slices(cuda):
    _check(cuda, False, 64, 64, 3, 1, 1, 0, 1, 96, 96, seed=1)                   # batch 1
    _check(cuda, False, 64, 64, 3, 1, 1, 0, 5, 13, 11, seed=2)                   # extents no box divides
    _check(cuda, False, 24, 40, 3, 1, 1, 0, 3, 9, 7, seed=3)                     # channel counts that are not multiples of 32
    _check(cuda, False, 80, 32, 3, 1, 1, 0, 2, 20, 20, seed=4)                   # Q of 80 channels: a 64 + 16 slice pair
    _check(cuda, True, 160, 64, 3, 2, 1, 1, 2, 10, 12, seed=5)                   # P of 160 channels: 64 + 64 + 32
    _check(cuda, False, 512, 512, 3, 1, 1, 0, 37, 3, 3, seed=6)                  # several whole images per box, ragged last box
    _check(cuda, False, 512, 512, 1, 1, 0, 0, 50, 1, 1, seed=7)                  # 1x1 on 1x1: K = batch
    _check(cuda, False, 32, 3, 1, 1, 0, 0, 2, 96, 96, seed=8)                    # RGB head: 3 couts
    _check(cuda, False, 64, 64, 3, 1, 1, 0, 2, 24, 24, seed=9, x_extra=96, dz_extra=32)    # operands are slices of wider buffers
    _check(cuda, True, 64, 32, 3, 2, 1, 0, 2, 7, 5, seed=10)                     # transposed without output padding
    _check(cuda, False, 16, 32, 5, (1, 2), (2, 1), 0, 2, 11, 21, seed=11)        # 5x5, anisotropic stride / pad


def test_weight_gradient_is_deterministic(cuda):
    torch.manual_seed(3)
    g = ConvGeom(0, 128, 128, 3, 3, 1, 1, 1, 1, 0, 0, ACT_NONE)
    xb = torch.randn(8, 48, 48, 128, device=cuda).to(torch.bfloat16)
    dzb = torch.randn(8, 48, 48, 128, device=cuda).to(torch.bfloat16)
    lib = _lib.load()
    outs = []
    for _ in range(2):
        dw = torch.empty(128, 128, 3, 3, device=cuda)
        check(lib.w2l_conv_wgrad_bf16(C.byref(g), _lib.current_stream(), 8, 48, 48, ptr(xb), 128, ptr(dzb), 128, ptr(dw)), "wgrad")
        outs.append(dw.clone())
    assert torch.equal(outs[0], outs[1])
