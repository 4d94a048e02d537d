"""Every distinct fused-conv signature on the hot path against the oracle's
block() on CPU, called through the C ABI (w2l_conv_create / w2l_conv_forward).  fp32, tolerance 1e-4 (abs + rel):
the kernel is an exact-fp32 fma chain, only the summation order differs from oneDNN's."""
import numpy as np
import pytest
import torch

from oracle import models_ref
from wav2lip_amd.models.conv import Conv2d, Conv2dTranspose, nonorm_Conv2d

pytestmark = pytest.mark.gpu

# (kind, k, stride, pad, cin, cout, H, W, residual, outpad)   kind: c conv+BN+ReLU, t convT+BN+ReLU, n conv+LReLU
SIGS = [
    ("c", 3, 1, 1, 1, 32, 80, 16, 0, 0), ("c", 3, 1, 1, 32, 32, 80, 16, 1, 0), ("c", 3, (3, 1), 1, 32, 64, 80, 16, 0, 0),
    ("c", 3, 1, 1, 64, 64, 27, 16, 1, 0), ("c", 3, 3, 1, 64, 128, 27, 16, 0, 0), ("c", 3, 1, 1, 128, 128, 9, 6, 1, 0),
    ("c", 3, (3, 2), 1, 128, 256, 9, 6, 0, 0), ("c", 3, 1, 1, 256, 256, 3, 3, 1, 0), ("c", 3, 1, 0, 256, 512, 3, 3, 0, 0),
    ("c", 1, 1, 0, 512, 512, 1, 1, 0, 0), ("c", 7, 1, 3, 6, 16, 96, 96, 0, 0), ("c", 3, 2, 1, 16, 32, 96, 96, 0, 0),
    ("c", 3, 1, 1, 32, 32, 48, 48, 1, 0), ("c", 3, 2, 1, 32, 64, 48, 48, 0, 0), ("c", 3, 1, 1, 64, 64, 24, 24, 1, 0),
    ("c", 3, 2, 1, 64, 128, 24, 24, 0, 0), ("c", 3, 1, 1, 128, 128, 12, 12, 1, 0), ("c", 3, 2, 1, 128, 256, 12, 12, 0, 0),
    ("c", 3, 1, 1, 256, 256, 6, 6, 1, 0), ("c", 3, 2, 1, 256, 512, 6, 6, 0, 0), ("c", 3, 1, 1, 512, 512, 3, 3, 1, 0),
    ("c", 3, 1, 0, 512, 512, 3, 3, 0, 0), ("t", 3, 1, 0, 1024, 512, 1, 1, 0, 0), ("t", 3, 2, 1, 1024, 512, 3, 3, 0, 1),
    ("c", 3, 1, 1, 512, 512, 6, 6, 1, 0), ("t", 3, 2, 1, 768, 384, 6, 6, 0, 1), ("c", 3, 1, 1, 384, 384, 12, 12, 1, 0),
    ("t", 3, 2, 1, 512, 256, 12, 12, 0, 1), ("c", 3, 1, 1, 256, 256, 24, 24, 1, 0), ("t", 3, 2, 1, 320, 128, 24, 24, 0, 1),
    ("c", 3, 1, 1, 128, 128, 48, 48, 1, 0), ("t", 3, 2, 1, 160, 64, 48, 48, 0, 1), ("c", 3, 1, 1, 64, 64, 96, 96, 1, 0),
    ("c", 3, 1, 1, 80, 32, 96, 96, 0, 0),
    ("c", 7, 1, 3, 15, 32, 48, 96, 0, 0), ("c", 5, (1, 2), 1, 32, 64, 48, 96, 0, 0), ("c", 3, 1, 1, 64, 64, 46, 47, 1, 0),
    ("c", 3, 2, 1, 64, 128, 46, 47, 0, 0), ("c", 3, 1, 1, 128, 128, 23, 24, 1, 0), ("c", 3, 2, 1, 128, 256, 23, 24, 0, 0),
    ("c", 3, 1, 1, 256, 256, 12, 12, 1, 0), ("c", 3, 2, 1, 256, 512, 12, 12, 0, 0), ("c", 3, 2, 1, 512, 512, 6, 6, 0, 0),
    ("n", 7, 1, 3, 3, 32, 48, 96, 0, 0), ("n", 5, (1, 2), 2, 32, 64, 48, 96, 0, 0), ("n", 5, 1, 2, 64, 64, 48, 48, 0, 0),
    ("n", 5, 2, 2, 64, 128, 48, 48, 0, 0), ("n", 5, 1, 2, 128, 128, 24, 24, 0, 0), ("n", 5, 2, 2, 128, 256, 24, 24, 0, 0),
    ("n", 5, 1, 2, 256, 256, 12, 12, 0, 0), ("n", 3, 2, 1, 256, 512, 12, 12, 0, 0), ("n", 3, 1, 1, 512, 512, 6, 6, 0, 0),
    ("n", 3, 2, 1, 512, 512, 6, 6, 0, 0), ("n", 3, 1, 1, 512, 512, 3, 3, 0, 0), ("n", 3, 1, 0, 512, 512, 3, 3, 0, 0),
    ("n", 1, 1, 0, 512, 512, 1, 1, 0, 0),
]


def _geom_string(kind, k, stride, pad, residual, outpad):
    s = stride if isinstance(stride, tuple) else (stride, stride)
    g = "k%ds%dx%dp%d" % (k, s[0], s[1], pad)
    if residual:
        g += "r"
    if kind == "t":
        g += "T" + ("o%d" % outpad if outpad else "")
    return g


def _make(kind, k, stride, pad, cin, cout, residual, outpad, seed):
    torch.manual_seed(seed)
    if kind == "t":
        m = Conv2dTranspose(cin, cout, kernel_size=k, stride=stride, padding=pad, output_padding=outpad)
    elif kind == "n":
        m = nonorm_Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad)
    else:
        m = Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, residual=bool(residual))
    if kind != "n":   # non-trivial BN statistics
        bn = m.conv_block[1]
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
    return m.eval()


def _check(sig, N, cuda, tile=None, seed=0):
    kind, k, stride, pad, cin, cout, H, W, residual, outpad = sig
    m = _make(kind, k, stride, pad, cin, cout, residual, outpad, seed)
    x = torch.randn(N, cin, H, W)
    sd = {"b." + key: v for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", _geom_string(kind, k, stride, pad, residual, outpad), norm=(kind != "n"))
    mg = m.to(cuda)
    if tile is not None:
        mg.fused().set_tile(tile)
    y = mg(x.to(cuda)).cpu()
    assert y.shape == ref.shape, (y.shape, ref.shape)
    err = (y - ref).abs()
    tol = 1e-4 + 1e-4 * ref.abs()
    assert bool((err <= tol).all()), "max err %.3e at |ref| up to %.3e" % (err.max().item(), ref.abs().max().item())


@pytest.mark.parametrize("idx", range(len(SIGS)))
def test_signature(idx, cuda):
    _check(SIGS[idx], 3, cuda, seed=idx)


@pytest.mark.parametrize("tile", range(23))
@pytest.mark.parametrize("idx", [1, 10, 12, 22, 23, 29, 35, 44])
def test_every_tile_config(idx, tile, cuda):
    """each tile configuration must give the same answer on ragged M / cout (not only the auto-picked one)"""
    _check(SIGS[idx], 2, cuda, tile=tile, seed=100 + idx)


def test_batch_one_and_odd_batch(cuda):
    _check(SIGS[32], 1, cuda)     # conv3x3 64->64 @96x96, M = 9216
    _check(SIGS[21], 5, cuda)     # 3x3 valid -> 1x1, M = 5 (one ragged tile)
    _check(SIGS[9], 1, cuda)      # 1x1 on 1x1, M = 1


def test_channel_sliced_io_and_residual_alias(cuda):
    """reads/writes through channel slices of wider NHWC buffers, as the concat-free plan does"""
    from wav2lip_amd import engine
    torch.manual_seed(3)
    m = _make("c", 3, 1, 1, 32, 32, 1, 0, 7).to(cuda)
    layer = m.fused()
    N, H, W = 2, 10, 12
    src = torch.randn(N, H, W, 48, device=cuda)
    dst = torch.full((N, H, W, 40), 7.0, device=cuda)
    a_in = engine.Act(src, 16, 32)
    a_out = engine.Act(dst, 4, 32)
    layer.forward_raw(N, H, W, a_in.ptr, a_in.cs, a_out.ptr, a_out.cs, a_in.ptr, a_in.cs)
    x = src[..., 16:48].permute(0, 3, 1, 2).contiguous().cpu()
    sd = {"b." + key: v.cpu() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", "k3p1r")
    got = dst[..., 4:36].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 1e-4
    assert bool((dst[..., :4] == 7.0).all()) and bool((dst[..., 36:] == 7.0).all()), "wrote outside its slice"


def test_bad_arguments_raise(cuda):
    m = _make("c", 3, 1, 1, 32, 32, 0, 0, 1).to(cuda)
    layer = m.fused()
    x = torch.zeros(1, 4, 4, 30, device=cuda)   # channel stride not a multiple of 4 / too small
    y = torch.zeros(1, 4, 4, 32, device=cuda)
    from wav2lip_amd import engine
    with pytest.raises(RuntimeError, match="x_cs"):
        layer.forward_raw(1, 4, 4, engine.ptr(x), 30, engine.ptr(y), 32)


def _plan_check(sig, N, cuda, tile, ksplit, autotune=False, seed=0, family=None):
    """one-launch plan with a forced (tile, split-K) configuration, or autotuned"""
    from wav2lip_amd import engine
    kind, k, stride, pad, cin, cout, H, W, residual, outpad = sig
    m = _make(kind, k, stride, pad, cin, cout, residual, outpad, seed)
    x = torch.randn(N, cin, H, W)
    sd = {"b." + key: v for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", _geom_string(kind, k, stride, pad, residual, outpad), norm=(kind != "n"))
    layer = m.to(cuda).fused()
    cin_p = layer.cin_p
    xin = torch.zeros(N, H, W, cin_p, device=cuda)
    xin[..., :cin] = x.permute(0, 2, 3, 1).to(cuda)
    ho, wo = layer.out_hw(H, W)
    y = torch.full((N, ho, wo, cout), 3.0, device=cuda)
    plan = engine.Plan()
    a_in = engine.Act(xin, 0, cin_p)
    plan.add("l", layer, a_in, engine.Act(y, 0, cout), a_in if residual else None)
    if autotune:
        plan.autotune()
    else:
        plan.tuned = True
        plan.set_config(0, tile, ksplit)
    if family is not None:   # the forced configuration must be the kernel that runs (an ineligible id falls back silently)
        assert plan.resolved()[0][2] == family, plan.resolved()
    plan.run()
    got = y.permute(0, 3, 1, 2).cpu()
    err = (got - ref).abs()
    assert bool((err <= 1e-4 + 1e-4 * ref.abs()).all()), "max err %.3e (tile %s ksplit %s)" % (err.max().item(), tile, ksplit)
    return plan


@pytest.mark.parametrize("ksplit", [2, 3, 8, 16])
@pytest.mark.parametrize("idx,tile", [(21, 3), (21, 5), (22, 3), (23, 2), (20, 3), (9, 5), (8, 3), (43, 1)])
def test_split_k_matches(idx, tile, ksplit, cuda):
    """split-K partial sums + reduce kernel == single pass (incl. transposed phases with unequal K and tiny M)"""
    _plan_check(SIGS[idx], 2, cuda, tile, ksplit, seed=300 + idx)


@pytest.mark.parametrize("ksplit", [1, 4])
@pytest.mark.parametrize("tile", range(6))
@pytest.mark.parametrize("idx", [1, 10, 12, 22, 23, 25, 29, 31, 35, 44, 55])
def test_split_operand_implicit_gemm(idx, tile, ksplit, cuda):
    """conv_igemm_bf16_kernel<.., 3> (configuration ids 13..18: the implicit-GEMM tiles with each fp32 operand as three bf16
    pieces, six bf16 MFMAs per K-chunk) gives an fp32 result: the same tolerance against the oracle as the fp32 kernels, on
    conv / strided / transposed / 7x7 / 1x1 / residual signatures, ragged M and cout, with and without split-K; the forced
    configuration must be the kernel that runs"""
    from wav2lip_amd import _lib
    lib = _lib.load()
    sid = lib.w2l_conv_num_tiles() - 10 + tile         # the six ids in front of the last four (conv_wino2s, conv_tp2s, conv_stem7s, conv_k3s)
    assert lib.w2l_conv_config_family(sid) == 5
    plan = _plan_check(SIGS[idx], 2, cuda, sid, ksplit, seed=900 + idx, family="split")
    assert plan.resolved()[0][3][0] == sid


def test_split_operand_kernel_is_as_accurate_as_the_fp32_kernel(cuda):
    """K = 4608 (3x3 on 512 channels): against an fp64 convolution the split kernel's error is not larger than 1.5x the fp32
    implicit GEMM's (tools/split_bf16_accuracy.py: the accumulator is rounded 6 times per 16 K instead of 8)"""
    from wav2lip_amd import engine, _lib
    lib = _lib.load()
    torch.manual_seed(77)
    m = _make("c", 3, 1, 1, 512, 512, 0, 0, 77)
    N, H, W = 2, 12, 12
    x = torch.randn(N, 512, H, W)
    with torch.no_grad():
        conv, bn = m.conv_block[0], m.conv_block[1]
        z = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        sc = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        ref = torch.relu((z - bn.running_mean.double()[None, :, None, None]) * sc[None, :, None, None] + bn.bias.double()[None, :, None, None])
    layer = m.to(cuda).fused()
    xin = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    errs = {}
    for name, tile in (("fp32", 0), ("split", lib.w2l_conv_num_tiles() - 10)):
        y = torch.zeros(N, H, W, 512, device=cuda)
        plan = engine.Plan()
        plan.add("l", layer, engine.Act(xin, 0, 512), engine.Act(y, 0, 512), None)
        plan.tuned = True
        plan.set_config(0, tile, 1)
        plan.run()
        errs[name] = (y.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item()
    assert errs["split"] <= 1.5 * errs["fp32"] + 1e-7, errs


def test_split_k_more_splits_than_steps(cuda):
    _plan_check(SIGS[0], 2, cuda, 4, 16, seed=5)     # 3x3 on 1 channel: K = 36 -> 2 steps only


@pytest.mark.parametrize("idx", [9, 21, 23, 25, 32])
def test_autotuned_plan_matches(idx, cuda):
    plan = _plan_check(SIGS[idx], 4, cuda, None, None, autotune=True, seed=400 + idx)
    (name, tile, ks), = plan.configs()
    # any configuration id the library knows may win the stopwatch (the bound used to be a literal 11 from the time there were 11
    # ids: the quarter-split F(2x2) shape, id 12, winning on signature 32 failed the suite once in round 4)
    from wav2lip_amd import _lib
    assert 0 <= tile < _lib.load().w2l_conv_num_tiles() and ks >= 1


@pytest.mark.parametrize("idx", [12, 23])
def test_layer_tune_key_is_the_key_the_library_files_the_launch_under(idx, cuda):
    """FusedConv.tune_key (tools/split_sweep.py --emit-table writes table entries with it) == the key w2l_plan_autotune stores its
    winner under, residual flag and shape included"""
    from wav2lip_amd import _lib
    lib = _lib.load()
    nk = lib.w2l_tune_key_ints()
    before = {tuple(e[:nk]) for e in _lib.export_tune_table(lib)}
    try:
        plan = _plan_check(SIGS[idx], 3, cuda, None, None, autotune=True, seed=950 + idx)
        (_, layer, N, H, W), = plan.records
        key = layer.tune_key(N, H, W, has_res=plan.has_res[0])
        after = {tuple(e[:nk]) for e in _lib.export_tune_table(lib)}
        assert key in after and (key in before or after - before == {key}), (key, after - before)
    finally:
        lib.w2l_tune_clear()
        _lib.load_tune_table(lib)


def _head_ref(m, conv1x1, x, geom):
    sd = {"b." + key: v for key, v in m.state_dict().items()}
    with torch.no_grad():
        h = models_ref.block(x, sd, "b", geom)
        return torch.sigmoid(torch.nn.functional.conv2d(h, conv1x1.weight, conv1x1.bias))


@pytest.mark.parametrize("tile", [None, 0, 1, 2, 3, 4, 5, 9, 12])     # 9 / 12: conv_wino2.hip shapes with the head in their last pass
@pytest.mark.parametrize("cin,cout,hc,H,W", [(80, 32, 3, 20, 24), (16, 64, 1, 9, 7), (8, 128, 4, 5, 5), (80, 32, 3, 96, 96)])
def test_fused_1x1_head(cin, cout, hc, H, W, tile, cuda):
    """conv3x3+BN+ReLU -> 1x1 conv -> sigmoid as one launch (w2l_conv_attach_head) == the oracle's two ops"""
    from wav2lip_amd import engine
    from wav2lip_amd.models.conv import HeadFusedBlock
    from wav2lip_amd._lib import ACT_SIGMOID
    m = _make("c", 3, 1, 1, cin, cout, 0, 0, 11)
    torch.manual_seed(12)
    conv1 = torch.nn.Conv2d(cout, hc, 1)
    N = 3
    x = torch.randn(N, cin, H, W)
    ref = _head_ref(m, conv1, x, "k3p1")
    m, conv1 = m.to(cuda), conv1.to(cuda)
    layer = HeadFusedBlock(m, conv1, ACT_SIGMOID).fused()
    assert layer.cout == hc and layer.macs(N, H, W) == N * H * W * (cin * cout * 9 + cout * hc)
    if tile is not None:
        layer.set_tile(tile)
    xin = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    y = torch.full((N, H, W, 5), 2.0, device=cuda)        # channel stride 5: the head writes scalars
    layer.forward_raw(N, H, W, engine.ptr(xin), cin, engine.ptr(y), 5)
    got = y[..., :hc].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 2e-6, (got - ref).abs().max()
    assert bool((y[..., hc:] == 2.0).all()), "wrote outside its channels"


@pytest.mark.parametrize("tile", [9, 12])
@pytest.mark.parametrize("N,H,W", [(3, 20, 24), (2, 96, 96), (5, 7, 9)])
def test_fused_1x1_head_on_the_winograd_shapes(N, H, W, tile, cuda):
    """the generator's output block (80 -> 32 conv3x3 + BN + ReLU, 32 -> 3 conv1x1, sigmoid) as the plan runs it - output buffer
    with a channel stride of 4 - on conv_wino2's two 32-cout shapes; the forced configuration must be the kernel that runs"""
    from wav2lip_amd import engine
    from wav2lip_amd.models.conv import HeadFusedBlock
    from wav2lip_amd._lib import ACT_SIGMOID
    m = _make("c", 3, 1, 1, 80, 32, 0, 0, 13)
    torch.manual_seed(14)
    conv1 = torch.nn.Conv2d(32, 3, 1)
    x = torch.randn(N, 80, H, W)
    ref = _head_ref(m, conv1, x, "k3p1")
    m, conv1 = m.to(cuda), conv1.to(cuda)
    layer = HeadFusedBlock(m, conv1, ACT_SIGMOID).fused()
    xin = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    y = torch.full((N, H, W, 4), 2.0, device=cuda)
    plan = engine.Plan()
    plan.add("l", layer, engine.Act(xin, 0, 80), engine.Act(y, 0, 3), None)
    plan.tuned = True
    plan.set_config(0, tile, 1)
    assert plan.resolved()[0][2] == "wino2" and plan.resolved()[0][3][0] == tile, plan.resolved()
    plan.run()
    got = y[..., :3].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 2e-6, (got - ref).abs().max()
    assert bool((y[..., 3:] == 2.0).all()), "wrote outside its channels"


def test_head_rejects_residual_and_bad_cout(cuda):
    from wav2lip_amd import engine
    from wav2lip_amd.models.conv import HeadFusedBlock
    from wav2lip_amd._lib import ACT_SIGMOID
    m = _make("c", 3, 1, 1, 8, 30, 0, 0, 1).to(cuda)
    with pytest.raises(RuntimeError, match="cout"):
        HeadFusedBlock(m, torch.nn.Conv2d(30, 3, 1).to(cuda), ACT_SIGMOID).fused()
    m = _make("c", 3, 1, 1, 32, 32, 0, 0, 1).to(cuda)
    layer = HeadFusedBlock(m, torch.nn.Conv2d(32, 3, 1).to(cuda), ACT_SIGMOID).fused()
    x = torch.zeros(1, 4, 4, 32, device=cuda)
    y = torch.zeros(1, 4, 4, 4, device=cuda)
    with pytest.raises(RuntimeError, match="residual"):
        layer.forward_raw(1, 4, 4, engine.ptr(x), 32, engine.ptr(y), 4, engine.ptr(x), 32)


@pytest.mark.parametrize("k,cin,cout,H,W,y_cs,yoff", [(7, 6, 16, 96, 96, 80, 64), (7, 6, 16, 10, 12, 16, 0),
                                                      (3, 4, 8, 9, 8, 8, 0), (5, 3, 12, 7, 6, 20, 4),
                                                      (7, 6, 16, 9, 7, 16, 0)])
def test_x_paired_small_cout_conv(k, cin, cout, H, W, y_cs, yoff, cuda):
    """cout <= 16, x-stride 1: two adjacent output pixels per GEMM row (N = 2*cout fills the 32-wide MFMA tile);
    odd widths fall back to the generic variant; both write through channel slices"""
    from wav2lip_amd import engine
    m = _make("c", k, 1, k // 2, cin, cout, 0, 0, 21)
    N = 2
    x = torch.randn(N, cin, H, W)
    sd = {"b." + key: v for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", "k%dp%d" % (k, k // 2))
    layer = m.to(cuda).fused()
    xin = torch.zeros(N, H, W, layer.cin_p, device=cuda)
    xin[..., :cin] = x.permute(0, 2, 3, 1).to(cuda)
    y = torch.full((N, H, W, y_cs), 4.0, device=cuda)
    dst = engine.Act(y, yoff, cout)
    layer.forward_raw(N, H, W, engine.ptr(xin), layer.cin_p, dst.ptr, y_cs)
    got = y[..., yoff:yoff + cout].permute(0, 3, 1, 2).cpu()
    err = (got - ref).abs()
    assert bool((err <= 1e-4 + 1e-4 * ref.abs()).all()), err.max()
    mask = torch.ones(y_cs, dtype=torch.bool)
    mask[yoff:yoff + cout] = False
    assert bool((y[..., mask.to(cuda)] == 4.0).all()), "wrote outside its slice"


# (cin, cout, H, W, residual)
WINO_SIGS = [(64, 64, 96, 96, 1), (128, 128, 48, 48, 1), (256, 256, 24, 24, 1), (384, 384, 12, 12, 1),
             (64, 64, 46, 47, 1), (128, 128, 23, 24, 1), (512, 512, 3, 3, 1), (64, 128, 9, 5, 0), (80, 64, 7, 8, 0),
             (8, 64, 5, 4, 0), (16, 128, 1, 1, 0)]


@pytest.mark.parametrize("cfg", [0, 1])
@pytest.mark.parametrize("idx", range(len(WINO_SIGS)))
def test_winograd_f2x2_matches_oracle(idx, cfg, cuda):
    """3x3 s1 p1 layers on the Winograd F(2x2,3x3) kernel (configuration ids 6, 7) == oracle, incl. odd extents,
    ragged tile blocks, single-pixel images and channel counts that only one configuration accepts"""
    cin, cout, H, W, res = WINO_SIGS[idx]
    ks, bc = ((8, 64), (16, 128))[cfg]
    if cin % ks or cout % bc:
        pytest.skip("configuration %d does not take %d->%d channels (falls back, covered elsewhere)" % (cfg, cin, cout))
    sig = ("c", 3, 1, 1, cin, cout, H, W, res, 0)
    _plan_check(sig, 3 if H * W > 100 else 5, cuda, 6 + cfg, 1, seed=500 + idx)


WINO2_EXTRA = [(32, 32, 48, 48, 1), (32, 32, 80, 16, 1), (80, 32, 20, 24, 0), (64, 96, 9, 5, 0), (80, 32, 96, 96, 0)]


@pytest.mark.parametrize("cfg", [8, 9])
@pytest.mark.parametrize("N", [1, 3, 9])
@pytest.mark.parametrize("idx", range(len(WINO_SIGS) + len(WINO2_EXTRA)))
def test_winograd_second_generation_matches_oracle(idx, N, cfg, cuda):
    """conv_wino2.hip (configuration ids 8: 32 tiles x 64 couts, two workgroups per CU; 9: 64 tiles x 32 couts; position-split
    waves, the input block staged once through LDS) == oracle on every Winograd shape: all tile-block geometries the host
    picks (8x8x1 ... 1x1x32), ragged blocks at odd extents, image groups that run past the batch, single-pixel images"""
    cin, cout, H, W, res = (WINO_SIGS + WINO2_EXTRA)[idx]
    if cin % 8 or cout % (64 if cfg == 8 else 32):
        pytest.skip("%d->%d channels do not fit configuration %d (falls back, covered elsewhere)" % (cin, cout, cfg))
    if N == 9 and H * W > 3000:
        N = 4
    _plan_check(("c", 3, 1, 1, cin, cout, H, W, res, 0), N, cuda, cfg, 1, seed=600 + idx, family="wino2")


def test_winograd_second_generation_leaky_no_norm_and_slices(cuda):
    """nonorm_Conv2d (LeakyReLU, no BN) 512->512 3x3 layers of the discriminator on configuration 8, and channel-sliced
    input / output / aliasing residual as the concat-free plan uses them"""
    from wav2lip_amd import engine
    _plan_check(("n", 3, 1, 1, 512, 512, 6, 6, 0, 0), 5, cuda, 8, 1, seed=31)
    _plan_check(("n", 3, 1, 1, 512, 512, 3, 3, 0, 0), 7, cuda, 8, 1, seed=32)
    _plan_check(("n", 3, 1, 1, 512, 512, 6, 6, 0, 0), 5, cuda, 9, 1, seed=33)
    m = _make("c", 3, 1, 1, 64, 64, 1, 0, 78).to(cuda)
    layer = m.fused()
    layer.set_tile(8)
    N, H, W = 2, 10, 13
    src = torch.randn(N, H, W, 96, device=cuda)
    dst = torch.full((N, H, W, 80), 7.0, device=cuda)
    a_in, a_out = engine.Act(src, 32, 64), engine.Act(dst, 8, 64)
    layer.forward_raw(N, H, W, a_in.ptr, a_in.cs, a_out.ptr, a_out.cs, a_in.ptr, a_in.cs)
    x = src[..., 32:96].permute(0, 3, 1, 2).contiguous().cpu()
    sd = {"b." + key: v.cpu() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", "k3p1r")
    got = dst[..., 8:72].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 1e-4
    assert bool((dst[..., :8] == 7.0).all()) and bool((dst[..., 72:] == 7.0).all()), "wrote outside its slice"


WINO4_EXTRA = [(64, 64, 13, 11, 1), (64, 64, 4, 4, 0), (72, 64, 33, 35, 0), (64, 192, 8, 8, 0), (64, 64, 2, 3, 1), (128, 64, 6, 6, 0)]


@pytest.mark.parametrize("N", [1, 3, 9])
@pytest.mark.parametrize("idx", range(len(WINO_SIGS) + len(WINO4_EXTRA)))
def test_winograd_f4x4_matches_oracle(idx, N, cuda):
    """conv_wino4.hip (configuration id 11: F(4x4,3x3), 36 positions split 2 x 2 over the waves, four partial inverse
    transforms meeting in LDS) == oracle: every tile-block geometry the host picks (4x8x1 ... 1x1x21), extents that are not
    multiples of 4 (ragged tiles masked on store), image groups running past the batch, residual, single-pixel images"""
    cin, cout, H, W, res = (WINO_SIGS + WINO4_EXTRA)[idx]
    if cin % 8 or cout % 64:
        pytest.skip("%d->%d channels do not fit configuration 11 (falls back, covered elsewhere)" % (cin, cout))
    if N == 9 and H * W > 3000:
        N = 4
    _plan_check(("c", 3, 1, 1, cin, cout, H, W, res, 0), N, cuda, 11, 1, seed=800 + idx, family="wino4")


def test_winograd_f4x4_leaky_no_norm_and_slices(cuda):
    """nonorm_Conv2d (LeakyReLU, no BN) 512->512 on configuration 11, and channel-sliced input / output / aliasing residual"""
    from wav2lip_amd import engine
    _plan_check(("n", 3, 1, 1, 512, 512, 6, 6, 0, 0), 5, cuda, 11, 1, seed=41, family="wino4")
    _plan_check(("n", 3, 1, 1, 512, 512, 3, 3, 0, 0), 7, cuda, 11, 1, seed=42, family="wino4")
    m = _make("c", 3, 1, 1, 64, 64, 1, 0, 79).to(cuda)
    layer = m.fused()
    layer.set_tile(11)
    N, H, W = 2, 10, 13
    src = torch.randn(N, H, W, 96, device=cuda)
    dst = torch.full((N, H, W, 80), 7.0, device=cuda)
    a_in, a_out = engine.Act(src, 32, 64), engine.Act(dst, 8, 64)
    layer.forward_raw(N, H, W, a_in.ptr, a_in.cs, a_out.ptr, a_out.cs, a_in.ptr, a_in.cs)
    x = src[..., 32:96].permute(0, 3, 1, 2).contiguous().cpu()
    sd = {"b." + key: v.cpu() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", "k3p1r")
    got = dst[..., 8:72].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 1e-4
    assert bool((dst[..., :8] == 7.0).all()) and bool((dst[..., 72:] == 7.0).all()), "wrote outside its slice"


def test_winograd_f4x4_data_gradient_form(cuda):
    """the transposed 3x3 s1 p1 layer (the data gradient of a conv) on configuration 11: flipped kernel, swapped channel roles"""
    _plan_check(("t", 3, 1, 1, 64, 128, 12, 12, 0, 0), 3, cuda, 11, 1, seed=43, family="wino4")


@pytest.mark.parametrize("N", [1, 3, 9])
@pytest.mark.parametrize("idx", range(len(WINO_SIGS) + len(WINO2_EXTRA)))
def test_winograd_quarter_split_matches_oracle(idx, N, cuda):
    """conv_wino2q (configuration id 12: 32 tiles x 32 couts, the 16 positions cut 2 x 2 over the four waves, four partial
    inverse transforms meeting in LDS, two workgroups per CU) == oracle on every Winograd shape: all tile-block geometries,
    ragged blocks at odd extents, image groups running past the batch, residual, single-pixel images"""
    cin, cout, H, W, res = (WINO_SIGS + WINO2_EXTRA)[idx]
    if cin % 8 or cout % 32:
        pytest.skip("%d->%d channels do not fit configuration 12 (falls back, covered elsewhere)" % (cin, cout))
    if N == 9 and H * W > 3000:
        N = 4
    _plan_check(("c", 3, 1, 1, cin, cout, H, W, res, 0), N, cuda, 12, 1, seed=900 + idx, family="wino2")


def test_winograd_quarter_split_leaky_no_norm(cuda):
    _plan_check(("n", 3, 1, 1, 512, 512, 6, 6, 0, 0), 5, cuda, 12, 1, seed=51, family="wino2")


TP2_SIGS = [(1024, 512, 3, 3), (768, 384, 6, 6), (512, 256, 12, 12), (320, 128, 24, 24), (160, 64, 48, 48), (8, 64, 5, 7),
            (16, 128, 1, 1), (64, 64, 9, 16)]


@pytest.mark.parametrize("N", [1, 3, 7])
@pytest.mark.parametrize("idx", range(len(TP2_SIGS)))
def test_fused_phase_transposed_conv_matches_oracle(idx, N, cuda):
    """conv_tp2.hip (configuration id 10: ConvTranspose2d(k3, s2, p1, op1) + BN + ReLU with the four output phases in one
    workgroup, the input block staged once through LDS) == oracle on the generator's five upsampling layers and on odd /
    single-pixel inputs: every pixel-block geometry the host picks, ragged blocks, image groups running past the batch"""
    cin, cout, H, W = TP2_SIGS[idx]
    if N == 7 and H * W > 1000:
        N = 2
    _plan_check(("t", 3, 2, 1, cin, cout, H, W, 0, 1), N, cuda, 10, 1, seed=700 + idx, family="tp2")


def _tp2s_id():
    from wav2lip_amd import _lib
    lib = _lib.load()
    sid = lib.w2l_conv_num_tiles() - 3
    assert lib.w2l_conv_config_family(sid) == 7
    return sid


TP2S_SIGS = TP2_SIGS + [(32, 64, 4, 4), (48, 192, 7, 3), (16, 64, 13, 1), (256, 64, 2, 2)]


@pytest.mark.parametrize("N", [1, 3, 7])
@pytest.mark.parametrize("idx", range(len(TP2S_SIGS)))
def test_fused_phase_transposed_conv_with_split_operands_matches_oracle(idx, N, cuda):
    """conv_tp2s.hip (the fused-phase kernel with every fp32 operand as three bf16 pieces on the bf16 matrix cores: the input block
    staged AND split once per 16-channel step for all nine (tap, phase) products) == oracle at the fp32 kernels' tolerance on the
    generator's five upsampling layers, odd / single-pixel / single-row inputs, every pixel-block geometry the host picks, image
    groups running past the batch, odd numbers of 16-channel chunks; the forced configuration must be the kernel that runs"""
    cin, cout, H, W = TP2S_SIGS[idx]
    if cin % 16:
        pytest.skip("%d input channels do not fit conv_tp2s (falls back, covered elsewhere)" % cin)
    if N == 7 and H * W > 1000:
        N = 2
    plan = _plan_check(("t", 3, 2, 1, cin, cout, H, W, 0, 1), N, cuda, _tp2s_id(), 1, seed=1700 + idx, family="tp2s")
    assert plan.resolved()[0][3][0] == _tp2s_id()


@pytest.mark.parametrize("ks", [2, 3, 4, 64])
@pytest.mark.parametrize("idx,N", [(0, 3), (1, 2), (7, 1), (8, 5)])
def test_fused_phase_split_operand_kernel_with_split_k(idx, N, ks, cuda):
    """split-K of conv_tp2s (layers with few pixel blocks: 1024 -> 512 at 3x3 is 72 work items for 256 CUs): the K ranges write raw
    partial sums, the fixed-order reduce launch applies scale / shift / activation; uneven ranges (3 of 4 chunks), more ranges than
    chunks (clamped), a two-chunk layer"""
    cin, cout, H, W = TP2S_SIGS[idx]
    plan = _plan_check(("t", 3, 2, 1, cin, cout, H, W, 0, 1), N, cuda, _tp2s_id(), ks, seed=1800 + idx + ks, family="tp2s")
    got = plan.resolved()[0][3]
    nkc = cin // 16
    assert got[0] == _tp2s_id() and got[1] == -(-nkc // -(-nkc // min(ks, nkc)))


def test_fused_phase_split_operand_kernel_slices_accuracy_and_weight_updates(cuda):
    """channel-sliced input / output as the decoder uses it; error against an fp64 contraction not above the fp32 fused-phase
    kernel's (same weights, same input); and w2l_conv_update re-splits the weights (the pre-split planes are built by the layer's
    first launch on this id: a second parameter version must not run on the first one's pieces)"""
    from wav2lip_amd import engine
    sid = _tp2s_id()
    m = _make("t", 3, 2, 1, 64, 64, 0, 1, 93).to(cuda)
    layer = m.fused()
    layer.set_tile(sid)
    N, H, W = 2, 5, 6
    src = torch.randn(N, H, W, 96, device=cuda)
    dst = torch.full((N, 2 * H, 2 * W, 80), 7.0, device=cuda)
    a_in, a_out = engine.Act(src, 32, 64), engine.Act(dst, 0, 64)
    layer.forward_raw(N, H, W, a_in.ptr, a_in.cs, a_out.ptr, a_out.cs)
    x = src[..., 32:96].permute(0, 3, 1, 2).contiguous().cpu()
    sd = {"b." + key: v.cpu() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", "k3s2x2p1To1")
    assert (dst[..., :64].permute(0, 3, 1, 2).cpu() - ref).abs().max() <= 1e-4
    assert bool((dst[..., 64:] == 7.0).all()), "wrote outside its slice"
    # accuracy against fp64: K = 9 taps x 512 channels on the deepest phase
    m = _make("t", 3, 2, 1, 512, 64, 0, 1, 94)
    x = torch.randn(2, 512, 6, 6)
    sd64 = {"b." + key: v.double() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref64 = models_ref.block(x.double(), sd64, "b", "k3s2x2p1To1")
    layer = m.to(cuda).fused()
    xin = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    errs = {}
    for name, tile in (("fp32", 10), ("split", sid)):
        y = torch.zeros(2, 12, 12, 64, device=cuda)
        plan = engine.Plan()
        plan.add("l", layer, engine.Act(xin, 0, 512), engine.Act(y, 0, 64), None)
        plan.tuned = True
        plan.set_config(0, tile, 1)
        assert plan.resolved()[0][3][0] == tile
        plan.run()
        errs[name] = (y.permute(0, 3, 1, 2).cpu().double() - ref64).abs()
    assert errs["split"].max() <= 1.5 * errs["fp32"].max() + 1e-7 and errs["split"].mean() <= 1.2 * errs["fp32"].mean() + 1e-8, \
        {k: (v.max().item(), v.mean().item()) for k, v in errs.items()}
    # a weight update reaches the split planes (w2l_conv_update: what a training step calls after the optimiser)
    from wav2lip_amd import _lib
    w2 = (m.conv_block[0].weight.detach() * -0.5).contiguous()
    _lib.check(_lib.load().w2l_conv_update(layer.handle, _lib.ptr(w2), None, None, _lib.current_stream()), "conv_update")
    with torch.no_grad():
        m.conv_block[0].weight.copy_(w2)
    layer.set_tile(sid)
    y2 = torch.zeros(2, 12, 12, 64, device=cuda)
    layer.forward_raw(2, 6, 6, engine.ptr(xin), 512, engine.ptr(y2), 64)
    sd2 = {"b." + key: v.cpu() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref2 = models_ref.block(x, sd2, "b", "k3s2x2p1To1")
    assert (y2.permute(0, 3, 1, 2).cpu() - ref2).abs().max() <= 1e-4 + 1e-4 * ref2.abs().max()


def _stem7s_id():
    from wav2lip_amd import _lib
    lib = _lib.load()
    sid = lib.w2l_conv_num_tiles() - 2
    assert lib.w2l_conv_config_family(sid) == 8
    return sid


@pytest.mark.parametrize("N,cin,H,W,kind", [(1, 6, 96, 96, "c"), (3, 6, 96, 96, "c"), (2, 6, 50, 37, "c"), (5, 6, 16, 16, "c"), (2, 5, 5, 3, "c"),
                                            (1, 8, 33, 64, "c"), (2, 6, 48, 48, "n")])
def test_first_layer_split_operand_kernel_matches_oracle(N, cin, H, W, kind, cuda):
    """conv_stem7s.hip (Conv2d(cin <= 8, 16, 7, 1, 3): the 22 x 22 input region of a 16 x 16 pixel block staged and split once, the
    whole 49-tap contraction out of LDS on v_mfma_f32_16x16x32_bf16) == oracle at the fp32 kernels' tolerance: the generator's first
    layer at its real size, ragged blocks in both directions, images smaller than a block (every tap partly outside), 5 / 6 / 8 input
    channels, the LeakyReLU no-norm form; the forced configuration must be the kernel that runs"""
    plan = _plan_check((kind, 7, 1, 3, cin, 16, H, W, 0, 0), N, cuda, _stem7s_id(), 1, seed=2100 + N + H, family="stem7s")
    assert plan.resolved()[0][3][0] == _stem7s_id()


def test_first_layer_split_operand_kernel_slices_accuracy_and_weight_updates(cuda):
    """channel-sliced input (stride 12, 8 channels read) and output (16 channels of a 24-wide buffer, the rest untouched); error against
    fp64 not above the fp32 implicit GEMM's; w2l_conv_update re-splits the weights"""
    from wav2lip_amd import engine, _lib
    sid = _stem7s_id()
    m = _make("c", 7, 1, 3, 6, 16, 0, 0, 95)
    N, H, W = 2, 40, 27
    x = torch.randn(N, 6, H, W)
    sd64 = {"b." + key: v.double() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref64 = models_ref.block(x.double(), sd64, "b", "k7s1x1p3")
    layer = m.to(cuda).fused()
    src = torch.zeros(N, H, W, 12, device=cuda)
    src[..., 4:10] = x.permute(0, 2, 3, 1).to(cuda)
    errs = {}
    for name, tile in (("fp32", 4), ("split", sid)):
        dst = torch.full((N, H, W, 24), 7.0, device=cuda)
        layer.set_tile(tile)
        a_in, a_out = engine.Act(src, 4, 8), engine.Act(dst, 4, 16)
        layer.forward_raw(N, H, W, a_in.ptr, a_in.cs, a_out.ptr, a_out.cs)
        assert bool((dst[..., :4] == 7.0).all()) and bool((dst[..., 20:] == 7.0).all()), "wrote outside its slice"
        errs[name] = (dst[..., 4:20].permute(0, 3, 1, 2).cpu().double() - ref64).abs()
    # (measured: max 2.1e-6 against 1.1e-6, mean 5.3e-8 against 4.3e-8 on outputs of magnitude ~1 - one element's maximum is not a
    # statistic; the mean is what the piece arithmetic is held to)
    assert errs["split"].max() <= 3.0 * errs["fp32"].max() + 1e-7 and errs["split"].mean() <= 1.5 * errs["fp32"].mean() + 1e-8, \
        {k: (v.max().item(), v.mean().item()) for k, v in errs.items()}
    w2 = (m.conv_block[0].weight.detach() * -0.5).contiguous()
    _lib.check(_lib.load().w2l_conv_update(layer.handle, _lib.ptr(w2), None, None, _lib.current_stream()), "conv_update")
    with torch.no_grad():
        m.conv_block[0].weight.copy_(w2)
    layer.set_tile(sid)
    dst = torch.zeros(N, H, W, 16, device=cuda)
    layer.forward_raw(N, H, W, engine.Act(src, 4, 8).ptr, 12, engine.ptr(dst), 16)
    sd2 = {"b." + key: v.cpu() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref2 = models_ref.block(x, sd2, "b", "k7s1x1p3")
    assert (dst.permute(0, 3, 1, 2).cpu() - ref2).abs().max() <= 1e-4 + 1e-4 * ref2.abs().max()


def _k3s_id():
    from wav2lip_amd import _lib
    lib = _lib.load()
    sid = lib.w2l_conv_num_tiles() - 1
    assert lib.w2l_conv_config_family(sid) == 9
    return sid


@pytest.mark.parametrize("cin,H,W,res,N,kind", [(80, 96, 96, 0, 2, "c"), (32, 48, 48, 1, 3, "c"), (16, 5, 7, 0, 5, "c"), (48, 1, 1, 0, 9, "c"),
                                                (32, 80, 16, 1, 2, "c"), (64, 17, 33, 0, 1, "c"), (32, 2, 2, 1, 30, "c"), (16, 12, 12, 0, 7, "n"),
                                                (32, 9, 50, 1, 4, "c")])
def test_direct_3x3_split_operand_kernel_matches_oracle(cin, H, W, res, N, kind, cuda):
    """conv_k3s.hip (3x3 stride-1 layers with 32 couts: the input block with its halo staged and split once per 16-channel step, nine
    taps = nine shifted views, six bf16 piece products per product) == oracle at the fp32 kernels' tolerance: the output block's
    shape, the 32 -> 32 residual blocks of both encoders, every pixel-block geometry the host picks, ragged blocks, image groups running
    past the batch, one to five chunks, LeakyReLU without norm; the forced configuration must be the kernel that runs"""
    plan = _plan_check((kind, 3, 1, 1, cin, 32, H, W, res, 0), N, cuda, _k3s_id(), 1, seed=2300 + cin + H, family="k3s")
    assert plan.resolved()[0][3][0] == _k3s_id()


@pytest.mark.parametrize("N,H,W,y_cs", [(3, 20, 24, 4), (2, 96, 96, 4), (5, 7, 9, 5), (1, 1, 3, 4)])
def test_direct_3x3_split_operand_kernel_with_the_fused_head(N, H, W, y_cs, cuda):
    """the generator's output block (80 -> 32 conv3x3 + BN + ReLU, 32 -> 3 conv1x1, sigmoid) as ONE conv_k3s launch, as the plan runs
    it (channel stride 4) and with a stride the head's scalar stores must respect; the other channels stay untouched"""
    from wav2lip_amd import engine
    from wav2lip_amd.models.conv import HeadFusedBlock
    from wav2lip_amd._lib import ACT_SIGMOID
    m = _make("c", 3, 1, 1, 80, 32, 0, 0, 15)
    torch.manual_seed(16)
    conv1 = torch.nn.Conv2d(32, 3, 1)
    x = torch.randn(N, 80, H, W)
    ref = _head_ref(m, conv1, x, "k3p1")
    m, conv1 = m.to(cuda), conv1.to(cuda)
    layer = HeadFusedBlock(m, conv1, ACT_SIGMOID).fused()
    xin = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    y = torch.full((N, H, W, y_cs), 2.0, device=cuda)
    plan = engine.Plan()
    plan.add("l", layer, engine.Act(xin, 0, 80), engine.Act(y, 0, 3), None)
    plan.tuned = True
    plan.set_config(0, _k3s_id(), 1)
    assert plan.resolved()[0][2] == "k3s" and plan.resolved()[0][3][0] == _k3s_id(), plan.resolved()
    plan.run()
    got = y[..., :3].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 2e-6, (got - ref).abs().max()
    assert bool((y[..., 3:] == 2.0).all()), "wrote outside its channels"


def test_fused_phase_transposed_conv_writes_channel_slices(cuda):
    """as the decoder uses it: output into channels [0, cout) of a wider concat buffer, input from a channel slice"""
    from wav2lip_amd import engine
    m = _make("t", 3, 2, 1, 64, 64, 0, 1, 91).to(cuda)
    layer = m.fused()
    layer.set_tile(10)
    N, H, W = 2, 5, 6
    src = torch.randn(N, H, W, 96, device=cuda)
    dst = torch.full((N, 2 * H, 2 * W, 80), 7.0, device=cuda)
    a_in, a_out = engine.Act(src, 32, 64), engine.Act(dst, 0, 64)
    layer.forward_raw(N, H, W, a_in.ptr, a_in.cs, a_out.ptr, a_out.cs)
    x = src[..., 32:96].permute(0, 3, 1, 2).contiguous().cpu()
    sd = {"b." + key: v.cpu() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", "k3s2x2p1To1")
    got = dst[..., :64].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 1e-4
    assert bool((dst[..., 64:] == 7.0).all()), "wrote outside its slice"


def test_winograd_is_the_default_on_big_layers_and_slices(cuda):
    """heuristic path (no plan): module forward of a big 3x3 layer runs Winograd; channel-sliced output + aliasing residual"""
    from wav2lip_amd import engine
    _check(("c", 3, 1, 1, 64, 64, 96, 96, 1, 0), 4, cuda, seed=77)
    m = _make("c", 3, 1, 1, 64, 64, 1, 0, 78).to(cuda)
    layer = m.fused()
    layer.set_tile(6)
    N, H, W = 2, 10, 13
    src = torch.randn(N, H, W, 96, device=cuda)
    dst = torch.full((N, H, W, 80), 7.0, device=cuda)
    a_in, a_out = engine.Act(src, 32, 64), engine.Act(dst, 8, 64)
    layer.forward_raw(N, H, W, a_in.ptr, a_in.cs, a_out.ptr, a_out.cs, a_in.ptr, a_in.cs)
    x = src[..., 32:96].permute(0, 3, 1, 2).contiguous().cpu()
    sd = {"b." + key: v.cpu() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", "k3p1r")
    got = dst[..., 8:72].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 1e-4
    assert bool((dst[..., :8] == 7.0).all()) and bool((dst[..., 72:] == 7.0).all()), "wrote outside its slice"


def test_argument_errors_are_codes_with_messages_not_crashes(cuda):
    """empty / oversized / misaligned requests return W2L_ERR_ARG and a message; nothing is launched"""
    import ctypes as C
    from wav2lip_amd import _lib, engine
    lib = _lib.load()
    m = _make("c", 3, 1, 1, 64, 64, 1, 0, 1).to(cuda)
    layer = m.fused()
    x = torch.zeros(1, 8, 8, 64, device=cuda)
    y = torch.zeros(1, 8, 8, 64, device=cuda)
    s = _lib.current_stream()

    def call(N, H, W, x_cs=64, y_cs=64, xp=None):
        return lib.w2l_conv_forward(layer.handle, s, N, H, W, xp or _lib.ptr(x), x_cs, _lib.ptr(y), y_cs, None, 0)
    assert call(0, 8, 8) == -1 and b"bad shape" in lib.w2l_last_error()                 # empty batch
    assert call(1, 8, 8, x_cs=62) == -1 and b"x_cs" in lib.w2l_last_error()              # channel stride not a multiple of 4
    assert call(1, 8, 8, y_cs=32) == -1 and b"y_cs" in lib.w2l_last_error()              # output slice too narrow
    assert call(1 << 14, 1024, 1024) == -1 and b"2 GiB" in lib.w2l_last_error()          # 32-bit buffer offsets: split the batch
    # the byte limit (not a pixel limit) holds in front of EVERY kernel family: 911 images of 96x96x64 fp32 are 2.15 GB, pixel
    # count far below 2^31; the Winograd / quarter-split / F(4x4) launchers (forced configuration ids) must never see it - their
    # buffer descriptors use 32-bit byte offsets with 0x80000000 as the padding sentinel, which would land INSIDE such a buffer
    for tid in range(lib.w2l_conv_num_tiles()):
        layer.set_tile(tid)
        assert call(911, 96, 96) == -1 and b"2 GiB" in lib.w2l_last_error(), tid
    layer.set_tile(-1)
    assert call(911, 96, 96) == -1 and b"2 GiB" in lib.w2l_last_error()
    assert call(1, 8, 8, xp=C.c_void_p(x.data_ptr() + 4)) == -1 and b"aligned" in lib.w2l_last_error()
    g = _lib.ConvGeom(0, 64, 64, 3, 3, 1, 1, 1, 1, 0, 0, 1)
    assert lib.w2l_conv_wgrad(C.byref(g), s, 1 << 14, 1024, 1024, _lib.ptr(x), 64, _lib.ptr(y), 64, _lib.ptr(y)) == -1
    assert call(1, 8, 8) == 0                                                             # and the handle still works
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.zeros(1, 64, 8, 8))                                                       # CPU tensor: no fallback


def test_two_streams_run_split_k_layers_concurrently(cuda):
    """the split-K scratch is per stream: the same deep layer on two streams at once gives the single-stream answer"""
    from wav2lip_amd import engine
    torch.manual_seed(5)
    m = _make("c", 3, 1, 1, 512, 512, 1, 0, 3).to(cuda)
    layer = m.fused()
    N, H, W = 8, 3, 3
    xs = [torch.randn(N, H, W, 512, device=cuda) for _ in range(2)]
    ys = [torch.zeros(N, H, W, 512, device=cuda) for _ in range(2)]
    ref = []
    for x, y in zip(xs, ys):
        plan = engine.Plan()
        plan.add("l", layer, engine.Act(x, 0, 512), engine.Act(y, 0, 512), engine.Act(x, 0, 512))
        plan.set_config(0, 3, 8)                       # 64x64 tiles, 8-way split-K
        plan.tuned = True
        plan.run()
        ref.append((plan, y.clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(20):
        for (plan, _), y, st in zip(ref, ys, streams):
            y.zero_()
            with torch.cuda.stream(st):
                plan.run()
        torch.cuda.synchronize()
        for (_, want), y in zip(ref, ys):
            assert torch.equal(y, want)


# ---------------------------------------------------------------- split-operand F(2x2,3x3) Winograd (conv_wino2s.hip, the last id)
WINO2S_EXTRA = [(64, 64, 13, 11, 1), (64, 64, 4, 4, 0), (80, 64, 33, 35, 0), (64, 192, 8, 8, 0), (64, 64, 2, 3, 1), (128, 64, 6, 6, 0),
                (16, 64, 24, 24, 0), (48, 128, 12, 12, 0), (256, 256, 12, 12, 1), (512, 512, 6, 6, 1), (32, 64, 1, 1, 0)]


def _wino2s_id():
    from wav2lip_amd import _lib
    lib = _lib.load()
    sid = lib.w2l_conv_num_tiles() - 4
    assert lib.w2l_conv_config_family(sid) == 6
    return sid


@pytest.mark.parametrize("N", [1, 3, 9])
@pytest.mark.parametrize("idx", range(len(WINO_SIGS) + len(WINO2S_EXTRA)))
def test_winograd_f2x2_split_operand_matches_oracle(idx, N, cuda):
    """conv_wino2s.hip (F(2x2,3x3) with every transformed operand as three bf16 pieces on the bf16 matrix cores, 64 tiles x 64
    couts per workgroup, row halves half a K-step apart, raw blocks by LDS-DMA) == oracle at the fp32 kernels' tolerance: every
    tile-block geometry the host picks (8x8x1 ... 1x1x30), odd extents (ragged tiles masked on store), image groups running past
    the batch, odd numbers of 16-channel chunks (one more chunk of zeros), residual, single-pixel images; the forced configuration
    must be the kernel that runs"""
    cin, cout, H, W, res = (WINO_SIGS + WINO2S_EXTRA)[idx]
    if cin % 16 or cout % 64:
        pytest.skip("%d->%d channels do not fit conv_wino2s (falls back, covered elsewhere)" % (cin, cout))
    if N == 9 and H * W > 3000:
        N = 4
    plan = _plan_check(("c", 3, 1, 1, cin, cout, H, W, res, 0), N, cuda, _wino2s_id(), 1, seed=1100 + idx, family="wino2s")
    assert plan.resolved()[0][3][0] == _wino2s_id()


def test_winograd_f2x2_split_operand_leaky_no_norm_slices_and_data_gradient_form(cuda):
    """nonorm_Conv2d (LeakyReLU, no BN) 512 -> 512; channel-sliced input / output with an aliasing residual; the transposed 3x3 s1
    p1 layer (the data gradient of a conv: flipped kernel, swapped channel roles, same transformed weights as the F(2x2) fp32 kernels)"""
    from wav2lip_amd import engine
    sid = _wino2s_id()
    _plan_check(("n", 3, 1, 1, 512, 512, 6, 6, 0, 0), 5, cuda, sid, 1, seed=51, family="wino2s")
    _plan_check(("n", 3, 1, 1, 512, 512, 3, 3, 0, 0), 7, cuda, sid, 1, seed=52, family="wino2s")
    _plan_check(("t", 3, 1, 1, 64, 128, 12, 12, 0, 0), 3, cuda, sid, 1, seed=53, family="wino2s")
    m = _make("c", 3, 1, 1, 64, 64, 1, 0, 89).to(cuda)
    layer = m.fused()
    layer.set_tile(sid)
    N, H, W = 2, 10, 13
    src = torch.randn(N, H, W, 96, device=cuda)
    dst = torch.full((N, H, W, 80), 7.0, device=cuda)
    a_in, a_out = engine.Act(src, 32, 64), engine.Act(dst, 8, 64)
    layer.forward_raw(N, H, W, a_in.ptr, a_in.cs, a_out.ptr, a_out.cs, a_in.ptr, a_in.cs)
    x = src[..., 32:96].permute(0, 3, 1, 2).contiguous().cpu()
    sd = {"b." + key: v.cpu() for key, v in m.state_dict().items()}
    with torch.no_grad():
        ref = models_ref.block(x, sd, "b", "k3p1r")
    got = dst[..., 8:72].permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max() <= 1e-4
    assert bool((dst[..., :8] == 7.0).all()) and bool((dst[..., 72:] == 7.0).all()), "wrote outside its slice"


def test_winograd_f2x2_split_operand_is_as_accurate_as_the_fp32_f2x2_kernel_and_tracks_weight_updates(cuda):
    """K = 512 channels at 12x12: against an fp64 convolution the split-operand kernel's error is not above 1.5x the fp32 F(2x2)
    kernel's (same transformed weights, products exact to 2^-24, six instead of eight accumulator roundings per 16 channels);
    after w2l_conv_update the bf16 planes are rebuilt from the new weights"""
    from wav2lip_amd import engine
    sid = _wino2s_id()
    torch.manual_seed(78)
    m = _make("c", 3, 1, 1, 512, 512, 0, 0, 78)
    N, H, W = 3, 12, 12
    x = torch.randn(N, 512, H, W)

    def ref64(mm):
        with torch.no_grad():
            conv, bn = mm.conv_block[0], mm.conv_block[1]
            z = torch.nn.functional.conv2d(x.double(), conv.weight.double().cpu(), conv.bias.double().cpu(), padding=1)
            sc = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)).cpu()
            return torch.relu((z - bn.running_mean.double().cpu()[None, :, None, None]) * sc[None, :, None, None]
                              + bn.bias.double().cpu()[None, :, None, None])

    ref = ref64(m)
    mg = m.to(cuda)
    layer = mg.fused()
    xin = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    errs = {}
    for name, tile in (("fp32", 8), ("split", sid)):
        y = torch.zeros(N, H, W, 512, device=cuda)
        plan = engine.Plan()
        plan.add("l", layer, engine.Act(xin, 0, 512), engine.Act(y, 0, 512), None)
        plan.tuned = True
        plan.set_config(0, tile, 1)
        assert plan.resolved()[0][3][0] == tile
        plan.run()
        errs[name] = (y.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item()
    assert errs["split"] <= 1.5 * errs["fp32"] + 1e-7, errs
    # w2l_conv_update (what a training step calls after the optimiser): the fp32 transformed weights are re-packed and the bf16
    # planes re-split from them
    from wav2lip_amd import _lib
    w2 = (mg.conv_block[0].weight.detach() * -0.5).contiguous()
    _lib.check(_lib.load().w2l_conv_update(layer.handle, _lib.ptr(w2), None, None, _lib.current_stream()), "conv_update")
    with torch.no_grad():
        mg.conv_block[0].weight.copy_(w2)
    layer.set_tile(sid)
    y = torch.zeros(N, H, W, 512, device=cuda)
    layer.forward_raw(N, H, W, engine.ptr(xin), 512, engine.ptr(y), 512)
    e2 = (y.permute(0, 3, 1, 2).cpu().double() - ref64(mg)).abs().max().item()
    assert e2 <= 1e-4, e2
