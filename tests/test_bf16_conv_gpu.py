"""bf16-storage conv kernels (csrc/conv_bf16.hip, through the C ABI: w2l_convb_*) against torch CPU convolutions of the SAME
bf16-rounded operands computed in float64: products of bf16 values are exact in fp32, so what is left is the fp32 accumulation
order (~1e-6 of sum|a*b|) and the ONE rounding of the result to bf16 (relative 2^-8; a result within accumulation noise of a
rounding boundary may land on the neighbouring bf16 value: 2^-7).  Tolerance, written out: |got - ref| <= |ref| / 128 + 2e-5 * S,
S = max |ref| of the layer."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_conv_gpu import SIGS
from wav2lip_amd import _lib, bf16
from wav2lip_amd._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, ConvGeom

pytestmark = pytest.mark.gpu


def _pair(v):
    return tuple(v) if isinstance(v, tuple) else (v, v)


def _rb(t):
    """round to bf16 and back (what the device tensors hold)"""
    return t.to(torch.bfloat16).to(torch.float64)


def _nhwc(x_nchw, cs=None, off=0, fill=0.0):
    """NCHW fp -> NHWC bf16 buffer with channel stride cs, data at channel offset off"""
    N, Cn, H, W = x_nchw.shape
    cs = cs or bf16.round8(Cn)
    buf = torch.full((N, H, W, cs), fill, dtype=torch.bfloat16)
    buf[..., off:off + Cn] = x_nchw.permute(0, 2, 3, 1).to(torch.bfloat16)
    buf[..., off + Cn:off + bf16.round8(Cn)] = 0            # pad channels of the slice are zero by contract
    return buf


def _ref(x, w, transposed, stride, pad, outpad, scale, shift, res, act):
    xd, wd = _rb(x), _rb(w)
    if transposed:
        z = F.conv_transpose2d(xd, wd, None, stride, pad, outpad)
    else:
        z = F.conv2d(xd, wd, None, stride, pad)
    if scale is not None:
        z = z * scale.double().view(1, -1, 1, 1)
    if shift is not None:
        z = z + shift.double().view(1, -1, 1, 1)
    if res is not None:
        z = z + _rb(res)
    if act == ACT_RELU:
        z = z.clamp_min(0)
    elif act == ACT_LEAKY:
        z = torch.where(z > 0, z, 0.01 * z)
    elif act == ACT_SIGMOID:
        z = torch.sigmoid(z)
    return z


def _run(cuda, transposed, cin, cout, k, stride, pad, outpad, N, H, W, act=ACT_RELU, with_res=False, affine=True, tile=None,
         ksplit=0, seed=0, wscale=None):
    torch.manual_seed(seed)
    kh, kw = _pair(k)
    s, p, op = _pair(stride), _pair(pad), _pair(outpad)
    wshape = (cin, cout, kh, kw) if transposed else (cout, cin, kh, kw)
    fan = (cin if not transposed else cin) * kh * kw
    w = torch.randn(wshape) * (wscale or (1.0 / np.sqrt(fan)))
    x = torch.randn(N, cin, H, W)
    scale = torch.rand(cout) + 0.5 if affine else None
    shift = torch.randn(cout) * 0.2 if affine else None
    g = ConvGeom(int(transposed), cin, cout, kh, kw, s[0], s[1], p[0], p[1], op[0], op[1], act)
    layer = bf16.ConvB(g, w.to(cuda))
    if tile is not None:
        layer.set_tile(tile)
    Ho, Wo = layer.out_hw(H, W)
    res = torch.randn(N, cout, Ho, Wo) if with_res else None
    ref = _ref(x, w, transposed, s, p, op, scale, shift, res, act)
    assert tuple(ref.shape) == (N, cout, Ho, Wo)
    xb = _nhwc(x).to(cuda)
    yb = torch.full((N, Ho, Wo, bf16.round8(cout)), 3.0, dtype=torch.bfloat16, device=cuda)
    rb = _nhwc(res).to(cuda) if with_res else None
    layer.run(bf16.ActB(xb, 0, cin), bf16.ActB(yb, 0, cout), bf16.ActB(rb, 0, cout) if with_res else None,
              scale.to(cuda) if affine else None, shift.to(cuda) if affine else None, ksplit)
    torch.cuda.synchronize()
    got = yb[..., :cout].permute(0, 3, 1, 2).double().cpu()
    err = (got - ref).abs()
    S = float(ref.abs().max())
    tol = ref.abs() / 128 + 2e-5 * S
    bad = int((err > tol).sum())
    assert bad == 0, "%d of %d outside tolerance; max err %.3e (S = %.3e), worst ratio %.2f" % (
        bad, err.numel(), float(err.max()), S, float((err / tol).max()))
    if bf16.round8(cout) > cout:
        assert bool((yb[..., cout:] == 0).all()), "pad channels must be written as zero"
    return float((err / (ref.abs() / 256 + 1e-30)).median())


@pytest.mark.parametrize("idx", range(len(SIGS)))
def test_forward_signature(idx, cuda):
    """all 58 fused-conv signatures of the hot path, N = 3"""
    kind, k, stride, pad, cin, cout, H, W, residual, outpad = SIGS[idx]
    _run(cuda, kind == "t", cin, cout, k, stride, pad, outpad, 3, H, W, act=ACT_LEAKY if kind == "n" else ACT_RELU,
         with_res=bool(residual), seed=idx)


@pytest.mark.parametrize("idx", range(len(SIGS)))
def test_data_gradient_geometry(idx, cuda):
    """the data gradient of every signature: the conv's weight tensor read as a transposed conv (and vice versa), the output
    gradient as input - what wav2lip_amd/autograd.py launches in backward; no scale / shift / activation, accumulating residual"""
    kind, k, stride, pad, cin, cout, H, W, residual, outpad = SIGS[idx]
    s, p = _pair(stride), _pair(pad)
    kh, kw = _pair(k)
    torch.manual_seed(1000 + idx)
    N = 2
    if kind != "t":
        Ho, Wo = (H + 2 * p[0] - kh) // s[0] + 1, (W + 2 * p[1] - kw) // s[1] + 1
        w = torch.randn(cout, cin, kh, kw) / np.sqrt(cout * kh * kw)
        dz = torch.randn(N, cout, Ho, Wo)
        op = ((H + 2 * p[0] - kh) % s[0], (W + 2 * p[1] - kw) % s[1])
        g = ConvGeom(1, cout, cin, kh, kw, s[0], s[1], p[0], p[1], op[0], op[1], ACT_NONE)
        ref = F.conv_transpose2d(_rb(dz), _rb(w), None, s, p, op)
    else:
        op = _pair(outpad)
        Ho, Wo = (H - 1) * s[0] - 2 * p[0] + kh + op[0], (W - 1) * s[1] - 2 * p[1] + kw + op[1]
        w = torch.randn(cin, cout, kh, kw) / np.sqrt(cout * kh * kw)
        dz = torch.randn(N, cout, Ho, Wo)
        g = ConvGeom(0, cout, cin, kh, kw, s[0], s[1], p[0], p[1], 0, 0, ACT_NONE)
        ref = F.conv2d(_rb(dz), _rb(w), None, s, p)
    assert tuple(ref.shape) == (N, cin, H, W)
    prev = torch.randn(N, cin, H, W)          # a contribution already sitting in the gradient buffer
    ref = ref + _rb(prev)
    layer = bf16.ConvB(g, w.to(cuda))
    dzb = _nhwc(dz).to(cuda)
    gxb = _nhwc(prev).to(cuda)
    gx = bf16.ActB(gxb, 0, cin)
    layer.run(bf16.ActB(dzb, 0, cout), gx, gx)          # res aliases y: accumulate in place
    torch.cuda.synchronize()
    got = gxb[..., :cin].permute(0, 3, 1, 2).double().cpu()
    err = (got - ref).abs()
    S = float(ref.abs().max())
    tol = ref.abs() / 128 + 2e-5 * S
    assert int((err > tol).sum()) == 0, "max err %.3e (S = %.3e)" % (float(err.max()), S)


@pytest.mark.parametrize("tile", range(5))
@pytest.mark.parametrize("idx", [1, 10, 12, 22, 23, 29, 33, 35, 44])
def test_every_tile(idx, tile, cuda):
    kind, k, stride, pad, cin, cout, H, W, residual, outpad = SIGS[idx]
    _run(cuda, kind == "t", cin, cout, k, stride, pad, outpad, 2, H, W, with_res=bool(residual), tile=tile, seed=200 + idx)


@pytest.mark.parametrize("ks", [2, 3, 8])
@pytest.mark.parametrize("idx", [20, 21, 22, 23, 25, 42])
def test_split_k(idx, ks, cuda):
    """deep small-spatial layers: fp32 partial sums in the workspace + the reduce kernel (also ragged transposed phases)"""
    kind, k, stride, pad, cin, cout, H, W, residual, outpad = SIGS[idx]
    _run(cuda, kind == "t", cin, cout, k, stride, pad, outpad, 5, H, W, with_res=bool(residual), ksplit=ks, seed=300 + idx)
    _run(cuda, kind == "t", cin, cout, k, stride, pad, outpad, 5, H, W, with_res=bool(residual), ksplit=ks, tile=3, seed=301 + idx)


def test_activations_ragged_shapes_and_batch_one(cuda):
    _run(cuda, False, 32, 3, 1, 1, 0, 0, 2, 96, 96, act=ACT_SIGMOID, affine=True, seed=1)          # RGB head: cout 3 -> 8 channels written
    _run(cuda, False, 512, 1, 1, 1, 0, 0, 7, 1, 1, act=ACT_SIGMOID, seed=2)                        # discriminator prediction
    _run(cuda, False, 64, 64, 3, 1, 1, 0, 1, 96, 96, act=ACT_NONE, affine=False, seed=3)           # batch 1
    _run(cuda, False, 24, 40, 3, 1, 1, 0, 3, 13, 11, act=ACT_LEAKY, with_res=True, seed=4)         # cin_p 24: K chunks wrap taps
    _run(cuda, False, 80, 32, 3, 1, 1, 0, 2, 17, 9, seed=5)                                        # cin_p 80 (64 % 80 != 0)
    _run(cuda, True, 16, 24, 3, 2, 1, 1, 2, 7, 5, seed=6)                                          # ragged transposed, odd extents
    _run(cuda, True, 16, 24, 3, 2, 1, 0, 2, 7, 5, seed=7)                                          # output_padding 0: phases do not tile evenly
    _run(cuda, False, 8, 8, 3, 1, 1, 0, 1, 1, 1, seed=8)                                           # a single pixel
    _run(cuda, False, 256, 256, 3, 1, 1, 0, 40, 24, 24, with_res=True, seed=9)                     # 180 M-tiles: several per XCD


def test_channel_sliced_io_writes_only_its_slice(cuda):
    """reads / writes through channel slices of wider NHWC buffers (the generator's concat-free skip connections)"""
    torch.manual_seed(11)
    cin, cout, N, H, W = 32, 32, 2, 10, 12
    w = torch.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)
    src = torch.randn(N, 48, H, W)
    g = ConvGeom(0, cin, cout, 3, 3, 1, 1, 1, 1, 0, 0, ACT_RELU)
    layer = bf16.ConvB(g, w.to(cuda))
    sb = src.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous().to(cuda)
    db = torch.full((N, H, W, 48), 7.0, dtype=torch.bfloat16, device=cuda)
    xa = bf16.ActB(sb, 16, cin)
    layer.run(xa, bf16.ActB(db, 8, cout), xa)       # residual = the input slice
    torch.cuda.synchronize()
    xs = src[:, 16:48]
    ref = (F.conv2d(_rb(xs), _rb(w), None, 1, 1) + _rb(xs)).clamp_min(0)
    got = db[..., 8:40].permute(0, 3, 1, 2).double().cpu()
    assert float(((got - ref).abs() - ref.abs() / 128).max()) <= 2e-5 * float(ref.abs().max())
    assert bool((db[..., :8] == 7.0).all()) and bool((db[..., 40:] == 7.0).all()), "wrote outside its slice"


def test_weight_update_repacks(cuda):
    torch.manual_seed(12)
    w0, w1 = torch.randn(64, 64, 3, 3) * 0.05, torch.randn(64, 64, 3, 3) * 0.05
    x = torch.randn(2, 64, 12, 12)
    g = ConvGeom(0, 64, 64, 3, 3, 1, 1, 1, 1, 0, 0, ACT_NONE)
    layer = bf16.ConvB(g, w0.to(cuda))
    xb = _nhwc(x).to(cuda)
    yb = bf16.new_buf(2, 12, 12, 64, cuda)
    layer.update(w1.to(cuda))
    layer.run(bf16.ActB(xb, 0, 64), bf16.ActB(yb, 0, 64))
    ref = F.conv2d(_rb(x), _rb(w1), None, 1, 1)
    got = yb.permute(0, 3, 1, 2).double().cpu()
    assert float(((got - ref).abs() - ref.abs() / 128).max()) <= 2e-5 * float(ref.abs().max())


def test_update_many_writes_what_the_single_updates_write(cuda):
    """w2l_convb_update_many (one launch for a whole train graph) == w2l_convb_update per layer, bit for bit: plain, strided,
    transposed (four phases), transposed on a 1x1 input (the second slab set) and a ragged-channel layer; the cached tables are
    re-used on the second call and survive the destruction of one of their layers"""
    torch.manual_seed(13)
    geoms = [(ConvGeom(0, 64, 64, 3, 3, 1, 1, 1, 1, 0, 0, ACT_NONE), (64, 64, 3, 3), (2, 64, 12, 12)),
             (ConvGeom(0, 16, 32, 3, 3, 2, 2, 1, 1, 0, 0, ACT_NONE), (32, 16, 3, 3), (2, 16, 12, 12)),
             (ConvGeom(1, 40, 24, 3, 3, 2, 2, 1, 1, 1, 1, ACT_NONE), (40, 24, 3, 3), (2, 40, 6, 6)),
             (ConvGeom(1, 32, 16, 3, 3, 1, 1, 0, 0, 0, 0, ACT_NONE), (32, 16, 3, 3), (2, 32, 1, 1)),
             (ConvGeom(0, 6, 16, 7, 7, 1, 1, 3, 3, 0, 0, ACT_NONE), (16, 6, 7, 7), (1, 6, 20, 20))]
    w0 = [(torch.randn(ws) * 0.05).to(cuda) for _, ws, _ in geoms]
    w1 = [(torch.randn(ws) * 0.05).to(cuda) for _, ws, _ in geoms]

    def outputs(layers):
        outs = []
        for layer, (g, _, xs) in zip(layers, geoms):
            x = torch.randn(xs, generator=torch.Generator().manual_seed(5))
            xb = _nhwc(x).to(cuda)
            ho, wo = layer.out_hw(xs[2], xs[3])
            yb = bf16.new_buf(xs[0], ho, wo, g.cout, cuda)
            layer.run(bf16.ActB(xb, 0, g.cin), bf16.ActB(yb, 0, g.cout))
            outs.append(yb.clone())
        return outs
    single = [bf16.ConvB(g, w) for (g, _, _), w in zip(geoms, w0)]
    many = [bf16.ConvB(g, w) for (g, _, _), w in zip(geoms, w0)]
    masters = [w.clone() for w in w0]          # the "parameters": updated in place, same storage every step
    for step_w in (w1, w0, w1):
        for layer, w, m in zip(single, step_w, masters):
            m.copy_(w)
            layer.update(w)
        bf16.ConvB.update_many(list(zip(many, masters)))
        for a, b in zip(outputs(single), outputs(many)):
            assert torch.equal(a, b)
    del many[1]                                  # its table entry must not be served to a new layer at the same address
    masters.pop(1)
    import gc
    gc.collect()
    bf16.ConvB.update_many(list(zip(many, masters)))
    for layer_s, layer_m, (g, _, xs) in zip([single[0]] + single[2:], many, [geoms[0]] + geoms[2:]):
        x = torch.randn(xs, generator=torch.Generator().manual_seed(5))
        xb = _nhwc(x).to(cuda)
        ho, wo = layer_s.out_hw(xs[2], xs[3])
        ya, yb = bf16.new_buf(xs[0], ho, wo, g.cout, cuda), bf16.new_buf(xs[0], ho, wo, g.cout, cuda)
        layer_s.run(bf16.ActB(xb, 0, g.cin), bf16.ActB(ya, 0, g.cout))
        layer_m.run(bf16.ActB(xb, 0, g.cin), bf16.ActB(yb, 0, g.cout))
        assert torch.equal(ya, yb)


def test_update_many_rejects_bad_arguments(cuda):
    lib = _lib.load()
    assert lib.w2l_convb_update_many(0, None, None, _lib.current_stream()) != 0
    assert b"update_many" in lib.w2l_last_error()


def test_argument_errors(cuda):
    import ctypes as C
    lib = _lib.load()
    g = ConvGeom(0, 64, 64, 3, 3, 1, 1, 1, 1, 0, 0, ACT_RELU)
    layer = bf16.ConvB(g, torch.zeros(64, 64, 3, 3, device=cuda))
    x = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16, device=cuda)
    y = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16, device=cuda)
    s = _lib.current_stream()

    def call(N, H, W, x_cs=64, y_cs=64, xp=None):
        return lib.w2l_convb_forward(layer.handle, s, N, H, W, xp or _lib.ptr(x), x_cs, _lib.ptr(y), y_cs, None, 0, None, None, 0)
    assert call(0, 8, 8) == -1 and b"bad shape" in lib.w2l_last_error()
    assert call(1, 8, 8, x_cs=60) == -1 and b"x_cs" in lib.w2l_last_error()
    assert call(1, 8, 8, y_cs=32) == -1 and b"y_cs" in lib.w2l_last_error()
    assert call(1900, 96, 96) == -1 and b"2 GiB" in lib.w2l_last_error()             # 2.24 GB of bf16: 32-bit buffer offsets
    assert call(1, 8, 8, xp=C.c_void_p(x.data_ptr() + 2)) == -1 and b"aligned" in lib.w2l_last_error()
    assert call(1, 8, 8) == 0
    with pytest.raises(RuntimeError, match="HIP device"):
        bf16.ConvB(g, torch.zeros(64, 64, 3, 3))


@pytest.mark.parametrize("case", ["conv64", "convT", "deep_splitk", "ragged", "stem", "large_mean", "box_ragged", "stem_box"])
def test_conv_with_batch_statistics_in_the_epilogue(case, cuda):
    """w2l_convb_forward_bn: z = conv(x) + bias in bf16 AND BatchNorm's batch statistics of z (mean, rstd, scale = gamma*rstd,
    shift = beta - mean*scale, running-stat update with momentum and the unbiased variance) - taken in the conv epilogue from the
    bf16-ROUNDED outputs (the z that is stored and that every later pass normalises) when the launch has no split-K ("deep_splitk":
    the stand-alone reduction over the stored z: the same definition) - against float64 of the exact conv AND, tightly, against
    float64 statistics of the stored z.  "large_mean": channels whose |mean| is 30 standard deviations (one-pass E[x^2]-E[x]^2 with
    fp32 per-lane partials: the documented loss is |mean|^2/var * 1e-7 relative on the variance, i.e. 1e-4 here)"""
    torch.manual_seed({"conv64": 1, "convT": 2, "deep_splitk": 3, "ragged": 4, "stem": 5, "large_mean": 6, "box_ragged": 7, "stem_box": 8}[case])
    tr, cin, cout, k, s, p, op, N, H, W = {
        "conv64": (False, 64, 64, 3, 1, 1, 0, 228, 48, 48), "convT": (True, 96, 40, 3, 2, 1, 1, 3, 11, 9),
        "deep_splitk": (False, 512, 512, 3, 1, 1, 0, 7, 3, 3), "ragged": (False, 24, 72, 3, 1, 1, 0, 3, 13, 7),
        "stem": (False, 6, 16, 7, 1, 3, 0, 2, 40, 40), "large_mean": (False, 64, 64, 3, 1, 1, 0, 6, 48, 48),
        "box_ragged": (False, 64, 64, 3, 1, 1, 0, 228, 46, 47),
        "stem_box": (False, 6, 16, 7, 1, 3, 0, 30, 96, 96)}[case]        # the stem kernel: z from it, statistics from the stand-alone pass      # the LDS-resident-box kernel with masked last tile rows / columns
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k)) / np.sqrt(cin * k * k)
    x = torch.randn(N, cin, H, W)
    bias, gamma, beta = torch.randn(cout) * 0.3, torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    if case == "large_mean":
        bias = bias + 30.0 * torch.sign(torch.randn(cout))
    rm0, rv0 = torch.randn(cout) * 0.1, torch.rand(cout) + 0.5
    eps, mom = 1e-5, 0.1
    z64 = (F.conv_transpose2d(_rb(x), _rb(w), None, s, p, op) if tr else F.conv2d(_rb(x), _rb(w), None, s, p)) + bias.double().view(1, -1, 1, 1)
    rows = z64.numel() // cout
    mean = z64.mean(dim=(0, 2, 3))
    var = z64.var(dim=(0, 2, 3), unbiased=False)
    g = ConvGeom(int(tr), cin, cout, k, k, s, s, p, p, op, op, ACT_NONE)
    layer = bf16.ConvB(g, w.to(cuda))
    Ho, Wo = layer.out_hw(H, W)
    Cp = bf16.round8(cout)
    xb = _nhwc(x).to(cuda)
    zb = torch.full((N, Ho, Wo, Cp), 3.0, dtype=torch.bfloat16, device=cuda)
    rm, rv = rm0.clone().to(cuda), rv0.clone().to(cuda)
    out = [torch.full((Cp,), 7.0, device=cuda) for _ in range(4)]
    layer.run_bn(bf16.ActB(xb, 0, cin), bf16.ActB(zb, 0, cout), bias.to(cuda), gamma.to(cuda), beta.to(cuda), eps, mom, rm, rv, *out)
    torch.cuda.synchronize()
    got_z = zb[..., :cout].permute(0, 3, 1, 2).double().cpu()
    S = float(z64.abs().max())
    assert float(((got_z - z64).abs() - z64.abs() / 128).max()) <= 2e-5 * S
    m_, r_, sc_, sh_ = [t.double().cpu() for t in out]
    sd = float(var.sqrt().mean())
    # the fused sums see the fp32 accumulators, the stand-alone reduction the bf16-rounded z: both within bf16 rounding of a mean
    # over `rows` values (relative 2^-9 / sqrt(rows) for the mean, 2^-8 for the variance), written as tolerances
    assert float((m_[:cout] - mean).abs().max()) <= 1e-3 * sd + 2.0 ** -9 * float(mean.abs().max())
    rstd = 1.0 / torch.sqrt(var + eps)
    # against the exact conv: rounding z to bf16 adds quantisation noise of (ulp(|mean|) / sqrt(12))^2 to the variance
    q = (2.0 ** -8 * float(mean.abs().max())) ** 2 / 12.0 / float(var.min())
    assert float(((r_[:cout] - rstd).abs() / rstd).max()) <= 2e-3 + q
    # against float64 statistics of the z that was stored: one definition, whatever the launch shape
    zs = zb[..., :cout].double().cpu().reshape(-1, cout)
    mean_s, var_s = zs.mean(dim=0), zs.var(dim=0, unbiased=False)
    assert float((m_[:cout] - mean_s).abs().max()) <= 1e-5 * sd + 2e-6 * float(mean_s.abs().max())
    assert float(((r_[:cout] - 1.0 / torch.sqrt(var_s + eps)).abs() * torch.sqrt(var_s + eps)).max()) <= 1e-3
    assert float((sc_[:cout] - gamma.double() * r_[:cout]).abs().max()) <= 1e-5 * float(sc_.abs().max())
    assert float((sh_[:cout] - (beta.double() - m_[:cout] * sc_[:cout])).abs().max()) <= 1e-5 * (float(sh_.abs().max()) + 1)
    assert bool((m_[cout:] == 0).all()) and bool((sc_[cout:] == 0).all()) and bool((sh_[cout:] == 0).all())
    unb = var * rows / (rows - 1)
    assert float((rm.double().cpu() - ((1 - mom) * rm0.double() + mom * mean)).abs().max()) <= 1e-3 * sd + 1e-3
    assert float(((rv.double().cpu() - ((1 - mom) * rv0.double() + mom * unb)).abs() / rv0.double()).max()) <= 2e-3


@pytest.mark.parametrize("case", ["res64", "plain_relu", "stride2", "ragged72", "deep_splitk", "leaky_y", "box_res", "box_plain"])
def test_data_gradient_launch_reduces_the_batchnorm_backward_sums(case, cuda):
    """w2l_convb_forward_bnbwd: the data-gradient launch of a consumer writes dy of the batch-statistics block in front of it AND
    that block's two BatchNorm-backward column sums (dbeta = sum g, dgamma = sum g * zhat; g = dy * act'(block output), zhat =
    (z - mean) * rstd) from its epilogue.  Checked: dy equals what w2l_convb_forward writes (bit for bit - same launch, same
    epilogue), the sums equal float64 sums over the STORED dy / z / y within fp32-partial accuracy, w2l_bn_train_bwd_apply_bf16
    with them writes the dz (and in-place g) of w2l_bn_train_bwd_bf16, and a split-K launch reports fused = 0 and leaves the sums
    to the caller.  Cases: residual block (mask from y, residual added in the same epilogue), ReLU block without residual (mask
    recomputed from z), the data gradient of a stride-2 conv (a transposed geometry with four phases), a ragged channel count, a
    deep small-spatial layer that splits K, LeakyReLU with y given; "box_*": 64 -> 64 layers large enough for the LDS-resident-box
    kernel (conv_box_bf16.hip, >= 2 048 tiles of 16 x 16 pixels; ragged 120 x 124 extents), whose epilogue takes the same sums."""
    torch.manual_seed({"res64": 1, "plain_relu": 2, "stride2": 3, "ragged72": 4, "deep_splitk": 5, "leaky_y": 6, "box_res": 7, "box_plain": 8}[case])
    # the CONSUMER conv (forward geometry) cin -> cout2 over the block output [N, cin, H, W]; its data gradient maps dz2 -> dx
    cin, cout2, k, s, p, N, H, W, with_res, give_y, bact = {
        "res64": (64, 64, 3, 1, 1, 8, 48, 48, True, True, ACT_RELU), "plain_relu": (64, 128, 3, 1, 1, 10, 48, 48, False, False, ACT_RELU),
        "stride2": (32, 64, 3, 2, 1, 3, 24, 24, False, False, ACT_RELU), "ragged72": (72, 40, 3, 1, 1, 2, 13, 7, False, True, ACT_RELU),
        "deep_splitk": (512, 512, 3, 1, 1, 7, 3, 3, False, False, ACT_RELU), "leaky_y": (16, 32, 5, 1, 2, 3, 17, 9, False, True, ACT_LEAKY),
        "box_res": (64, 64, 3, 1, 1, 32, 120, 124, True, True, ACT_RELU), "box_plain": (64, 64, 3, 1, 1, 32, 128, 128, False, False, ACT_RELU)}[case]
    w = torch.randn(cout2, cin, k, k) / np.sqrt(cin * k * k)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dg = ConvGeom(1, cout2, cin, k, k, s, s, p, p, (H + 2 * p - k) % s, (W + 2 * p - k) % s, ACT_NONE)    # as autograd.NodeB builds it
    layer = bf16.ConvB(dg, w.to(cuda))
    assert layer.out_hw(Ho, Wo) == (H, W)
    Cp = bf16.round8(cin)
    dz2 = _nhwc(torch.randn(N, cout2, Ho, Wo)).to(cuda)
    z = torch.randn(N, cin, H, W) * 1.5 + 0.3
    gamma, beta = torch.rand(cin) + 0.5, torch.randn(cin) * 0.3
    zb = _nhwc(z).to(cuda)
    zs = zb[..., :cin].double().cpu()                                    # the stored z, [N,H,W,C]
    mean = zs.reshape(-1, cin).mean(0)
    rstd = 1.0 / torch.sqrt(zs.reshape(-1, cin).var(0, unbiased=False) + 1e-5)
    scale = gamma.double() * rstd
    shift = beta.double() - mean * scale
    resid_in = torch.randn(N, cin, H, W) if with_res else None           # the block's own residual input (its x)
    yfull = zs * scale + shift + (_rb(resid_in).permute(0, 2, 3, 1) if with_res else 0.0)
    yfull = torch.where(yfull > 0, yfull, (0.01 if bact == ACT_LEAKY else 0.0) * yfull)
    yb = torch.zeros(N, H, W, Cp, dtype=torch.bfloat16)
    yb[..., :cin] = yfull.to(torch.bfloat16)
    yb = yb.to(cuda)

    def vec(t):
        v = torch.zeros(Cp)
        v[:cin] = t.float()
        return v.to(cuda)
    mean_d, rstd_d, scale_d, shift_d = vec(mean), vec(rstd), vec(scale), vec(shift)
    # the consumer's own residual pass-through (its masked dy) added in the same epilogue
    gres = _nhwc(torch.randn(N, cin, H, W)).to(cuda) if with_res else None
    dy_a = torch.full((N, H, W, Cp), 3.0, dtype=torch.bfloat16, device=cuda)
    dy_b = torch.full((N, H, W, Cp), 5.0, dtype=torch.bfloat16, device=cuda)
    A = bf16.ActB
    layer.run(A(dz2, 0, cout2), A(dy_a, 0, cin), A(gres, 0, cin) if with_res else None)
    dgamma, dbeta = torch.full((Cp,), 7.0, device=cuda), torch.full((Cp,), 7.0, device=cuda)
    fused = layer.run_bnbwd(A(dz2, 0, cout2), A(dy_b, 0, cin), A(gres, 0, cin) if with_res else None, A(zb, 0, cin),
                            A(yb, 0, cin) if give_y else None, bact, mean_d, rstd_d, scale_d, shift_d, dgamma, dbeta)
    torch.cuda.synchronize()
    assert torch.equal(dy_a, dy_b), "the launch with sums must write the dy of the plain launch"
    # small grids split K (pickb): the sums then stay with the caller - "leaky_y" is such a shape, "deep_splitk" by construction
    assert fused == {"res64": True, "plain_relu": True, "stride2": True, "ragged72": True, "deep_splitk": False, "leaky_y": fused,
                     "box_res": True, "box_plain": True}[case]
    lib = _lib.load()
    s_ = _lib.current_stream()
    rows = N * H * W
    # reference sums in float64 over the stored tensors
    dys = dy_b[..., :cin].double().cpu()
    ys = yb[..., :cin].double().cpu() if give_y else (zs * scale_d[:cin].double().cpu() + shift_d[:cin].double().cpu())
    neg = {ACT_RELU: 0.0, ACT_LEAKY: 0.01}[bact]
    g = dys * torch.where(ys > 0, torch.ones_like(ys), torch.full_like(ys, neg))
    zh = (zs - mean_d[:cin].double().cpu()) * rstd_d[:cin].double().cpu()
    ref_db, ref_dg = g.reshape(-1, cin).sum(0), (g * zh).reshape(-1, cin).sum(0)
    # the stand-alone path on the same inputs
    dg2, db2 = torch.empty(Cp, device=cuda), torch.empty(Cp, device=cuda)
    dz_ref = torch.zeros(N, H, W, Cp, dtype=torch.bfloat16, device=cuda)
    yptr = _lib.ptr(yb) if give_y else None
    _lib.check(lib.w2l_bn_train_bwd_bf16(s_, rows, Cp, cin, _lib.ptr(dy_b), Cp, yptr, Cp, _lib.ptr(zb), Cp, bact, _lib.ptr(mean_d),
                                         _lib.ptr(rstd_d), _lib.ptr(scale_d), _lib.ptr(shift_d), _lib.ptr(dg2), _lib.ptr(db2),
                                         _lib.ptr(dz_ref), Cp, None, 0), "bn_train_bwd_bf16")
    torch.cuda.synchronize()
    Sb = float(g.abs().reshape(-1, cin).sum(0).max()) + 1e-30          # scale of a column sum: sum |g|
    assert float((db2[:cin].double().cpu() - ref_db).abs().max()) <= 1e-6 * Sb
    if not fused:
        return
    # fp32 per-lane / per-wave partials, fp64 above: a few 1e-7 of sum |g| (|zhat| ~ 1)
    assert float((dbeta[:cin].double().cpu() - ref_db).abs().max()) <= 2e-6 * Sb, "dbeta"
    assert float((dgamma[:cin].double().cpu() - ref_dg).abs().max()) <= 6e-6 * Sb, "dgamma"
    assert bool((dbeta[cin:] == 0).all()) and bool((dgamma[cin:] == 0).all())
    dz_got = torch.zeros(N, H, W, Cp, dtype=torch.bfloat16, device=cuda)
    _lib.check(lib.w2l_bn_train_bwd_apply_bf16(s_, rows, Cp, _lib.ptr(dy_b), Cp, yptr, Cp, _lib.ptr(zb), Cp, bact, _lib.ptr(mean_d),
                                               _lib.ptr(rstd_d), _lib.ptr(scale_d), _lib.ptr(shift_d), _lib.ptr(dgamma),
                                               _lib.ptr(dbeta), _lib.ptr(dz_got), Cp, None, 0), "bn_train_bwd_apply_bf16")
    torch.cuda.synchronize()
    d = (dz_got.double() - dz_ref.double()).abs().cpu()
    # the two dz differ only through the sums (means of g): one bf16 rounding step where a value sits on a boundary
    assert float(d.max()) <= float(dz_ref.double().abs().max()) * 2.0 ** -7
    assert float((d > 0).double().mean()) <= 0.02


@pytest.mark.parametrize("case", ["res64", "plain_relu", "ragged72", "box_res", "box_plain"])
def test_data_gradient_launch_can_store_the_masked_gradient(case, cuda):
    """W2L_BNBWD_STORE_MASKED: the launch that completes a ReLU block's dy stores g = dy * [block output > 0] instead of dy - bit for
    bit what w2l_bn_train_bwd_bf16 writes as its in-place g - with the same two column sums, and the block's backward pass run
    as w2l_bn_train_bwd_apply_bf16(dy = g, y = NULL, act = none) writes bit for bit the dz of the unmasked route.  Residual block
    (mask from y), ReLU block without residual (mask recomputed from z), ragged channel count."""
    torch.manual_seed({"res64": 11, "plain_relu": 12, "ragged72": 14, "box_res": 15, "box_plain": 16}[case])
    cin, cout2, k, N, H, W, with_res, give_y = {"res64": (64, 64, 3, 8, 48, 48, True, True), "plain_relu": (64, 128, 3, 10, 48, 48, False, False),
                                                "ragged72": (72, 40, 3, 2, 13, 7, False, True),
                                                "box_res": (64, 64, 3, 32, 120, 124, True, True),       # conv_box_bf16.hip's epilogue
                                                "box_plain": (64, 64, 3, 32, 128, 128, False, False)}[case]
    w = torch.randn(cout2, cin, k, k) / np.sqrt(cin * k * k)
    dg = ConvGeom(1, cout2, cin, k, k, 1, 1, 1, 1, 0, 0, ACT_NONE)
    layer = bf16.ConvB(dg, w.to(cuda))
    Cp = bf16.round8(cin)
    dz2 = _nhwc(torch.randn(N, cout2, H, W)).to(cuda)
    zb = _nhwc(torch.randn(N, cin, H, W) * 1.5 + 0.3).to(cuda)
    zs = zb[..., :cin].double().cpu()
    gamma, beta = torch.rand(cin) + 0.5, torch.randn(cin) * 0.3
    mean = zs.reshape(-1, cin).mean(0)
    rstd = 1.0 / torch.sqrt(zs.reshape(-1, cin).var(0, unbiased=False) + 1e-5)
    scale = gamma.double() * rstd
    shift = beta.double() - mean * scale
    yfull = zs * scale + shift + (_rb(torch.randn(N, cin, H, W)).permute(0, 2, 3, 1) if with_res else 0.0)
    yb = torch.zeros(N, H, W, Cp, dtype=torch.bfloat16)
    yb[..., :cin] = torch.relu(yfull).to(torch.bfloat16)
    yb = yb.to(cuda)

    def vec(t):
        v = torch.zeros(Cp)
        v[:cin] = t.float()
        return v.to(cuda)
    mean_d, rstd_d, scale_d, shift_d = vec(mean), vec(rstd), vec(scale), vec(shift)
    gres = _nhwc(torch.randn(N, cin, H, W)).to(cuda) if with_res else None
    A = bf16.ActB
    lib, s_, rows = _lib.load(), _lib.current_stream(), N * H * W
    yptr = _lib.ptr(yb) if give_y else None
    out = {}
    for masked in (False, True):
        dy = torch.full((N, H, W, Cp), 3.0, dtype=torch.bfloat16, device=cuda)
        dgamma, dbeta = torch.full((Cp,), 7.0, device=cuda), torch.full((Cp,), 7.0, device=cuda)
        fused = layer.run_bnbwd(A(dz2, 0, cout2), A(dy, 0, cin), A(gres, 0, cin) if with_res else None, A(zb, 0, cin),
                                A(yb, 0, cin) if give_y else None, ACT_RELU, mean_d, rstd_d, scale_d, shift_d, dgamma, dbeta,
                                store_masked=masked)
        assert fused
        dz = torch.zeros(N, H, W, Cp, dtype=torch.bfloat16, device=cuda)
        g = torch.zeros(N, H, W, Cp, dtype=torch.bfloat16, device=cuda)
        if masked:
            _lib.check(lib.w2l_bn_train_bwd_apply_bf16(s_, rows, Cp, _lib.ptr(dy), Cp, None, 0, _lib.ptr(zb), Cp, ACT_NONE, _lib.ptr(mean_d),
                                                       _lib.ptr(rstd_d), _lib.ptr(scale_d), _lib.ptr(shift_d), _lib.ptr(dgamma),
                                                       _lib.ptr(dbeta), _lib.ptr(dz), Cp, None, 0), "bn_train_bwd_apply_bf16")
            g = dy
        else:
            # the unmasked route; g is written out of place here so that both tensors can be compared
            _lib.check(lib.w2l_bn_train_bwd_apply_bf16(s_, rows, Cp, _lib.ptr(dy), Cp, yptr, Cp, _lib.ptr(zb), Cp, ACT_RELU, _lib.ptr(mean_d),
                                                       _lib.ptr(rstd_d), _lib.ptr(scale_d), _lib.ptr(shift_d), _lib.ptr(dgamma),
                                                       _lib.ptr(dbeta), _lib.ptr(dz), Cp, _lib.ptr(g) if give_y else None, Cp),
                       "bn_train_bwd_apply_bf16")
            if not give_y:      # a block without residual writes no g: build it from the mask the kernel rebuilds
                g = torch.where((zb.float() * scale_d + shift_d) > 0, dy, torch.zeros_like(dy))
        torch.cuda.synchronize()
        out[masked] = (g.clone(), dz, dgamma, dbeta)
    for a, b, what in zip(out[False], out[True], ("g", "dz", "dgamma", "dbeta")):
        assert torch.equal(a, b), what
    with pytest.raises(RuntimeError, match="ReLU block only"):
        layer.run_bnbwd(A(dz2, 0, cout2), A(dy, 0, cin), None, A(zb, 0, cin), A(yb, 0, cin), ACT_LEAKY, mean_d, rstd_d, scale_d, shift_d,
                        dgamma, dbeta, store_masked=True)


@pytest.mark.parametrize("cin,cout,npix,act", [(32, 3, 5 * 96 * 96, ACT_SIGMOID), (24, 4, 1237, ACT_NONE), (8, 1, 70001, ACT_RELU),
                                               (32, 2, 9, ACT_LEAKY)])
def test_thin_1x1_row_kernels(cin, cout, npix, act, cuda):
    """w2l_thin1x1_{forward,dgrad,wgrad}_bf16 (the generator's 32 -> 3 output layer in a bf16 training step) against float64 over
    the same bf16-rounded operands: forward within one bf16 rounding of the result, data gradient likewise (with and without the
    accumulate input), weight / bias gradient within 1e-5 of sum |dz x| (fp32 partials of <= a few thousand pixels, fp64 above);
    channel slices wider than the layer, ragged pixel counts, every activation"""
    torch.manual_seed(cin * 100 + cout)
    lib = _lib.load()
    s_ = _lib.current_stream()
    cin8 = bf16.round8(cin)
    xcs, ycs = cin8 + 8, 16                                  # slices of wider buffers
    x = torch.randn(npix, cin)
    w = torch.randn(cout, cin) / np.sqrt(cin)
    b = torch.randn(cout) * 0.3
    xb = torch.full((npix, xcs), 9.0, dtype=torch.bfloat16)
    xb[:, :cin] = x.to(torch.bfloat16)
    xb[:, cin:cin8] = 0
    xb = xb.to(cuda)
    wd, bd = w.to(cuda).contiguous(), b.to(cuda)
    yb = torch.full((npix, ycs), 5.0, dtype=torch.bfloat16, device=cuda)
    _lib.check(lib.w2l_thin1x1_forward_bf16(s_, npix, cin, cout, _lib.ptr(xb), xcs, _lib.ptr(wd), _lib.ptr(bd), act, _lib.ptr(yb), ycs),
               "thin1x1_forward_bf16")
    torch.cuda.synchronize()
    ref = _rb(x) @ _rb(w).t() + b.double()
    if act == ACT_SIGMOID:
        ref = torch.sigmoid(ref)
    elif act == ACT_RELU:
        ref = ref.clamp_min(0)
    elif act == ACT_LEAKY:
        ref = torch.where(ref > 0, ref, 0.01 * ref)
    got = yb[:, :cout].double().cpu()
    S = float(ref.abs().max())
    assert float(((got - ref).abs() - ref.abs() / 128).max()) <= 2e-5 * S
    assert bool((yb[:, cout:8] == 0).all()) and bool((yb[:, 8:] == 5.0).all())
    # data gradient, plain and accumulating
    dz = torch.randn(npix, cout)
    dzb = torch.zeros(npix, 8, dtype=torch.bfloat16)
    dzb[:, :cout] = dz.to(torch.bfloat16)
    dzb = dzb.to(cuda)
    dxb = torch.full((npix, xcs), 7.0, dtype=torch.bfloat16, device=cuda)
    _lib.check(lib.w2l_thin1x1_dgrad_bf16(s_, npix, cin, cout, _lib.ptr(dzb), 8, _lib.ptr(wd), None, 0, _lib.ptr(dxb), xcs),
               "thin1x1_dgrad_bf16")
    torch.cuda.synchronize()
    dref = _rb(dz) @ _rb(w)
    dgot = dxb[:, :cin].double().cpu()
    Sd = float(dref.abs().max())
    assert float(((dgot - dref).abs() - dref.abs() / 128).max()) <= 2e-5 * Sd
    assert bool((dxb[:, cin:cin8] == 0).all()) and bool((dxb[:, cin8:] == 7.0).all())
    prev = dxb.clone()
    _lib.check(lib.w2l_thin1x1_dgrad_bf16(s_, npix, cin, cout, _lib.ptr(dzb), 8, _lib.ptr(wd), _lib.ptr(dxb), xcs, _lib.ptr(dxb), xcs),
               "thin1x1_dgrad_bf16 accumulate")
    torch.cuda.synchronize()
    aref = dref + prev[:, :cin].double().cpu()
    agot = dxb[:, :cin].double().cpu()
    assert float(((agot - aref).abs() - aref.abs() / 128).max()) <= 2e-5 * float(aref.abs().max())
    # weight + bias gradient
    dw, db = torch.full((cout, cin), 3.0, device=cuda), torch.full((cout,), 3.0, device=cuda)
    _lib.check(lib.w2l_thin1x1_wgrad_bf16(s_, npix, cin, cout, _lib.ptr(xb), xcs, _lib.ptr(dzb), 8, _lib.ptr(dw), _lib.ptr(db)),
               "thin1x1_wgrad_bf16")
    torch.cuda.synchronize()
    wref = _rb(dz).t() @ _rb(x)
    Sw = float((_rb(dz).abs().t() @ _rb(x).abs()).max())
    assert float((dw.double().cpu() - wref).abs().max()) <= 1e-5 * Sw
    assert float((db.double().cpu() - _rb(dz).sum(0)).abs().max()) <= 1e-5 * float(_rb(dz).abs().sum(0).max())
    # argument errors are codes with messages
    assert lib.w2l_thin1x1_forward_bf16(s_, npix, 40, cout, _lib.ptr(xb), xcs, _lib.ptr(wd), None, act, _lib.ptr(yb), ycs) == -1
    assert b"thin 1x1 path" in lib.w2l_last_error()
    assert lib.w2l_thin1x1_wgrad_bf16(s_, npix, cin, cout, _lib.ptr(xb), xcs + 4, _lib.ptr(dzb), 8, _lib.ptr(dw), None) == -1


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("N,H,W,act,with_res", [(2048, 16, 16, ACT_RELU, False), (342, 48, 32, ACT_RELU, True), (57, 96, 96, ACT_NONE, True),
                                                (1025, 32, 16, ACT_LEAKY, False), (2051, 16, 16, ACT_RELU, True), (228, 46, 47, ACT_RELU, True),
                                                (700, 15, 30, ACT_NONE, True)])
def test_lds_resident_box_kernel(transposed, N, H, W, act, with_res, cuda):
    """csrc/conv_box_bf16.hip (3x3, stride 1, 64 -> 64, extents divisible by 16: weights and the input box resident in LDS) through
    the same entry point and against the same float64 reference and tolerance as every other bf16 conv: forward geometry and the
    data-gradient (transposed) geometry, image borders (the box halo is zero-filled by out-of-range DMA), one tile and many tiles
    per workgroup, odd tile counts, ragged extents (SyncNet's 46x47, 15x30: the last tile row / column is masked), residual,
    activations.  Every shape is above the launcher's size rule (>= 2048 tiles, >= 85 % tile fill) - smaller ones run the implicit GEMM"""
    _run(cuda, transposed, 64, 64, 3, 1, 1, 0, N, H, W, act=act, with_res=with_res, seed=N + H)


@pytest.mark.parametrize("cin,cout,N,H,W,act", [(6, 16, 30, 96, 96, ACT_RELU), (15, 32, 60, 48, 96, ACT_RELU), (3, 32, 60, 48, 96, ACT_LEAKY),
                                                (15, 32, 64, 46, 90, ACT_NONE), (8, 24, 1025, 16, 16, ACT_SIGMOID)])
def test_stem_7x7_kernel(cin, cout, N, H, W, act, cuda):
    """csrc/conv_stem_bf16.hip (7x7 / stride 1 / pad 3 with 8 or 16 channels per pixel and <= 32 couts: the first layers of the
    generator, SyncNet and the discriminator; weights + the tile's 22x22 input box resident in LDS) through the same entry point and
    against the same float64 reference and tolerance as every other bf16 conv: both channel packings (one tap / two taps per MFMA
    K-step), image borders, ragged extents, pad couts, every activation.  Shapes sit above the launcher's size rule (>= 1024 tiles)"""
    _run(cuda, False, cin, cout, 7, 1, 3, 0, N, H, W, act=act, with_res=False, seed=cin + N)


@pytest.mark.parametrize("transposed,cin,cout,N,H,W,act,with_res", [
    (False, 80, 32, 57, 96, 96, ACT_RELU, False),       # the output block forward: 45 K-steps, 176-byte box rows
    (True, 32, 80, 57, 96, 96, ACT_NONE, False),        # its data gradient: three cout tiles, 80 of 96 couts exist
    (False, 32, 32, 228, 48, 48, ACT_RELU, True),       # a 32 -> 32 residual block, two workgroups per CU
    (True, 32, 32, 228, 48, 48, ACT_NONE, True),        # its data gradient accumulating into the residual path
    (False, 80, 24, 64, 46, 90, ACT_LEAKY, False)])     # ragged extents, pad couts
def test_small_channel_3x3_layers_on_the_resident_box_kernel(transposed, cin, cout, N, H, W, act, with_res, cuda):
    """the KS = 3 instantiations of csrc/conv_stem_bf16.hip (few channels on one side at full resolution: output block, its data
    gradient, the 32 -> 32 residual blocks) through the common entry point against the common float64 reference and tolerance"""
    _run(cuda, transposed, cin, cout, 3, 1, 1, 0, N, H, W, act=act, with_res=with_res, seed=cin + cout)


@pytest.mark.parametrize("cin,cout,k,N,H,W,with_res", [
    (64, 48, 3, 57, 96, 96, True),      # cout_p == 64 but round8(cout) == 48: the 64-wide box kernel must not take it (its epilogue stores 64 channels)
    (60, 40, 3, 57, 96, 96, False),
    (80, 32, 3, 57, 96, 96, True),      # stem families 1 / 2 are instantiated without the residual read: a launch with one stays on the implicit GEMM
    (6, 16, 7, 30, 96, 96, True)])
def test_resident_box_kernels_decline_what_they_do_not_implement(cin, cout, k, N, H, W, with_res, cuda):
    """shapes ABOVE the size rules of conv_box_bf16.hip / conv_stem_bf16.hip that those kernels cannot serve (a narrower cout
    than the 64 channels the box epilogue writes; a residual on the stem families) must give the common reference's result -
    i.e. run on the implicit GEMM - and leave the buffer's pad channels and neighbouring pixels alone"""
    _run(cuda, False, cin, cout, k, 1, k // 2, 0, N, H, W, act=ACT_RELU, with_res=with_res, seed=cin + cout + k)


# ---------------------------------------------------------------- conv_tp2b_bf16.hip: all four phases of a stride-2 transposed layer per workgroup
TP2B_CASES = {
    # name: (cin, cout, N, H, W, act, with_res, affine) - every one launches >= 512 workgroups, so the fused-phase kernel is chosen
    "dec6_160_64": (160, 64, 20, 24, 24, ACT_NONE, False, True),         # models/wav2lip.py:76 at a smaller batch / extent
    "dec5_320_128": (320, 128, 12, 16, 24, ACT_RELU, False, True),       # two cout tiles
    "ragged_96_72": (96, 72, 9, 13, 11, ACT_LEAKY, True, True),          # ragged tiles, cout not a multiple of 32, residual
    "thin_32_16": (32, 16, 40, 24, 24, ACT_NONE, True, False),           # the 32-cout tile half empty: face_encoder_blocks.1.0's data gradient
    "thin_64_32": (64, 32, 24, 16, 32, ACT_NONE, False, False),
}


@pytest.mark.parametrize("case", sorted(TP2B_CASES))
def test_fused_phase_transposed_kernel_forward(case, cuda):
    """3x3 / stride 2 / padding 1 / output padding 1 transposed layers (models/wav2lip.py:60-79) on conv_tp2b_bf16.hip against the
    float64 transposed convolution of the same bf16 operands (tolerance of this file), and against the four-phase implicit GEMM the
    layer ran on before (selected by a tile override): the same math in another summation order - equal within one bf16 step."""
    cin, cout, N, H, W, act, with_res, affine = TP2B_CASES[case]
    _run(cuda, True, cin, cout, 3, 2, 1, 1, N, H, W, act=act, with_res=with_res, affine=affine, seed=hash(case) % 1000)
    # same inputs on both kernels
    torch.manual_seed(5)
    w = torch.randn(cin, cout, 3, 3) / np.sqrt(cin * 9)
    x = torch.randn(N, cin, H, W)
    g = ConvGeom(1, cin, cout, 3, 3, 2, 2, 1, 1, 1, 1, act)
    outs = []
    for tile in (None, 1 if bf16.round8(cout) > 32 else 4):
        layer = bf16.ConvB(g, w.to(cuda))
        if tile is not None:
            layer.set_tile(tile)
        yb = torch.full((N, 2 * H, 2 * W, bf16.round8(cout)), 3.0, dtype=torch.bfloat16, device=cuda)
        layer.run(bf16.ActB(_nhwc(x).to(cuda), 0, cin), bf16.ActB(yb, 0, cout), None, None, None, 0)
        torch.cuda.synchronize()
        outs.append(yb.double().cpu())
    d = (outs[0] - outs[1]).abs()
    assert float(d.max()) <= float(outs[1].abs().max()) * 2.0 ** -7
    assert float((d > 0).double().mean()) <= 0.05


@pytest.mark.parametrize("case", ["dec6_160_64", "dec5_320_128", "ragged_96_72"])
def test_fused_phase_transposed_kernel_batch_statistics(case, cuda):
    """w2l_convb_forward_bn through conv_tp2b_bf16.hip: z as the plain launch writes it and mean / rstd / scale / shift / running
    statistics from the kernel's per-wave partials equal to the statistics of the STORED z (float64), as the implicit GEMM's route"""
    cin, cout, N, H, W, _, _, _ = TP2B_CASES[case]
    torch.manual_seed(7)
    w = torch.randn(cin, cout, 3, 3) / np.sqrt(cin * 9)
    x = torch.randn(N, cin, H, W)
    g = ConvGeom(1, cin, cout, 3, 3, 2, 2, 1, 1, 1, 1, ACT_NONE)
    layer = bf16.ConvB(g, w.to(cuda))
    Cp = bf16.round8(cout)
    xb = _nhwc(x).to(cuda)
    z0 = torch.full((N, 2 * H, 2 * W, Cp), 3.0, dtype=torch.bfloat16, device=cuda)
    z1 = torch.full((N, 2 * H, 2 * W, Cp), 5.0, dtype=torch.bfloat16, device=cuda)
    bias = (torch.randn(cout) * 0.1).to(cuda)
    layer.run(bf16.ActB(xb, 0, cin), bf16.ActB(z0, 0, cout), None, None, bias, 0)
    gamma, beta = (torch.rand(cout) + 0.5).to(cuda), (torch.randn(cout) * 0.3).to(cuda)
    rm, rv = torch.zeros(cout, device=cuda), torch.ones(cout, device=cuda)
    mean, rstd, scale, shift = (torch.full((Cp,), 9.0, device=cuda) for _ in range(4))
    layer.run_bn(bf16.ActB(xb, 0, cin), bf16.ActB(z1, 0, cout), bias, gamma, beta, 1e-5, 0.1, rm, rv, mean, rstd, scale, shift)
    torch.cuda.synchronize()
    assert torch.equal(z0, z1)
    zs = z1[..., :cout].double().cpu().reshape(-1, cout)
    m, v = zs.mean(0), zs.var(0, unbiased=False)
    sd = float(zs.std())
    assert float((mean[:cout].double().cpu() - m).abs().max()) <= 1e-5 * sd + 2e-6 * float(m.abs().max())
    assert float(((rstd[:cout].double().cpu() - 1 / torch.sqrt(v + 1e-5)).abs() * torch.sqrt(v + 1e-5)).max()) <= 1e-3
    assert float((scale[:cout].double().cpu() - gamma.double().cpu() * rstd[:cout].double().cpu()).abs().max()) <= 1e-5 * float(scale.abs().max())
    assert bool((mean[cout:] == 0).all()) and bool((scale[cout:] == 0).all())
    rows = zs.shape[0]
    assert float((rm.double().cpu() - 0.1 * m).abs().max()) <= 1e-3 * sd + 1e-3
    assert float((rv.double().cpu() - (0.9 + 0.1 * v * rows / (rows - 1))).abs().max()) <= 2e-3 * float(v.max()) + 1e-3


@pytest.mark.parametrize("cin,cout,N,H,W", [(16, 32, 40, 48, 48), (64, 128, 16, 24, 32), (128, 256, 40, 12, 12)])
def test_fused_phase_kernel_serves_the_data_gradient_of_stride_2_convs(cin, cout, N, H, W, cuda):
    """The data gradient of `Conv2d(cin, cout, kernel_size=3, stride=2, padding=1)` (models/wav2lip.py:17-30) as autograd.NodeB
    builds it - the weight tensor read as a transposed layer cout -> cin, accumulating into an existing gradient - against float64"""
    torch.manual_seed(cin)
    Ho, Wo = H // 2, W // 2
    w = torch.randn(cout, cin, 3, 3) / np.sqrt(cout * 9)
    dz = torch.randn(N, cout, Ho, Wo)
    acc = torch.randn(N, cin, H, W)
    g = ConvGeom(1, cout, cin, 3, 3, 2, 2, 1, 1, (H + 2 - 3) % 2, (W + 2 - 3) % 2, ACT_NONE)
    layer = bf16.ConvB(g, w.to(cuda))
    assert layer.out_hw(Ho, Wo) == (H, W)
    ref = F.conv_transpose2d(_rb(dz), _rb(w), None, 2, 1, 1) + _rb(acc)
    gx = _nhwc(acc).to(cuda)
    layer.run(bf16.ActB(_nhwc(dz).to(cuda), 0, cout), bf16.ActB(gx, 0, cin), bf16.ActB(gx, 0, cin), None, None, 0)
    torch.cuda.synchronize()
    got = gx[..., :cin].permute(0, 3, 1, 2).double().cpu()
    err, S = (got - ref).abs(), float(ref.abs().max())
    assert int((err > ref.abs() / 128 + 2e-5 * S).sum()) == 0, float(err.max())


@pytest.mark.parametrize("case", ["leaky_5x5_s2", "relu_3x3_res", "deep_splitk"])
def test_data_gradient_launch_can_store_the_activation_gradient_of_a_block_without_batchnorm(case, cuda):
    """w2l_convb_forward_actbwd (the discriminator's conv + LeakyReLU blocks, models/conv.py:22-31): the launch that completes such a
    block's dy stores dz = dy * act'(block output) - bit for bit what the plain launch followed by w2l_act_bwd_bf16 writes; a split-K
    launch reports fused = 0 and stores plain dy."""
    torch.manual_seed({"leaky_5x5_s2": 21, "relu_3x3_res": 22, "deep_splitk": 23}[case])
    # the CONSUMER conv (forward geometry) cin -> cout2 over the block output [N, cin, H, W]; its data gradient maps dz2 -> dy of the block
    cin, cout2, k, s, p, N, H, W, with_res, bact = {"leaky_5x5_s2": (64, 128, 5, 2, 2, 64, 24, 48, False, ACT_LEAKY),
                                                     "relu_3x3_res": (72, 40, 3, 1, 1, 5, 20, 12, True, ACT_RELU),
                                                     "deep_splitk": (512, 512, 3, 1, 1, 7, 3, 3, False, ACT_LEAKY)}[case]
    w = torch.randn(cout2, cin, k, k) / np.sqrt(cin * k * k)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dg = ConvGeom(1, cout2, cin, k, k, s, s, p, p, (H + 2 * p - k) % s, (W + 2 * p - k) % s, ACT_NONE)
    layer = bf16.ConvB(dg, w.to(cuda))
    assert layer.out_hw(Ho, Wo) == (H, W)
    Cp = bf16.round8(cin)
    dz2 = _nhwc(torch.randn(N, cout2, Ho, Wo)).to(cuda)
    yb = _nhwc(torch.randn(N, cin, H, W)).to(cuda)                       # the block's output (sign pattern of the activation)
    gres = _nhwc(torch.randn(N, cin, H, W)).to(cuda) if with_res else None
    A = bf16.ActB
    dy = torch.full((N, H, W, Cp), 3.0, dtype=torch.bfloat16, device=cuda)
    layer.run(A(dz2, 0, cout2), A(dy, 0, cin), A(gres, 0, cin) if with_res else None)
    dz_ref = torch.zeros(N, H, W, Cp, dtype=torch.bfloat16, device=cuda)
    lib = _lib.load()
    _lib.check(lib.w2l_act_bwd_bf16(_lib.current_stream(), N * H * W, Cp, _lib.ptr(dy), Cp, _lib.ptr(yb), Cp, bact, None, _lib.ptr(dz_ref), Cp,
                                    None, 0), "act_bwd_bf16")
    got = torch.full((N, H, W, Cp), 5.0, dtype=torch.bfloat16, device=cuda)
    fused = layer.run_actbwd(A(dz2, 0, cout2), A(got, 0, cin), A(gres, 0, cin) if with_res else None, A(yb, 0, cin), bact)
    torch.cuda.synchronize()
    assert fused == (case != "deep_splitk")
    assert torch.equal(got, dz_ref if fused else dy)
    # with the bias gradient: the same dz, and its column sums (fp32 per-wave partials, fp64 above) against float64 over the stored dz
    got2 = torch.full((N, H, W, Cp), 7.0, dtype=torch.bfloat16, device=cuda)
    db = torch.full((bf16.round8(cin) + 24,), 9.0, device=cuda)
    fused2 = layer.run_actbwd(A(dz2, 0, cout2), A(got2, 0, cin), A(gres, 0, cin) if with_res else None, A(yb, 0, cin), bact, db)
    torch.cuda.synchronize()
    assert fused2 == fused and torch.equal(got2, got)
    if fused:
        ref = dz_ref[..., :cin].double().cpu().reshape(-1, cin)
        S = float(ref.abs().sum(0).max()) + 1e-30
        assert float((db[:cin].double().cpu() - ref.sum(0)).abs().max()) <= 2e-6 * S
        assert bool((db[cin:Cp] == 0).all()) and bool((db[Cp:] == 9.0).all())
    else:
        assert bool((db == 9.0).all())
    with pytest.raises(RuntimeError, match="ReLU / LeakyReLU"):
        layer.run_actbwd(A(dz2, 0, cout2), A(got, 0, cin), None, A(yb, 0, cin), ACT_SIGMOID)
