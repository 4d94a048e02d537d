"""Training path on the GPU, through the C ABI: weight gradient, data gradient (the forward kernel on the transposed
geometry), train-mode BatchNorm forward/backward, activation backward, losses forward+backward, fused Adam — each against
torch autograd on CPU (the arithmetic the reference's loss.backward() runs) — and the three reference training steps
against the golden fixtures frozen from the REAL reference (tests/golden/make_golden_train.py).

Tolerances: per-op gradients <= 2e-4 of the tensor's own L-inf scale (fp32 sums of up to ~10^5 terms in a different
order); losses <= 1e-5 relative.  Whole-network gradients vs the golden: the reference's OWN fp32 result moves by up to
8e-4 (SyncNet gradient norms, 4-sample batch statistics) and 2.6e-2 (discriminator input gradient, L-inf) when the same
graph is evaluated in fp64 (measured with oracle/models_ref.py), so the bounds are 5e-3 on gradient norms, 2e-3 on kept
gradient tensors and 1e-2 on the discriminator's input gradient; measured deviations are 1.4e-3 / 1e-3.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT
from oracle import models_ref
from wav2lip_amd import synthetic as synth

pytestmark = pytest.mark.gpu


def _lib():
    from wav2lip_amd import _lib
    return _lib, _lib.load()


def nhwc(x, pad_to=None):
    """NCHW cpu tensor -> NHWC cuda tensor with channels zero-padded to a multiple of 4"""
    N, Cn, H, W = x.shape
    Cp = pad_to or (Cn + 3) // 4 * 4
    out = torch.zeros(N, H, W, Cp)
    out[..., :Cn] = x.permute(0, 2, 3, 1)
    return out.cuda().contiguous()


def rel_err(got, ref):
    return (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)


# (transposed, cin, cout, k, stride, pad, outpad, H, W)
WGRAD_SIGS = [
    (0, 64, 64, 3, 1, 1, 0, 24, 24), (0, 6, 16, 7, 1, 3, 0, 32, 32), (0, 16, 32, 3, 2, 1, 0, 32, 32),
    (0, 1, 32, 3, 1, 1, 0, 80, 16), (0, 32, 64, 3, (3, 1), 1, 0, 80, 16), (0, 128, 256, 3, (3, 2), 1, 0, 9, 6),
    (0, 256, 512, 3, 1, 0, 0, 3, 3), (0, 512, 512, 1, 1, 0, 0, 1, 1), (0, 80, 32, 3, 1, 1, 0, 20, 20),
    (0, 32, 3, 1, 1, 0, 0, 16, 16), (0, 15, 32, 7, 1, 3, 0, 24, 48), (0, 32, 64, 5, (1, 2), 1, 0, 24, 48),
    (0, 64, 128, 3, 2, 1, 0, 23, 24), (0, 3, 32, 7, 1, 3, 0, 24, 48), (0, 64, 128, 5, 2, 2, 0, 24, 24),
    (0, 512, 1, 1, 1, 0, 0, 1, 1), (0, 384, 384, 3, 1, 1, 0, 6, 6), (0, 256, 64, 1, 1, 0, 0, 8, 8),
    (1, 1024, 512, 3, 1, 0, 0, 1, 1), (1, 160, 64, 3, 2, 1, 1, 12, 12), (1, 768, 384, 3, 2, 1, 1, 3, 3),
    (1, 320, 128, 3, 2, 1, 1, 6, 6),
]


def _conv_ref(sig, N, seed):
    tr, cin, cout, k, s, p, op, H, W = sig
    torch.manual_seed(seed)
    x = torch.randn(N, cin, H, W)
    if tr:
        w = (torch.randn(cin, cout, k, k) / (cin * k * k / 4) ** 0.5).requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        y = F.conv_transpose2d(xr, w, None, stride=s, padding=p, output_padding=op)
    else:
        w = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        y = F.conv2d(xr, w, None, stride=s, padding=p)
    dz = torch.randn_like(y)
    y.backward(dz)
    return x, w.detach(), dz, w.grad, xr.grad


def _geom(sig, act=0):
    _l, _ = _lib()
    tr, cin, cout, k, s, p, op, H, W = sig
    s = s if isinstance(s, tuple) else (s, s)
    return _l.ConvGeom(tr, cin, cout, k, k, s[0], s[1], p, p, op, op, act)


@pytest.mark.parametrize("idx", range(len(WGRAD_SIGS)))
def test_wgrad_matches_autograd(idx, cuda):
    _l, lib = _lib()
    sig = WGRAD_SIGS[idx]
    N = 3
    x, w, dz, dw_ref, _ = _conv_ref(sig, N, idx)
    xg, dzg = nhwc(x), nhwc(dz)
    dw = torch.full(w.shape, float("nan"), device=cuda)
    g = _geom(sig)
    _l.check(lib.w2l_conv_wgrad(C.byref(g), _l.current_stream(), N, sig[7], sig[8], _l.ptr(xg), xg.shape[3], _l.ptr(dzg),
                                dzg.shape[3], _l.ptr(dw)), "wgrad")
    e = rel_err(dw.cpu(), dw_ref)
    assert e <= 2e-4, "wgrad %s: relative error %.3e" % (sig, e)


def test_wgrad_large_k_split_is_deterministic(cuda):
    """many K splits (batch 16 at 48x48) + bit-identical results run to run (fixed-order reduction)"""
    _l, lib = _lib()
    sig = (0, 32, 32, 3, 1, 1, 0, 48, 48)
    x, w, dz, dw_ref, _ = _conv_ref(sig, 16, 5)
    xg, dzg = nhwc(x), nhwc(dz)
    g = _geom(sig)
    outs = []
    for _ in range(2):
        dw = torch.empty(w.shape, device=cuda)
        _l.check(lib.w2l_conv_wgrad(C.byref(g), _l.current_stream(), 16, 48, 48, _l.ptr(xg), 32, _l.ptr(dzg), 32, _l.ptr(dw)),
                 "wgrad")
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0], dw_ref) <= 2e-4


# (cin, cout, H, W, N, x channel stride, dz channel stride): 3x3 / s1 / p1 layers large enough for the Winograd F(3x3,2x2)
# weight-gradient kernel (conv_wino_wgrad.hip): even and odd sizes, ragged channel counts, strided (concat-free) buffers
WINO_WGRAD_SIGS = [
    (64, 64, 24, 24, 4, 64, 64), (80, 32, 21, 19, 6, 80, 32), (32, 96, 48, 48, 2, 32, 96), (96, 160, 13, 13, 12, 96, 160),
    (64, 64, 12, 12, 16, 128, 96), (36, 44, 17, 30, 5, 40, 48), (128, 128, 24, 24, 2, 128, 128), (256, 64, 7, 9, 40, 256, 64),
]


@pytest.mark.parametrize("idx", range(len(WINO_WGRAD_SIGS)))
def test_winograd_wgrad_matches_autograd(idx, cuda):
    _l, lib = _lib()
    cin, cout, H, W, N, xcs, dcs = WINO_WGRAD_SIGS[idx]
    sig = (0, cin, cout, 3, 1, 1, 0, H, W)
    x, w, dz, dw_ref, _ = _conv_ref(sig, N, 100 + idx)
    xg, dzg = nhwc(x, xcs), nhwc(dz, dcs)
    if xcs > cin:
        xg[..., cin:] = 7.0                    # neighbours in a shared buffer: must not leak into the gradient
    if dcs > cout:
        dzg[..., cout:] = -3.0
    dw = torch.full(w.shape, float("nan"), device=cuda)
    g = _geom(sig)
    _l.check(lib.w2l_conv_wgrad(C.byref(g), _l.current_stream(), N, H, W, _l.ptr(xg), xcs, _l.ptr(dzg), dcs, _l.ptr(dw)), "wgrad")
    e = rel_err(dw.cpu(), dw_ref)
    assert e <= 2e-4, "winograd wgrad %s: relative error %.3e" % (WINO_WGRAD_SIGS[idx], e)


@pytest.mark.parametrize("idx", [0, 1, 2, 4, 5, 6, 10, 11, 12, 14, 16, 18, 19, 20])
def test_dgrad_is_the_forward_kernel_on_the_transposed_geometry(idx, cuda):
    from wav2lip_amd import autograd, engine
    _l, lib = _lib()
    sig = WGRAD_SIGS[idx]
    tr, cin, cout, k, s, p, op, H, W = sig
    s2 = s if isinstance(s, tuple) else (s, s)
    N = 2
    x, w, dz, _, dx_ref = _conv_ref(sig, N, 50 + idx)
    if tr:
        dg = _l.ConvGeom(0, cout, cin, k, k, s2[0], s2[1], p, p, 0, 0, 0)
    else:
        dg = _l.ConvGeom(1, cout, cin, k, k, s2[0], s2[1], p, p, (H + 2 * p - k) % s2[0], (W + 2 * p - k) % s2[1], 0)
    wg = w.cuda().contiguous()
    conv = autograd.RawConv(dg, wg, torch.ones(cin, device=cuda), torch.zeros(cin, device=cuda))
    dzg = nhwc(dz)
    cin_p = (cin + 3) // 4 * 4
    out = torch.zeros(N, H, W, cin_p, device=cuda)
    prior = torch.randn(N, H, W, cin_p, device=cuda)
    conv.run(engine.Act(dzg, 0, dzg.shape[3]), engine.Act(out, 0, cin))
    got = out[..., :cin].permute(0, 3, 1, 2).cpu()
    assert got.shape == dx_ref.shape
    assert rel_err(got, dx_ref) <= 2e-4
    # accumulate into an existing gradient through the residual input, in place
    acc = prior.clone()
    conv.run(engine.Act(dzg, 0, dzg.shape[3]), engine.Act(acc, 0, cin), engine.Act(acc, 0, cin))
    got2 = (acc - prior)[..., :cin].permute(0, 3, 1, 2).cpu()
    assert rel_err(got2, dx_ref) <= 4e-4
    # re-pack after a weight change (w2l_conv_update)
    conv.update(weight=(2.0 * wg))
    conv.run(engine.Act(dzg, 0, dzg.shape[3]), engine.Act(out, 0, cin))
    assert rel_err(out[..., :cin].permute(0, 3, 1, 2).cpu(), 2.0 * dx_ref) <= 2e-4


@pytest.mark.parametrize("tile", [6, 7])
def test_dgrad_winograd_path(tile, cuda, monkeypatch):
    """the data gradient of a 3x3 / stride 1 / pad 1 conv through the Winograd kernel (weights read flipped + transposed)"""
    from wav2lip_amd import autograd, engine
    monkeypatch.setattr(engine, "AUTOTUNE", False)     # the forced configuration below must be the one that runs
    _l, lib = _lib()
    cin, cout, H, W, N = 64, 128, 22, 18, 3
    sig = (0, cin, cout, 3, 1, 1, 0, H, W)
    x, w, dz, _, dx_ref = _conv_ref(sig, N, 77)
    dg = _l.ConvGeom(1, cout, cin, 3, 3, 1, 1, 1, 1, 0, 0, 0)
    conv = autograd.RawConv(dg, w.cuda().contiguous(), torch.ones(cin, device=cuda), torch.zeros(cin, device=cuda))
    _l.check(lib.w2l_conv_set_tile(conv.handle, tile), "set_tile")
    dzg = nhwc(dz)
    g = torch.randn(N, H, W, cin, device=cuda)
    out = torch.zeros(N, H, W, cin, device=cuda)
    conv.run(engine.Act(dzg, 0, cout), engine.Act(out, 0, cin), engine.Act(g, 0, cin))
    got = (out - g).permute(0, 3, 1, 2).cpu()
    assert rel_err(got, dx_ref) <= 4e-4
    conv.update(weight=(-1.0 * w).cuda().contiguous())
    conv.run(engine.Act(dzg, 0, cout), engine.Act(out, 0, cin))
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), -dx_ref) <= 2e-4


@pytest.mark.parametrize("shape", [(4, 16, 12, 10), (2, 384, 5, 7), (3, 512, 1, 1), (2, 64, 33, 31), (5, 128, 9, 6)])
@pytest.mark.parametrize("residual", [False, True])
def test_bn_train_forward_backward(shape, residual, cuda):
    _l, lib = _lib()
    N, Cn, H, W = shape
    torch.manual_seed(N * 1000 + Cn)
    z = (torch.randn(shape) * 1.7 + 0.3).requires_grad_(True)
    gamma = torch.empty(Cn).uniform_(0.5, 1.5).requires_grad_(True)
    beta = (torch.randn(Cn) * 0.2).requires_grad_(True)
    res = torch.randn(shape).requires_grad_(True)
    rm, rv = torch.randn(Cn) * 0.1, torch.empty(Cn).uniform_(0.5, 1.5)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(z, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if residual:
        y = y + res
    y = F.relu(y)
    dy = torch.randn(shape)
    y.backward(dy)
    rows = N * H * W
    s = _l.current_stream()
    zg, dyg = nhwc(z.detach()), nhwc(dy)
    resg = nhwc(res.detach())
    dev = lambda t: t.detach().clone().cuda()
    gam, bet, rmg, rvg = dev(gamma), dev(beta), dev(rm), dev(rv)
    mean, rstd, scale, shift = (torch.empty(Cn, device=cuda) for _ in range(4))
    _l.check(lib.w2l_bn_train_stats(s, rows, Cn, _l.ptr(zg), Cn, _l.ptr(gam), _l.ptr(bet), 1e-5, 0.1, _l.ptr(rmg), _l.ptr(rvg),
                                    _l.ptr(mean), _l.ptr(rstd), _l.ptr(scale), _l.ptr(shift)), "stats")
    yg = torch.empty_like(zg)
    _l.check(lib.w2l_affine_act(s, rows, Cn, _l.ptr(zg), Cn, _l.ptr(scale), _l.ptr(shift), _l.ptr(resg) if residual else None,
                                Cn, 1, _l.ptr(yg), Cn), "affine_act")
    assert rel_err(yg.permute(0, 3, 1, 2).cpu(), y.detach()) <= 1e-5
    assert rel_err(rmg.cpu(), rm_ref) <= 1e-5 and rel_err(rvg.cpu(), rv_ref) <= 1e-5
    dgam, dbet = torch.empty(Cn, device=cuda), torch.empty(Cn, device=cuda)
    dzg = torch.empty_like(zg)
    _l.check(lib.w2l_bn_train_bwd(s, rows, Cn, _l.ptr(dyg), Cn, _l.ptr(yg), Cn, _l.ptr(zg), Cn, 1, _l.ptr(mean), _l.ptr(rstd),
                                  _l.ptr(scale), _l.ptr(dgam), _l.ptr(dbet), _l.ptr(dzg), Cn, _l.ptr(dyg) if residual else None,
                                  Cn), "bn_bwd")
    assert rel_err(dzg.permute(0, 3, 1, 2).cpu(), z.grad) <= 2e-4
    assert rel_err(dgam.cpu(), gamma.grad) <= 2e-4 and rel_err(dbet.cpu(), beta.grad) <= 2e-4
    if residual:   # dy was overwritten with the masked gradient = the residual branch's gradient
        assert rel_err(dyg.permute(0, 3, 1, 2).cpu(), res.grad) <= 1e-6


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_act_bwd_col_sum_add_rows(act, cuda):
    _l, lib = _lib()
    torch.manual_seed(act)
    N, Cn, H, W = 3, 32, 7, 5
    pre = torch.randn(N, Cn, H, W, requires_grad=True)
    y = [pre, F.relu(pre), torch.sigmoid(pre), F.leaky_relu(pre, 0.01)][act]
    sc = torch.empty(Cn).uniform_(0.5, 2.0)
    dy = torch.randn(N, Cn, H, W)
    (y * 1.0).backward(dy)
    ref = pre.grad * sc.view(1, -1, 1, 1)
    rows = N * H * W
    s = _l.current_stream()
    yg, dyg = nhwc(y.detach()), nhwc(dy)
    dz = torch.empty_like(yg)
    g_out = torch.empty_like(yg)
    _l.check(lib.w2l_act_bwd(s, rows, Cn, _l.ptr(dyg), Cn, _l.ptr(yg), Cn, act, _l.ptr(sc.cuda()), _l.ptr(dz), Cn,
                             _l.ptr(g_out), Cn), "act_bwd")
    assert rel_err(dz.permute(0, 3, 1, 2).cpu(), ref) <= 1e-6
    assert rel_err(g_out.permute(0, 3, 1, 2).cpu(), pre.grad) <= 1e-6
    cs = torch.empty(Cn, device=cuda)
    _l.check(lib.w2l_col_sum(s, rows, Cn, _l.ptr(dz), Cn, _l.ptr(cs)), "col_sum")
    assert rel_err(cs.cpu(), ref.sum(dim=(0, 2, 3))) <= 1e-5
    out = torch.empty_like(dz)
    _l.check(lib.w2l_add_rows(s, rows, Cn, _l.ptr(dz), Cn, _l.ptr(g_out), Cn, _l.ptr(out), Cn), "add_rows")
    assert torch.equal(out, dz + g_out)


def test_losses_forward_backward(cuda):
    from wav2lip_amd import autograd, losses
    torch.manual_seed(0)
    # L1
    a = torch.rand(2, 3, 5, 24, 24, requires_grad=True)
    b = torch.rand(2, 3, 5, 24, 24)
    (F.l1_loss(a, b) * 0.7).backward()
    ag = a.detach().cuda().requires_grad_(True)
    l = losses.L1Loss()(ag, b.cuda())
    (l * 0.7).backward()
    assert abs(l.item() - F.l1_loss(a, b).item()) <= 1e-6
    assert rel_err(ag.grad.cpu(), a.grad) <= 1e-6
    # cosine + BCE on post-ReLU style embeddings, through F.normalize
    ea = torch.rand(6, 512, requires_grad=True)
    ev = torch.rand(6, 512, requires_grad=True)
    y = torch.tensor([[1.], [0.], [1.], [1.], [0.], [0.]])
    ref = F.binary_cross_entropy(F.cosine_similarity(F.normalize(ea, p=2, dim=1), F.normalize(ev, p=2, dim=1)).unsqueeze(1), y)
    (ref * 1.3).backward()
    eag, evg = ea.detach().cuda().requires_grad_(True), ev.detach().cuda().requires_grad_(True)
    got = losses.cosine_loss(autograd.L2NormRows.apply(eag), autograd.L2NormRows.apply(evg), y.cuda())
    (got * 1.3).backward()
    assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert rel_err(eag.grad.cpu(), ea.grad) <= 2e-4 and rel_err(evg.grad.cpu(), ev.grad) <= 2e-4
    # BCE
    p = torch.rand(10, 1).clamp(0.02, 0.98).requires_grad_(True)
    t = torch.tensor([1., 0.] * 5).view(10, 1)
    refb = F.binary_cross_entropy(p, t)
    refb.backward()
    pg = p.detach().cuda().requires_grad_(True)
    gotb = losses.BCELoss()(pg, t.cuda())
    gotb.backward()
    assert abs(gotb.item() - refb.item()) <= 1e-6
    assert rel_err(pg.grad.cpu(), p.grad) <= 1e-5


def _golden_train():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_train_v1.npz"))


def _golden_train_f64():
    """fp64 companions of the gradient tensors golden_train_v1.npz keeps in fp32 (tests/golden/make_golden_train_f64.py)"""
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_train_f64_v1.npz"))


def test_adam_matches_torch_optim_golden(cuda):
    from wav2lip_amd import optim
    g = _golden_train()
    params = [torch.nn.Parameter(torch.from_numpy(g["adam_p0/%d" % i]).cuda()) for i in range(3)]
    opt = optim.Adam(params, lr=1e-3, betas=(0.5, 0.999))
    for s in range(3):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(g["adam_g%d/%d" % (s, i)]).cuda()
        opt.step()
    for i, p in enumerate(params):
        ref = torch.from_numpy(g["adam_p3/%d" % i])
        assert (p.detach().cpu() - ref).abs().max().item() <= 2e-6, i
    sd = opt.state_dict()
    assert int(sd["state"][0]["step"]) == 3 and sd["state"][2]["exp_avg"].shape == (20000,)


def _check_grads(tag, model, g, bn_bias_names=True, direct=5e-3, kept=1e-2):
    """Gradient norms of every parameter against the reference golden (`direct` relative bound) and, because small-batch
    train-mode BatchNorm makes the reference's own fp32 gradients inexact, against the fp64 evaluation of the same graph:
    the HIP path's relative distance to fp64 must stay within 3x the reference's own (max and median over parameters)."""
    names = [str(n) for n in g[tag + "_grad_names"]]
    norms, norms64 = g[tag + "_grad_norms"], g[tag + "_grad_norms64"]
    g64 = _golden_train_f64()
    named = dict(model.named_parameters())
    assert sorted(named) == names
    wnorm = {n: v for n, v in zip(names, norms)}
    ours_e, ref_e = [], []
    for n, ref, r64 in zip(names, norms, norms64):
        got = float(named[n].grad.double().norm())
        if bn_bias_names and n.endswith("conv_block.0.bias"):
            # a conv bias in front of a BatchNorm has zero gradient in exact arithmetic: the reference holds rounding
            # noise there, the HIP path returns exact zeros
            scale = wnorm[n.replace("conv_block.0.bias", "conv_block.0.weight")]
            assert got <= 1e-4 * scale + 1e-6 and ref <= 1e-4 * scale + 1e-6, (n, got, ref, scale)
            continue
        e = abs(got - ref) / (ref + 1e-12)
        ours_e.append(abs(got - r64) / (r64 + 1e-12))
        ref_e.append(abs(ref - r64) / (r64 + 1e-12))
        # never tighter than the golden is itself: a value within 3x the reference's own distance to fp64 may sit 4x that
        # distance from the fp32 golden
        bound = max(direct, 4 * ref_e[-1])
        assert e <= bound, "%s: |grad| %.6e vs reference %.6e (bound %.1e)" % (n, got, ref, bound)
    assert max(ours_e) <= 3 * max(ref_e) + 1e-4, (max(ours_e), max(ref_e))
    assert np.median(ours_e) <= 3 * np.median(ref_e) + 1e-5, (np.median(ours_e), np.median(ref_e))
    for key in g.files:
        if key.startswith(tag + "_grad/"):
            n = key[len(tag) + 6:]
            if bn_bias_names and n.endswith("conv_block.0.bias"):
                continue   # exact zero here vs rounding noise in the reference (see above)
            e = rel_err(named[n].grad.cpu(), torch.from_numpy(g[key]))
            # the golden tensor's own L-inf distance to the fp64 evaluation of the same graph (3e-6 .. 3e-2 here: few-sample
            # BatchNorm statistics) is the yardstick; `kept` is the floor
            yard = rel_err(torch.from_numpy(g[key]).double(), torch.from_numpy(g64[tag + "_grad64/" + n]))
            bound = max(kept, 4 * yard)
            assert e <= bound, "%s: relative error %.3e (bound %.1e, the reference's own fp32 is %.1e from fp64)" % (n, e, bound, yard)
    return max(ours_e), max(ref_e)


def test_syncnet_train_step_matches_reference_golden(cuda):
    from wav2lip_amd import losses, models
    g = _golden_train()
    S = models.SyncNet_color()
    S.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in S.state_dict().items()}, seed=2))
    S = S.to(cuda).train()
    x = torch.from_numpy(synth.sync_faces(4, seed=11)).to(cuda)
    mel = torch.from_numpy(synth.mel_windows(4, seed=11)).unsqueeze(1).to(cuda)
    y = torch.tensor([[1.], [0.], [1.], [0.]], device=cuda)
    a, v = S(mel, x)
    loss = losses.cosine_loss(a, v, y)
    loss.backward()
    assert np.abs(a.detach().cpu().numpy() - g["sync_a"]).max() <= 1e-5
    assert np.abs(v.detach().cpu().numpy() - g["sync_v"]).max() <= 1e-5
    assert abs(loss.item() - float(g["sync_loss"])) <= 1e-5 * float(g["sync_loss"])
    _check_grads("sync", S, g)
    sd = S.state_dict()
    assert np.abs(sd["face_encoder.0.conv_block.1.running_mean"].cpu().numpy() - g["sync_running_mean/face_encoder.0"]).max() <= 1e-6
    assert np.abs(sd["face_encoder.0.conv_block.1.running_var"].cpu().numpy() - g["sync_running_var/face_encoder.0"]).max() <= 1e-6
    assert int(sd["face_encoder.0.conv_block.1.num_batches_tracked"]) == 101


def _gen_inputs(B=2):
    r = np.random.default_rng(21)
    T = 5
    gt = torch.from_numpy(r.uniform(0, 1, (B, 3, T, 96, 96)).astype(np.float32))
    wrong = torch.from_numpy(r.uniform(0, 1, (B, 3, T, 96, 96)).astype(np.float32))
    masked = gt.clone()
    masked[:, :, :, 48:] = 0.
    xin = torch.cat([masked, wrong], dim=1)
    indiv = torch.from_numpy(r.uniform(-4, 4, (B, T, 1, 80, 16)).astype(np.float32))
    melw = torch.from_numpy(r.uniform(-4, 4, (B, 1, 80, 16)).astype(np.float32))
    return xin, indiv, melw, gt


def _load(cls, seed, cuda):
    m = cls()
    m.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed))
    return m.to(cuda)


def test_generator_train_step_matches_reference_golden(cuda):
    """wav2lip_train.py:211-229 with the frozen, train-mode SyncNet: losses, output and every gradient norm"""
    from wav2lip_amd import losses, models
    g = _golden_train()
    G = _load(models.Wav2Lip, 0, cuda).train()
    S = _load(models.SyncNet_color, 2, cuda)
    for p in S.parameters():
        p.requires_grad = False
    xin, indiv, melw, gt = (t.to(cuda) for t in _gen_inputs(4))
    out = G(indiv, xin)
    sync = losses.get_sync_loss(S, melw, out)
    l1 = losses.l1_loss(out, gt)
    loss = 0.03 * sync + 0.97 * l1
    loss.backward()
    assert np.abs(out.detach()[:, :, 0, ::8, ::8].cpu().numpy() - g["gen_out_t0"]).max() <= 1e-5
    assert abs(out.detach().double().mean().item() - float(g["gen_out_mean"])) <= 1e-6
    assert abs(l1.item() - float(g["gen_l1"])) <= 1e-5 * float(g["gen_l1"])
    assert abs(sync.item() - float(g["gen_sync"])) <= 1e-3 * float(g["gen_sync"])   # 2-sample batch statistics inside
    assert abs(loss.item() - float(g["gen_loss"])) <= 1e-5 * float(g["gen_loss"])
    # 4-sample batch statistics in the frozen SyncNet and 20-sample ones at the generator's 1x1 bottleneck: the reference's
    # own fp32 gradients are 0.8 % (norms) / 7 % (L-inf) away from fp64 here, hence the loose direct bounds
    _check_grads("gen", G, g, direct=3e-2, kept=0.15)
    rv = G.state_dict()["output_block.0.conv_block.1.running_var"].cpu().numpy()
    assert np.abs(rv - g["gen_running_var/output_block.0"]).max() <= 1e-6
    assert all(p.grad is None for p in S.parameters())
    # the backward pass walks only what somebody wants a gradient from: the frozen expert's face encoder (the gradient flows to
    # the generated frames) but not its audio encoder (frozen parameters, a mel that needs no gradient) - as torch's engine decides
    ran = [n for lst in S._train_graphs.graphs.values() for gr in lst for n in gr.backward_nodes]
    assert ran and all(n.startswith("face_encoder") for n in ran), ran
    ran_g = [n for lst in G._train_graphs.graphs.values() for gr in lst for n in gr.backward_nodes]
    assert any(n.startswith("audio_encoder") for n in ran_g) and any(n.startswith("face_encoder_blocks.0") for n in ran_g)


def test_disc_steps_match_reference_golden(cuda):
    """hq_wav2lip_train.py:233 (perceptual loss, gradient w.r.t. the fake frames) and :247-254 (D real / D fake)"""
    from wav2lip_amd import losses, models
    g = _golden_train()
    D = _load(models.Wav2Lip_disc_qual, 4, cuda).train()
    fake = torch.from_numpy(synth.disc_frames(1, 5, seed=31)).to(cuda).requires_grad_(True)
    real = torch.from_numpy(synth.disc_frames(1, 5, seed=32)).to(cuda)
    perc = D.perceptual_forward(fake)
    perc.backward()
    assert abs(perc.item() - float(g["disc_perceptual"])) <= 1e-5
    assert float(fake.grad[:, :, :, :48].abs().max()) == 0.0
    # a gradient of magnitude 4e-5 through five-sample BatchNorms: the REAL reference's fp32 result is itself 2.9e-2 (L-inf,
    # relative) away from its fp64 result (tests/golden/make_golden_train_f64.py), so the fp32 golden pins a summation order, not
    # the math.  Yardstick form, as in _check_grads: our distance to fp64 within 3x the reference's own.
    g64 = torch.from_numpy(_golden_train_f64()["disc_perceptual_dfake64"])
    yard = rel_err(torch.from_numpy(g["disc_perceptual_dfake"]).double(), g64)
    ours = rel_err(fake.grad[:, :, :, 48::4, ::4].cpu().double(), g64)
    assert 1e-2 <= yard <= 5e-2, yard
    assert ours <= 3 * yard, "perceptual-loss gradient: %.3e from fp64, the reference's own fp32 is %.3e away (bound 3x)" % (ours, yard)
    assert abs(float(fake.grad.double().norm()) - float(g["disc_perceptual_dfake_norm"])) <= 5e-3 * float(g["disc_perceptual_dfake_norm"])
    D.zero_grad()
    pred = D(real)
    lr = losses.bce_mean(pred, torch.ones((len(pred), 1), device=cuda))
    lr.backward()
    pred = D(fake.detach())
    lf = losses.bce_mean(pred, torch.zeros((len(pred), 1), device=cuda))
    lf.backward()
    assert abs(lr.item() - float(g["disc_real_loss"])) <= 1e-5 and abs(lf.item() - float(g["disc_fake_loss"])) <= 1e-5
    _check_grads("disc", D, g, bn_bias_names=False)


def test_two_live_forwards_and_stale_backward(cuda):
    """D(real) and D(fake) recorded before either backward get separate buffer sets"""
    from wav2lip_amd import losses, models
    D = _load(models.Wav2Lip_disc_qual, 4, cuda).train()
    a = torch.from_numpy(synth.disc_frames(1, 2, seed=1)).to(cuda)
    b = torch.from_numpy(synth.disc_frames(1, 2, seed=2)).to(cuda)
    la = losses.bce_mean(D(a), torch.ones(2, 1, device=cuda))
    lb = losses.bce_mean(D(b), torch.zeros(2, 1, device=cuda))
    (la + lb).backward()
    g_both = [p.grad.clone() for p in D.parameters()]
    D.zero_grad()
    losses.bce_mean(D(a), torch.ones(2, 1, device=cuda)).backward()
    losses.bce_mean(D(b), torch.zeros(2, 1, device=cuda)).backward()
    for p, gb in zip(D.parameters(), g_both):
        assert rel_err(p.grad, gb) <= 1e-5


class _forced_relu_masks:
    """Context manager: the i-th F.relu call inside returns z * masks[i] (and records z).  With the masks taken from the HIP
    path's own activations ([y > 0]) the oracle graph is evaluated on exactly the linear piece of the network the HIP path
    differentiated: a unit whose pre-activation is within rounding of zero may legitimately land on either side of the ReLU
    kink in two fp32 evaluations, and ONE such flip in a late layer moves the input gradient by 0.3 - 4 % (measured on this
    very input: face_encoder.12 holds a unit at |z| = 5e-6 rms) - comparing across different pieces tests luck, not arithmetic."""

    def __init__(self, masks):
        self.masks, self.seen = masks, []

    def __enter__(self):
        self.relu = F.relu
        it = iter(self.masks)

        def relu(z, *a, **k):
            m = next(it)
            self.seen.append(z.detach())
            return z * m.to(z.dtype)
        F.relu = relu
        return self

    def __exit__(self, *a):
        F.relu = self.relu


def test_eval_mode_syncnet_propagates_data_gradient_only(cuda):
    """an .eval() expert (BN folded) inside a loss (wav2lip_train.py:187-198 with a frozen expert): loss and the gradient
    w.r.t. its face input vs the fp64 evaluation of the oracle graph ON THE SAME ReLU PIECE (see _forced_relu_masks), the
    ReLU decisions themselves vs the oracle's wherever they are not within rounding of the kink, and no parameter grads"""
    from wav2lip_amd import losses, models
    S = _load(models.SyncNet_color, 2, cuda).eval()
    sd = {k: v.cpu() for k, v in S.state_dict().items()}
    x = torch.from_numpy(synth.sync_faces(3, seed=5))
    mel = torch.from_numpy(synth.mel_windows(3, seed=5)).unsqueeze(1)
    y = torch.ones(3, 1)
    xg = x.to(cuda).requires_grad_(True)
    a, v = S(mel.to(cuda), xg)
    l = losses.cosine_loss(a, v, y.to(cuda))
    l.backward()
    graph = [g for lst in S._train_graphs.graphs.values() for g in lst][0]
    masks = [(n.y.view() > 0).cpu() for n in graph.nodes]           # face encoder blocks, then audio encoder blocks
    assert [n.name.split(".")[0] for n in graph.nodes] == ["face_encoder"] * 17 + ["audio_encoder"] * 14

    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in sd.items()}
    xr = x.double().requires_grad_(True)
    with _forced_relu_masks(masks) as fm:
        ao, vo = models_ref.syncnet_graph(sd64, mel.double(), xr, training=False)
        lo = models_ref.cosine_loss(ao, vo, y.double())
    lo.backward()
    assert len(fm.seen) == len(masks)
    # the HIP path's ReLU decisions differ from the exact ones only where the exact pre-activation is within fp32 rounding
    # of zero (1e-4 of the layer's rms is ~100x the measured forward error), and only in a handful of units
    flips = 0
    for z, m in zip(fm.seen, masks):
        dis = (z > 0) != m
        if bool(dis.any()):
            flips += int(dis.sum())
            assert float(z[dis].abs().max()) <= 1e-4 * float(z.pow(2).mean().sqrt()), "a ReLU decision differs far from the kink"
    assert flips <= 50, flips
    assert abs(l.item() - lo.item()) <= 1e-5 * abs(lo.item())
    assert rel_err(xg.grad.cpu().double(), xr.grad) <= 2e-4
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in S.parameters())


def test_training_steps_run_and_learn(cuda):
    """two optimiser steps of each reference loop on a fixed batch: finite losses, parameters move, loss goes down"""
    from wav2lip_amd import models, optim, train
    torch.manual_seed(0)
    S = models.SyncNet_color().to(cuda)
    optS = optim.Adam([p for p in S.parameters() if p.requires_grad], lr=1e-3)
    x = torch.from_numpy(synth.sync_faces(8, seed=9)).to(cuda)
    mel = torch.from_numpy(synth.mel_windows(8, seed=9)).unsqueeze(1).to(cuda)
    y = torch.tensor([[1.], [0.]] * 4, device=cuda)
    ls = [train.syncnet_train_step(S, optS, x, mel, y).item() for _ in range(6)]
    assert all(np.isfinite(ls)) and min(ls[1:]) < ls[0], ls
    G = models.Wav2Lip().to(cuda)
    D = models.Wav2Lip_disc_qual().to(cuda)
    for p in S.parameters():
        p.requires_grad = False
    optG = optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999))
    optD = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
    xin, indiv, melw, gt = (t.to(cuda) for t in _gen_inputs())   # B=2: train-mode BN needs >1 value per channel
    w0 = G.output_block[1].weight.detach().clone()
    r = [train.hq_train_step(G, D, S, optG, optD, xin, indiv, melw, gt, syncnet_wt=0.03, disc_wt=0.07) for _ in range(3)]
    vals = [[float(v.detach()) for v in step.values()] for step in r]
    assert np.isfinite(np.asarray(vals)).all(), vals
    assert float((G.output_block[1].weight.detach() - w0).abs().max()) > 0
    l = [train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03)[0].item() for _ in range(2)]
    assert all(np.isfinite(l))
    gt2 = torch.empty_like(gt)           # a learnable target (uniform noise has no signal: L1 sits at its floor of 0.25)
    gt2[:, 0], gt2[:, 1], gt2[:, 2] = 0.9, 0.1, 0.6
    l1 = [train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt2, syncnet_wt=0.0)[1].item() for _ in range(6)]
    assert all(np.isfinite(l1)) and min(l1[1:]) < l1[0], l1     # pure reconstruction on a fixed batch must go down
    # eval after training: the inference plan sees the updated weights
    G.eval()
    with torch.no_grad():
        out = G(indiv, xin)
    assert out.shape == (2, 3, 5, 96, 96) and bool(torch.isfinite(out).all())


def test_checkpoint_resume_reproduces_the_next_step(cuda, tmp_path, monkeypatch):
    """save (model + fused-Adam state) after one step, resume in fresh objects, take the second step: bit-identical
    parameters (launch autotuning off, so both runs use the same launch configurations = the same summation orders; every
    kernel on the path is deterministic)"""
    from wav2lip_amd import checkpoint, engine, models, optim, train
    monkeypatch.setattr(engine, "AUTOTUNE", False)
    torch.manual_seed(1)
    x = torch.from_numpy(synth.sync_faces(4, seed=9)).to(cuda)
    mel = torch.from_numpy(synth.mel_windows(4, seed=9)).unsqueeze(1).to(cuda)
    y = torch.tensor([[1.], [0.], [0.], [1.]], device=cuda)
    S = _load(models.SyncNet_color, 2, cuda)
    opt = optim.Adam([p for p in S.parameters() if p.requires_grad], lr=1e-3)
    train.syncnet_train_step(S, opt, x, mel, y)
    path = checkpoint.save_checkpoint(S, opt, 1, str(tmp_path), 0)
    train.syncnet_train_step(S, opt, x, mel, y)
    S2 = models.SyncNet_color().to(cuda)
    opt2 = optim.Adam([p for p in S2.parameters() if p.requires_grad], lr=1e-3)
    _, step, _ = checkpoint.load_checkpoint(path, S2, opt2)
    assert step == 1
    train.syncnet_train_step(S2, opt2, x, mel, y)
    for (n, a), (_, b) in zip(S.state_dict().items(), S2.state_dict().items()):
        assert torch.equal(a, b), n


def test_grad_reducer_path_returns_the_same_gradients(cuda):
    """backward with a GradReducer attached (bucket flatten + views; world size 1, so no collective) == plain backward"""
    from wav2lip_amd import losses, models
    from wav2lip_amd.sharding import GradReducer

    class OneRank:
        @staticmethod
        def get_world_size():
            return 1

    D = _load(models.Wav2Lip_disc_qual, 4, cuda).train()
    x = torch.from_numpy(synth.disc_frames(1, 3, seed=8)).to(cuda)
    losses.bce_mean(D(x), torch.ones(3, 1, device=cuda)).backward()
    plain = [p.grad.clone() for p in D.parameters()]
    D.zero_grad()
    red = GradReducer(OneRank(), bucket_bytes=1 << 20).attach(D)
    losses.bce_mean(D(x), torch.ones(3, 1, device=cuda)).backward()
    for p, g in zip(D.parameters(), plain):
        assert p.grad.shape == g.shape and rel_err(p.grad, g) <= 1e-6
    assert red._inflight == [] and red._open == []
    GradReducer.detach(D)


def test_hq_step_with_one_reducer_on_both_networks_and_frame_gather_equals_plain_step(cuda):
    """hq_wav2lip_train step (hq_wav2lip_train.py:221-257) with ONE GradReducer attached to generator AND discriminator and the
    optional frame all-gather switched on, on a one-rank world (the collectives are identities): identical parameters after
    the step as the plain step.  The two-rank arithmetic of the same protocol runs under gloo in tests/test_train_host.py."""
    from wav2lip_amd import models, optim, train
    from wav2lip_amd.sharding import GradReducer

    class OneRank:
        @staticmethod
        def get_world_size():
            return 1

        @staticmethod
        def all_gather_into_tensor(out, t):
            out.copy_(t)

    xin, indiv, melw, gt = (t.to(cuda) for t in _gen_inputs(2))
    results = []
    for with_reducer in (False, True):
        G = _load(models.Wav2Lip, 0, cuda)
        D = _load(models.Wav2Lip_disc_qual, 4, cuda)
        S = _load(models.SyncNet_color, 2, cuda)
        for p in S.parameters():
            p.requires_grad = False
        optG = optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
        optD = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
        red = None
        if with_reducer:
            red = GradReducer(OneRank(), bucket_bytes=4 << 20).attach(G, D)
        out = train.hq_train_step(G, D, S, optG, optD, xin, indiv, melw, gt, syncnet_wt=0.03, disc_wt=0.07,
                                  gather_frames=OneRank() if with_reducer else None, return_generated=True)
        if red is not None:
            assert red._inflight == [] and red._open == []
            GradReducer.detach(G, D)
        results.append(([p.detach().clone() for p in G.parameters()], [p.detach().clone() for p in D.parameters()],
                        {k: float(v) for k, v in out.items() if k != "g"}))
    (g0, d0, l0), (g1, d1, l1) = results
    assert l0 == l1
    for a, b in zip(g0 + d0, g1 + d1):
        assert torch.equal(a, b)          # same kernels, same configurations, same order: bit-identical


# ---------------------------------------------------------------- bf16 matrix-core contraction (W2L_PREC_BF16)
def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("idx", [0, 1, 2, 4, 5, 6, 8, 10, 11, 12, 14, 18, 19, 20, 21])
def test_bf16_conv_equals_fp32_conv_of_bf16_rounded_operands(idx, cuda):
    """forward of every layer shape with the bf16 kernel == torch fp32 conv of the bf16-rounded input and weight (products
    of bf16 values are exact in fp32, so only the summation order differs): tolerance as for the fp32 kernel"""
    from wav2lip_amd import autograd, engine
    _l, lib = _lib()
    sig = WGRAD_SIGS[idx]
    tr, cin, cout, k, s, p, op, H, W = sig
    N = 2
    torch.manual_seed(300 + idx)
    x = torch.randn(N, cin, H, W)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.randn(wshape) / (cin * k * k) ** 0.5
    bias = torch.randn(cout) * 0.1
    xr, wr = _bf16_round(x), _bf16_round(w)
    if tr:
        ref = F.conv_transpose2d(xr, wr, bias, stride=s, padding=p, output_padding=op)
    else:
        ref = F.conv2d(xr, wr, bias, stride=s, padding=p)
    ref = F.relu(ref)
    conv = autograd.RawConv(_geom(sig, act=1), w.cuda().contiguous(), torch.ones(cout, device=cuda), bias.cuda(), "bf16c")
    xg = nhwc(x)
    Ho, Wo = ref.shape[2], ref.shape[3]
    cp = (cout + 3) // 4 * 4
    y = torch.zeros(N, Ho, Wo, cp, device=cuda)
    conv.run(engine.Act(xg, 0, xg.shape[3]), engine.Act(y, 0, cout))
    got = y[..., :cout].permute(0, 3, 1, 2).cpu()
    assert rel_err(got, ref) <= 2e-4, rel_err(got, ref)
    full = F.relu(F.conv_transpose2d(x, w, bias, stride=s, padding=p, output_padding=op) if tr else F.conv2d(x, w, bias, stride=s, padding=p))
    assert rel_err(got, full) <= 3e-2        # and within bf16 rounding of the exact fp32 layer


@pytest.mark.parametrize("idx", range(len(WGRAD_SIGS)))
def test_bf16_wgrad_equals_fp32_wgrad_of_bf16_rounded_operands(idx, cuda):
    _l, lib = _lib()
    sig = WGRAD_SIGS[idx]
    tr, cin, cout, k, s, p, op, H, W = sig
    N = 3
    torch.manual_seed(400 + idx)
    x = _bf16_round(torch.randn(N, cin, H, W))
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.zeros(wshape, requires_grad=True)
    y = F.conv_transpose2d(x, w, None, stride=s, padding=p, output_padding=op) if tr else F.conv2d(x, w, None, stride=s, padding=p)
    dz = _bf16_round(torch.randn_like(y))
    y.backward(dz)
    xg, dzg = nhwc(x), nhwc(dz)
    dw = torch.full(wshape, float("nan"), device=cuda)
    g = _geom(sig)
    _l.check(lib.w2l_conv_wgrad_prec(C.byref(g), _l.current_stream(), N, H, W, _l.ptr(xg), xg.shape[3], _l.ptr(dzg),
                                     dzg.shape[3], _l.ptr(dw), 1), "wgrad bf16")
    e = rel_err(dw.cpu(), w.grad)
    assert e <= 2e-4, "bf16 wgrad %s: relative error %.3e" % (sig, e)


class _Jitter(torch.autograd.Function):
    """y = x * (1 + eps * u), u ~ U(-1, 1) per element, forward AND backward (independent draws): what storing a tensor and its
    gradient in a format with relative rounding error eps does to an exact graph"""

    @staticmethod
    def forward(ctx, x, eps, gen):
        ctx.eps, ctx.gen = eps, gen
        return x * (1 + eps * (2 * torch.rand(x.shape, generator=gen, dtype=x.dtype) - 1))

    @staticmethod
    def backward(ctx, g):
        return g * (1 + ctx.eps * (2 * torch.rand(g.shape, generator=ctx.gen, dtype=g.dtype) - 1)), None, None


class _bf16_storage_noise:
    """Context manager: inside, every F.conv2d / F.conv_transpose2d of the (fp64) oracle graph sees its input, its weight and
    its output - and, on the way back, their gradients - jittered by eps = 2**-8 relative, the rounding error bound of bf16
    (8 significant bits).  It is the bf16-storage path's error model applied to the exact graph: the spread of the results over
    seeds measures how far bf16 rounding ALONE can move a loss or a gradient of this test case - the yardstick the HIP path is
    held to, instead of a fitted percentage."""

    def __init__(self, seed, eps=2.0 ** -8):
        self.eps, self.gen = eps, torch.Generator().manual_seed(seed)

    def __enter__(self):
        self.c, self.ct = F.conv2d, F.conv_transpose2d
        eps, gen = self.eps, self.gen

        def wrap(fn):
            def f(x, w, b=None, **kw):
                return _Jitter.apply(fn(_Jitter.apply(x, eps, gen), _Jitter.apply(w, eps, gen), b, **kw), eps, gen)
            return f
        F.conv2d, F.conv_transpose2d = wrap(self.c), wrap(self.ct)

    def __exit__(self, *a):
        F.conv2d, F.conv_transpose2d = self.c, self.ct


def _bf16_yardstick_check(what, named_params, groups, g64, noisy, loss_hip, loss64, noisy_losses):
    """every parameter gradient of the HIP bf16 path against the fp64 oracle: relative L2 distance of the TENSOR (direction and
    length) and relative distance of its norm.  Per parameter GROUP (the blocks of one resolution: their gradients share the
    conditioning of the graph behind them) the worst and the median distance must stay within 3x the worst / median distance the
    bf16 error model produced in that group over all seeds (floor 2**-8: one bf16 rounding) - a per-parameter yardstick from a
    handful of seeds would be a max of few samples and flake.  Every assertion message carries the bound that was applied."""
    rows = {}
    for n, p in named_params:
        if n.endswith("conv_block.0.bias") or p.grad is None or n not in g64:
            continue
        ref = g64[n]
        rn = float(ref.norm()) + 1e-30
        got = p.grad.detach().double().cpu()
        rows[n] = ((float((got - ref).norm()) / rn, abs(float(got.norm()) - rn) / rn),
                   [(float((g[n] - ref).norm()) / rn, abs(float(g[n].norm()) - rn) / rn) for g in noisy])
    assert len(rows) > 10
    lines = []
    for grp in groups:
        ns = [n for n in rows if n.startswith(grp)]
        if not ns:
            continue
        for k, kind in ((0, "tensor"), (1, "norm")):
            ours = np.array([rows[n][0][k] for n in ns])
            yards = np.array([[rows[n][1][sd_][k] for n in ns] for sd_ in range(len(noisy))])     # [seed][param]
            bmax = 3 * max(float(yards.max()), 2.0 ** -8)
            bmed = 3 * max(float(np.median(yards, axis=1).max()), 2.0 ** -8)
            line = "%s / %s (%d gradients) %s distance to fp64: worst %.3e (bound %.3e), median %.3e (bound %.3e)" % (
                what, grp, len(ns), kind, ours.max(), bmax, np.median(ours), bmed)
            lines.append(line)
            assert ours.max() <= bmax and np.median(ours) <= bmed, line
    print("\n".join(lines))
    lb = 3 * max(max(abs(l - loss64) for l in noisy_losses), 2.0 ** -8 * abs(loss64))
    assert abs(loss_hip - loss64) <= lb, "%s: loss %.6f vs fp64 %.6f, bound %.3e" % (what, loss_hip, loss64, lb)


def test_bf16_training_step_tracks_the_fp32_step(cuda):
    """The bf16-STORAGE training path (BASELINE configs[3] / [4]; wav2lip_amd/autograd.py NodeB, csrc/conv_bf16.hip,
    wgrad_bf16.hip, train_bf16.hip) on a SyncNet step (color_syncnet_train.py:155-165, train-mode BatchNorm, batch 8) and on a
    generator L1 step (wav2lip_train.py:220-231, 6 frames): loss and EVERY parameter gradient against the fp64 evaluation of the
    oracle graph, bounded per parameter group by a MEASURED bf16 yardstick - 3x the largest distance of three / four fp64 evaluations
    with 2**-8 relative noise injected at every conv input, weight and output and at their gradients (_bf16_storage_noise) -
    instead of round 2's "median 5 % / max 50 %"."""
    from wav2lip_amd import engine, losses, models
    B = 16
    S = _load(models.SyncNet_color, 2, cuda).train()
    sd = {k: v.detach().cpu().clone() for k, v in S.state_dict().items()}
    x = torch.from_numpy(synth.sync_faces(B, seed=21))
    mel = torch.from_numpy(synth.mel_windows(B, seed=21)).unsqueeze(1)
    y = torch.tensor([[1.], [0.]] * (B // 2))

    def graph_sd(dt):
        osd = {}
        for k, v in sd.items():
            t = v.clone()
            if t.is_floating_point():
                t = t.to(dt)
                if "running_" not in k:
                    t.requires_grad_(True)
            osd[k] = t
        return osd

    def sync_oracle():
        osd = graph_sd(torch.float64)
        a, v = models_ref.syncnet_graph(osd, mel.double(), x.double(), training=True)
        loss = models_ref.cosine_loss(a, v, y.double())
        loss.backward()
        return loss.item(), {k: t.grad.detach().clone() for k, t in osd.items() if t.requires_grad and t.grad is not None}

    l64, g64 = sync_oracle()
    noisy = []
    for seed in (1, 2, 3, 4):
        with _bf16_storage_noise(seed):
            noisy.append(sync_oracle())
    engine.set_train_precision("bf16")
    try:
        a, v = S(mel.to(cuda), x.to(cuda))
        loss = losses.cosine_loss(a, v, y.to(cuda))
        loss.backward()
        assert any(isinstance(n, __import__("wav2lip_amd.autograd", fromlist=["NodeB"]).NodeB)
                   for lst in S._train_graphs.graphs.values() for gph in lst for n in gph.nodes), "the bf16-storage graph did not run"
        _bf16_yardstick_check("SyncNet step, batch %d" % B, list(S.named_parameters()), ["face_encoder", "audio_encoder"], g64,
                              [g for _, g in noisy], loss.item(), l64, [l for l, _ in noisy])

        # ---- generator, L1 only, 4-D call on 6 frames
        torch.manual_seed(9)
        G = _load(models.Wav2Lip, 0, cuda).train()
        sd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
        Bg = 6
        face, melg, gt = torch.rand(Bg, 6, 96, 96), torch.rand(Bg, 1, 80, 16) * 8 - 4, torch.rand(Bg, 3, 96, 96)

        def gen_oracle():
            osd = graph_sd(torch.float64)
            out = models_ref.wav2lip_graph(osd, melg.double(), face.double(), training=True)
            loss = F.l1_loss(out, gt.double())
            loss.backward()
            return loss.item(), {k: t.grad.detach().clone() for k, t in osd.items() if t.requires_grad and t.grad is not None}

        l64, g64 = gen_oracle()
        noisy = []
        for seed in (5, 6, 7):
            with _bf16_storage_noise(seed):
                noisy.append(gen_oracle())
        out = G(melg.to(cuda), face.to(cuda))
        loss = losses.l1_loss(out, gt.to(cuda))
        loss.backward()
        ggroups = (["output_block"] + ["face_decoder_blocks.%d" % i for i in range(6, -1, -1)] +
                   ["face_encoder_blocks.%d" % i for i in range(7)] + ["audio_encoder"])
        _bf16_yardstick_check("generator L1 step, %d frames" % Bg, list(G.named_parameters()), ggroups, g64, [g for _, g in noisy],
                              loss.item(), l64, [l for l, _ in noisy])
    finally:
        engine.set_train_precision("f32")


@pytest.mark.parametrize("kind", ["res", "convT", "nonorm", "stem"])
def test_bf16_block_gradients_are_tight_when_well_conditioned(kind, cuda):
    """One block in train mode over thousands of pixels per channel (BatchNorm statistics are then well conditioned, unlike the
    few-sample bottleneck of the whole networks above): forward, input gradient and every parameter gradient of the bf16-storage
    path against the fp64 evaluation of the same block, held to 3x the bf16 error model's own distance - which here is a few
    per cent, so this is the tight check of the NodeB arithmetic (conv, batch statistics, residual, ReLU mask, dgrad, wgrad)."""
    from wav2lip_amd import engine
    from wav2lip_amd.models.conv import Conv2d, Conv2dTranspose, nonorm_Conv2d
    torch.manual_seed({"res": 1, "convT": 2, "nonorm": 3, "stem": 4}[kind])
    if kind == "res":
        blk, cin, H, W = Conv2d(64, 64, 3, 1, 1, residual=True), 64, 24, 24
    elif kind == "convT":
        blk, cin, H, W = Conv2dTranspose(96, 32, 3, 2, 1, 1), 96, 12, 12
    elif kind == "nonorm":
        blk, cin, H, W = nonorm_Conv2d(32, 64, 5, (1, 2), 2), 32, 24, 24
    else:
        blk, cin, H, W = Conv2d(6, 16, 7, 1, 3), 6, 48, 48
    N = 4
    x = torch.randn(N, cin, H, W)
    blk = blk.train()
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    conv_w, conv_b = sd["conv_block.0.weight"], sd["conv_block.0.bias"]

    def oracle():
        w, b = conv_w.double().requires_grad_(True), conv_b.double().requires_grad_(True)
        xi = x.double().requires_grad_(True)
        params = {"conv_block.0.weight": w, "conv_block.0.bias": b}
        if kind == "convT":
            z = F.conv_transpose2d(xi, w, b, stride=2, padding=1, output_padding=1)
        elif kind == "nonorm":
            z = F.conv2d(xi, w, b, stride=(1, 2), padding=2)
        elif kind == "stem":
            z = F.conv2d(xi, w, b, stride=1, padding=3)
        else:
            z = F.conv2d(xi, w, b, stride=1, padding=1)
        if kind == "nonorm":
            y = F.leaky_relu(z, 0.01)
        else:
            g, be = sd["conv_block.1.weight"].double().requires_grad_(True), sd["conv_block.1.bias"].double().requires_grad_(True)
            params.update({"conv_block.1.weight": g, "conv_block.1.bias": be})
            zn = F.batch_norm(z, None, None, g, be, True, 0.1, 1e-5)
            y = F.relu(zn + xi) if kind == "res" else F.relu(zn)
        loss = (y * torch.linspace(0.5, 1.5, y.numel(), dtype=torch.float64).view(y.shape)).sum() / y.numel()
        loss.backward()
        grads = {k: t.grad.detach().clone() for k, t in params.items()}
        grads["input"] = xi.grad.detach().clone()
        return y.detach(), grads

    y64, g64 = oracle()
    noisy = []
    for seed in (1, 2, 3):
        with _bf16_storage_noise(seed):
            noisy.append(oracle())
    engine.set_train_precision("bf16")
    try:
        m = blk.to(cuda)
        xg = x.to(cuda).requires_grad_(True)
        y = m(xg)
        wgt = torch.linspace(0.5, 1.5, y.numel(), device=cuda).view(y.shape)
        ((y * wgt).sum() / y.numel()).backward()
    finally:
        engine.set_train_precision("f32")
    fy = float((y.detach().double().cpu() - y64).norm() / y64.norm())
    by = 3 * max(max(float((yn - y64).norm() / y64.norm()) for yn, _ in noisy), 2.0 ** -8)
    assert fy <= by, "forward: distance %.3e, bound %.3e" % (fy, by)
    got = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()}
    got["input"] = xg.grad.detach().double().cpu()
    lines = []
    for n, ref in g64.items():
        if n == "conv_block.0.bias" and kind != "nonorm":
            continue                       # exactly zero in front of batch statistics (rounding noise in torch)
        rn = float(ref.norm())
        d = float((got[n] - ref).norm()) / rn
        yard = max(float((gn[n] - ref).norm()) / rn for _, gn in noisy)
        lines.append("%s %s: distance to fp64 %.3e, bf16 error model %.3e, bound %.3e" % (kind, n, d, yard, 3 * max(yard, 2.0 ** -8)))
        assert d <= 3 * max(yard, 2.0 ** -8), lines[-1]
        assert yard <= 0.15, "the case is meant to be well conditioned: " + lines[-1]
    print("\n".join(lines))


class _perturbed_convs:
    """Context manager: every F.conv2d / F.conv_transpose2d result inside is multiplied by (1 + eps * N(0,1)) - a model of
    what ANY fp32 evaluation of the graph does to the exact one (a different summation order, Winograd, FMA contraction move
    each conv output by about one fp32 ulp = 2**-23 relative).  Evaluated in fp64 it measures how far fp32 rounding alone
    can move a result: the conditioning of the test case, independent of any implementation."""

    def __init__(self, eps, seed):
        self.eps, self.gen = eps, torch.Generator().manual_seed(seed)

    def __enter__(self):
        self.c, self.ct = F.conv2d, F.conv_transpose2d
        eps, gen = self.eps, self.gen

        def wrap(fn):
            def f(x, w, b=None, **kw):
                y = fn(x, w, b, **kw)
                return y * (1 + eps * torch.randn(y.shape, generator=gen, dtype=y.dtype))
            return f
        F.conv2d, F.conv_transpose2d = wrap(self.c), wrap(self.ct)

    def __exit__(self, *a):
        F.conv2d, F.conv_transpose2d = self.c, self.ct


def test_generator_4d_train_step_against_the_oracle_graph(cuda):
    """4-D call (B,6,96,96)/(B,1,80,16) in train mode, odd batch, L1 only (wav2lip_train.py:220-231 without the sync term):
    output, loss, running statistics and every gradient norm.

    Three samples at the generator's 1x1 bottleneck make this graph ill-conditioned - BatchNorm over 3 values amplifies a
    one-ulp change of a conv output into percent-level changes of the encoder gradients - so a fixed tolerance against the
    fp32 CPU oracle is a coin toss (the oracle's own fp32 gradients are 0.4 % median / 1.2 % max away from its fp64
    evaluation).  The gradient check is therefore anchored to the fp64 evaluation of the oracle graph, per parameter group,
    and the yardstick is measured, not fitted: the HIP path may be as far from fp64 as 3x the largest of (i) the fp32 CPU
    oracle's own distance and (ii) the distances of the fp64 graph with fp32-sized relative noise (2**-22, 2**-20) injected at
    every conv output.  The well-conditioned tail (output block) must be tight in absolute terms."""
    from wav2lip_amd import losses, models
    torch.manual_seed(7)
    G = _load(models.Wav2Lip, 0, cuda).train()
    sd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    B = 3
    face = torch.rand(B, 6, 96, 96)
    mel = torch.rand(B, 1, 80, 16) * 8 - 4
    gt = torch.rand(B, 3, 96, 96)

    def oracle(dt):
        osd = {}
        for k, v in sd.items():
            t = v.clone()
            if t.is_floating_point():
                t = t.to(dt)
                if "running_" not in k:
                    t.requires_grad_(True)
            osd[k] = t
        out = models_ref.wav2lip_graph(osd, mel.to(dt), face.to(dt), training=True)
        loss = F.l1_loss(out, gt.to(dt))
        loss.backward()
        return osd, out.detach(), loss.item(), {k: float(v.grad.double().norm()) for k, v in osd.items() if v.requires_grad}

    osd, ref, lref, g32 = oracle(torch.float32)
    _, ref64, lref64, g64 = oracle(torch.float64)
    # injected relative noise: 2**-22 ~ a reordered fp32 sum, 2**-20 ~ the forward error of the fp32 Winograd kernels (the conv
    # tests bound every HIP layer at 1e-4 of its output scale and measure ~1e-6)
    ginj = []
    for eps, seed in ((2.0 ** -22, 1), (2.0 ** -22, 2), (2.0 ** -20, 3), (2.0 ** -20, 4)):
        with _perturbed_convs(eps, seed):
            ginj.append(oracle(torch.float64)[3])

    out = G(mel.to(cuda), face.to(cuda))
    loss = losses.l1_loss(out, gt.to(cuda))
    loss.backward()
    assert out.shape == (B, 3, 96, 96)
    # forward: against the fp32 oracle and, tighter in spirit, no further from fp64 than 3x the fp32 oracle is
    oerr = (out.detach().cpu() - ref).abs().max().item()
    assert oerr <= 2e-4, oerr
    o64 = (out.detach().cpu().double() - ref64).abs().max().item()
    r64 = (ref.double() - ref64).abs().max().item()
    assert o64 <= 3 * r64 + 2e-5, (o64, r64)
    assert abs(loss.item() - lref) <= 1e-5 * lref and abs(loss.item() - lref64) <= 1e-5 * lref64

    names = [n for n, _ in G.named_parameters() if not n.endswith("conv_block.0.bias")]
    got = {n: float(p.grad.double().norm()) for n, p in G.named_parameters()}

    def dist(g, n):
        return abs(g[n] - g64[n]) / (g64[n] + 1e-12)

    groups = ["output_block"] + ["face_decoder_blocks.%d" % i for i in range(6, -1, -1)] + ["face_encoder_blocks", "audio_encoder"]
    report = []
    for grp in groups:
        ns = [n for n in names if n.startswith(grp)]
        ours = np.array([dist(got, n) for n in ns])
        yards = [np.array([dist(g, n) for n in ns]) for g in [g32] + ginj]
        ymax, ymed = max(y_.max() for y_ in yards), max(np.median(y_) for y_ in yards)
        report.append((grp, ours.max(), ymax))
        msg = ("%s: worst gradient-norm distance to fp64 %.3e against the applied bound %.3e (= 3 x %.3e, the largest of the fp32 "
               "CPU oracle's own distance and the 2^-22 / 2^-20 injected-noise runs); median %.3e against %.3e"
               % (grp, ours.max(), 3 * ymax + 1e-5, ymax, np.median(ours), 3 * ymed + 1e-5))
        assert ours.max() <= 3 * ymax + 1e-5, msg
        assert np.median(ours) <= 3 * ymed + 1e-5, msg
    # the well-conditioned tail of the network (gradients that never pass the 3-sample bottleneck) must be tight in absolute terms
    assert dict((g, o) for g, o, _ in report)["output_block"] <= 5e-4, report
    # conv biases in front of a batch-statistics BatchNorm: exact zero here, rounding noise in the reference
    for n, p in G.named_parameters():
        if n.endswith("conv_block.0.bias"):
            wn = got[n.replace("conv_block.0.bias", "conv_block.0.weight")]
            assert got[n] <= 1e-4 * wn + 1e-6, (n, got[n], wn)
    rm = G.state_dict()["face_decoder_blocks.3.1.conv_block.1.running_mean"].cpu()
    assert (rm - osd["face_decoder_blocks.3.1.conv_block.1.running_mean"]).abs().max().item() <= 1e-5
