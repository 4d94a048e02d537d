"""The three training steps AT THE BASELINE LAUNCH SHAPES (BASELINE.json configs[2..4]: SyncNet batch 512; generator and hq
GAN step batch 64 x 5 frames) on the HIP path against the golden frozen from the REAL reference modules under torch autograd
(tests/golden/make_golden_train_baseline.py; reference color_syncnet_train.py:155-165, wav2lip_train.py:220-231,
hq_wav2lip_train.py:221-256).  The launch table and the bf16 tile rule resolve other (tile, split-K) configurations at N = 512 /
320 than at the N <= 16 of tests/test_train_gpu.py: these are the configurations tools/train_bench.py times.

What is compared (the golden holds, per parameter gradient, its L2 norm and four sketches <g, r_k> with fixed +-1 vectors, from
the fp32 reference, from the fp64 evaluation of the oracle graph and from fp64 evaluations under the bf16-storage error model):
  fp32 path   losses <= 1e-4 relative to the reference's (the cosine loss through a frozen train-mode SyncNet: 1e-3);
              network outputs <= 2e-5; every gradient's TENSOR distance to fp64 (direction and length, estimated from the
              sketches) within 3x the reference's own fp32 distance per parameter group, worst and median (floor 1e-4; the
              reference itself is 0.3 % - 3 % from fp64 at these shapes: the graphs stay ill-conditioned through their train-mode
              BatchNorms); the NORM distances within 3x the reference's over the whole network, worst and median; every norm
              within max(1e-3, 4x the reference's own norm distance) of the reference's
  bf16 path   losses within 3x the bf16 error model's spread; tensor distances per group and norm distances over the network
              within 3x the error model's (floor 2^-8)
No CPU graph runs here: a step at these shapes takes the CPU oracle minutes, the GPU milliseconds."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from wav2lip_amd import synthetic as synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(ROOT, "tests", "golden", "golden_train_baseline_v1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _load(cls, seed, cuda):
    m = cls()
    m.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed))
    return m.to(cuda)


def _batch(cfg, B, cuda):
    return {k: torch.from_numpy(v).to(cuda) for k, v in synth.train_batch(cfg, B, 5).items()}


def _summaries(model, names):
    """norm and sketches of every parameter gradient, on the device, float64"""
    named = dict(model.named_parameters())
    assert sorted(named) == list(names), "state-dict surface differs from the golden's"
    norms, sk = np.zeros(len(names)), np.zeros((len(names), 4))
    for i, n in enumerate(names):
        g = named[n].grad
        assert g is not None, n
        g = g.detach().double().reshape(-1)
        r = torch.from_numpy(synth.sketch_vectors(n, g.numel())).to(g.device).double()
        norms[i] = float(g.norm())
        sk[i] = (r @ g).cpu().numpy()
    return norms, sk


def _dist(norms, sk, norms64, sk64):
    """(tensor distance estimate, norm distance) to the fp64 gradients, relative to their norm"""
    rn = norms64 + 1e-300
    return np.sqrt(((sk - sk64) ** 2).mean(axis=1)) / rn, np.abs(norms - norms64) / rn


def _check_gradients(what, gold, tag, net, model, groups, precision):
    names = [str(n) for n in gold["%s_%s_names" % (tag, net)]]
    norms, sk = _summaries(model, names)
    n64, s64 = gold["%s_%s_norms64" % (tag, net)], gold["%s_%s_sketch64" % (tag, net)]
    ours = _dist(norms, sk, n64, s64)
    if precision == "f32":
        yards = [_dist(gold["%s_%s_norms" % (tag, net)], gold["%s_%s_sketch" % (tag, net)], n64, s64)]
        floor = 1e-4
    else:
        yards = [_dist(gold["%s_%s_norms_noise%d" % (tag, net, s)], gold["%s_%s_sketch_noise%d" % (tag, net, s)], n64, s64)
                 for s in range(int(gold[tag + "_noise_seeds"]))]
        floor = 2.0 ** -8
    # a conv bias in front of a BatchNorm has zero gradient in exact arithmetic (rounding noise in torch, exact zeros here)
    live = np.array([not (n.endswith("conv_block.0.bias") and (n[:-len("0.bias")] + "1.weight") in names) and n64[i] > 0
                     for i, n in enumerate(names)])
    lines, failures, covered = [], [], np.zeros(len(names), bool)
    for grp in groups:
        idx = np.array([i for i, n in enumerate(names) if n.startswith(grp) and live[i]], dtype=int)
        if idx.size == 0:
            continue
        covered[idx] = True
        for k, kind in ((0, "tensor"), (1, "norm")):
            o = ours[k][idx]
            y = np.array([yd[k][idx] for yd in yards])
            bmax = 3 * max(float(y.max()), floor)
            bmed = 3 * max(float(np.median(y, axis=1).max()), floor)
            line = "%s / %s %s (%d gradients) %s distance to fp64: worst %.3e (bound %.3e), median %.3e (bound %.3e)" % (
                what, net, grp, idx.size, kind, o.max(), bmax, np.median(o), bmed)
            lines.append(line)
            # the TENSOR distance (direction and length) is held per group.  The NORM distance - a much better conditioned number:
            # ~1e-3 here where the tensors are 2e-2 apart - of one fp32 evaluation against another's is a coin toss inside a group
            # of 3..12 gradients (measured: the HIP path's group medians sit 0.2x .. 3.8x the reference's); it is held over the
            # whole network below, as tests/test_train_gpu.py does at small batches, and reported per group
            if k == 0 and not (o.max() <= bmax and np.median(o) <= bmed):
                failures.append(line)
    o = ours[1][live]
    y = np.array([yd[1][live] for yd in yards])
    bmax, bmed = 3 * max(float(y.max()), floor), 3 * max(float(np.median(y, axis=1).max()), floor)
    line = "%s / %s all %d gradients: norm distance to fp64: worst %.3e (bound %.3e), median %.3e (bound %.3e)" % (
        what, net, int(live.sum()), o.max(), bmax, np.median(o), bmed)
    lines.append(line)
    if not (o.max() <= bmax and np.median(o) <= bmed):
        failures.append(line)
    assert covered[live].all(), "parameter groups do not cover %s" % [n for i, n in enumerate(names) if live[i] and not covered[i]]
    if precision == "f32":
        # directly against the reference's fp32 norms: never tighter than the golden is itself (its own norms sit up to 8e-3 from
        # the fp64 evaluation at these shapes - the graphs stay ill-conditioned through their train-mode BatchNorms)
        rel = np.abs(norms - gold["%s_%s_norms" % (tag, net)])[live] / (gold["%s_%s_norms" % (tag, net)][live] + 1e-300)
        bound = max(1e-3, 4 * float(yards[0][1][live].max()))
        lines.append("%s / %s: gradient norms within %.3e of the reference's fp32 (bound %.1e)" % (what, net, rel.max(), bound))
        if not rel.max() <= bound:
            failures.append(lines[-1])
    print("\n".join(lines))
    assert not failures, "\n".join(failures)


def _check_losses(what, gold, tag, got, precision, loose=("sync",)):
    for k, v in got.items():
        ref, r64 = float(gold["%s_%s" % (tag, k)]), float(gold["%s_%s64" % (tag, k)])
        if precision == "f32":
            bound = (1e-3 if k in loose else 1e-4) * abs(ref)
            assert abs(v - ref) <= bound, "%s: %s loss %.7f vs reference %.7f (bound %.1e)" % (what, k, v, ref, bound)
        else:
            spread = max(abs(float(gold["%s_%s_noise%d" % (tag, k, s)]) - r64) for s in range(int(gold[tag + "_noise_seeds"])))
            bound = 3 * max(spread, 2.0 ** -8 * abs(r64))
            assert abs(v - r64) <= bound, "%s: %s loss %.6f vs fp64 %.6f (bound %.2e)" % (what, k, v, r64, bound)


GEN_GROUPS = (["output_block"] + ["face_decoder_blocks.%d" % i for i in range(6, -1, -1)] +
              ["face_encoder_blocks.%d" % i for i in range(7)] + ["audio_encoder"])
DISC_GROUPS = ["face_encoder_blocks.%d" % i for i in range(7)] + ["binary_pred"]


@pytest.fixture(params=["f32", "bf16"])
def precision(request):
    from wav2lip_amd import engine
    engine.set_train_precision(request.param)
    yield request.param
    engine.set_train_precision("f32")


def test_syncnet_step_at_batch_512(gold, precision, cuda):
    """BASELINE configs[2]: color_syncnet_train.py:155-165 at its batch size"""
    from wav2lip_amd import losses, models
    B = int(gold["cfg3_batch"])
    assert B == 512
    S = _load(models.SyncNet_color, 2, cuda)
    b = _batch(3, B, cuda)
    S.train()
    a, v = S(b["mel"], b["x"])                                  # the body of train.syncnet_train_step, outputs kept
    loss = losses.cosine_loss(a, v, b["y"])
    loss.backward()
    got = torch.cat([a, v], 1).detach().cpu()[::7, ::3].numpy()
    ref = gold["cfg3_out_slice"] if precision == "f32" else gold["cfg3_out_slice64"]
    assert np.abs(got - ref).max() <= (2e-5 if precision == "f32" else 3e-2), np.abs(got - ref).max()
    _check_losses("SyncNet step, batch 512, " + precision, gold, "cfg3", {"loss": float(loss)}, precision)
    _check_gradients("SyncNet step, batch 512, " + precision, gold, "cfg3", "S", S, ["face_encoder", "audio_encoder"], precision)


def test_wav2lip_train_step_at_batch_64x5(gold, precision, cuda):
    """BASELINE configs[3]: wav2lip_train.py:220-231 at batch 64, T = 5 (320 frames through the generator, 64 windows through the
    frozen train-mode SyncNet)"""
    from wav2lip_amd import models, optim, train
    B = int(gold["cfg4_batch"])
    assert B == 64
    G = _load(models.Wav2Lip, 0, cuda)
    S = _load(models.SyncNet_color, 2, cuda)
    for p in S.parameters():
        p.requires_grad = False
    b = _batch(4, B, cuda)
    opt = optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-8, betas=(0.5, 0.999))
    loss, l1, sync, g = train.wav2lip_train_step(G, S, opt, b["x"], b["indiv_mels"], b["mel"], b["gt"], syncnet_wt=0.03,
                                                 return_generated=True)
    what = "wav2lip_train step, 64 x 5 frames, " + precision
    got = g.cpu()[::9, :, ::2, ::12, ::12].numpy()
    ref = gold["cfg4_out_slice"] if precision == "f32" else gold["cfg4_out_slice64"]
    oerr = np.abs(got - ref)
    print("%s: generated frames (%d sampled values) max |err| %.3e, mean |err| %.3e" % (what, oerr.size, oerr.max(), oerr.mean()))
    # bf16: 40 layers of bf16 storage; measured max 6.2e-2 / mean 8.4e-3 on sigmoid outputs whose L1 loss agrees to 1e-6 (unbiased)
    assert oerr.max() <= (2e-5 if precision == "f32" else 0.15) and oerr.mean() <= (2e-6 if precision == "f32" else 2e-2), (oerr.max(), oerr.mean())
    _check_losses(what, gold, "cfg4", {"loss": float(loss), "l1": float(l1), "sync": float(sync)}, precision)
    _check_gradients(what, gold, "cfg4", "G", G, GEN_GROUPS, precision)
    assert all(p.grad is None for p in S.parameters())


def test_hq_train_step_at_batch_64x5(gold, precision, cuda):
    """BASELINE configs[4]: hq_wav2lip_train.py:221-256 at batch 64, T = 5: generator step with the perceptual term through
    Wav2Lip_disc_qual, then D(real) / D(fake)"""
    from wav2lip_amd import models, optim, train
    B = int(gold["cfg5_batch"])
    G = _load(models.Wav2Lip, 0, cuda)
    S = _load(models.SyncNet_color, 2, cuda)
    D = _load(models.Wav2Lip_disc_qual, 4, cuda)
    for p in S.parameters():
        p.requires_grad = False
    b = _batch(5, B, cuda)
    optG = optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-8, betas=(0.5, 0.999))
    optD = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=1e-8, betas=(0.5, 0.999))
    out = train.hq_train_step(G, D, S, optG, optD, b["x"], b["indiv_mels"], b["mel"], b["gt"], syncnet_wt=0.03, disc_wt=0.07)
    what = "hq_wav2lip_train step, 64 x 5 frames, " + precision
    _check_losses(what, gold, "cfg5", {k: float(v) for k, v in out.items()}, precision)
    _check_gradients(what, gold, "cfg5", "G", G, GEN_GROUPS, precision)
    _check_gradients(what, gold, "cfg5", "D", D, DISC_GROUPS, precision)
