"""Generates the committed TRAINING golden fixtures from the REAL reference (its nn.Modules under torch autograd on
CPU fp32).  Runs only in the build container (needs /root/reference); the fixtures travel, the reference does not.

    python tests/golden/make_golden_train.py

Three cases, each following the reference loop it names, with seeded synthetic weights (wav2lip_amd/synthetic.py: pure data generation) and inputs:
  sync_*   color_syncnet_train.py:150-164   SyncNet_color.train(); cosine_loss; backward              (B=4)
  gen_*    wav2lip_train.py:211-229         Wav2Lip.train(); frozen train-mode SyncNet; 0.03*sync + 0.97*L1; backward
                                            (B=4, T=5)
  disc_*   hq_wav2lip_train.py:233-253      perceptual loss wrt the fake frames; D(real)/D(fake) BCE; backward  (B=1, T=5)
For every case the oracle's differentiable restatement (oracle/models_ref.py *_graph) is checked against the reference
(loss and every gradient), then the losses, the L2 norm of every parameter gradient, a few whole gradient tensors and the
updated BatchNorm running statistics are frozen into golden_train_v1.npz.  Train-mode BatchNorm over a handful of samples
makes these gradients ill-conditioned (the reference's own fp32 result is several per cent away from exact arithmetic), so
the same graphs are also evaluated in fp64 and those gradient norms are stored (`*_grad_norms64`): the GPU tests bound the
HIP path's distance to the fp64 result by a small multiple of the reference's own distance.
Also freezes three torch.optim.Adam steps on seeded tensors (the optimiser the reference constructs).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import models_ref  # noqa: E402
from wav2lip_amd import synthetic as synth  # noqa: E402
from make_golden import ref_models  # noqa: E402


def rng(seed):
    return np.random.default_rng(seed)


def load(module, seed):
    sd = synth.synthetic_state_dict({k: tuple(v.shape) for k, v in module.state_dict().items()}, seed=seed)
    module.load_state_dict(sd)
    return sd


def oracle_sd(sd, requires_grad=True):
    out = {}
    for k, v in sd.items():
        t = v.clone()
        if t.is_floating_point() and "running_" not in k and requires_grad:
            t.requires_grad_(True)
        out[k] = t
    return out


def to64(sd, requires_grad=True):
    out = {}
    for k, v in sd.items():
        t = v.clone()
        if t.is_floating_point():
            t = t.double()
            if "running_" not in k and requires_grad:
                t.requires_grad_(True)
        out[k] = t
    return out


def norms64(osd64, names):
    return np.array([float(osd64[n].grad.norm()) for n in names], dtype=np.float64)


def grad_report(named_grads, keep):
    """L2 norms of all gradients (sorted by name) + the full tensors named in `keep`"""
    names = sorted(named_grads)
    norms = np.array([float(named_grads[n].double().norm()) for n in names], dtype=np.float64)
    return names, norms, {n: named_grads[n].numpy().copy() for n in keep}


def check_oracle(tag, ref_named, ora_named):
    worst = 0.0
    for n, g in ref_named.items():
        d = (g - ora_named[n]).abs().max().item()
        s = g.abs().max().item()
        worst = max(worst, d / (s + 1e-30))
    print("%s: oracle vs reference, worst relative gradient difference %.3e" % (tag, worst))
    assert worst <= 1e-5, "oracle/models_ref.py *_graph does not reproduce the reference gradients"


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    rm = ref_models()
    out = {}

    # ------------------------------------------------------------ SyncNet train step
    S = rm.SyncNet_color().train()
    sds = load(S, seed=2)
    x = torch.from_numpy(synth.sync_faces(4, seed=11))
    mel = torch.from_numpy(synth.mel_windows(4, seed=11)).unsqueeze(1)
    y = torch.tensor([[1.], [0.], [1.], [0.]])
    a, v = S(mel, x)
    loss = F.binary_cross_entropy(F.cosine_similarity(a, v).unsqueeze(1), y)
    loss.backward()
    ref_g = {n: p.grad.detach() for n, p in S.named_parameters()}
    osd = oracle_sd(sds)
    ao, vo = models_ref.syncnet_graph(osd, mel, x, training=True)
    lo = models_ref.cosine_loss(ao, vo, y)
    lo.backward()
    assert abs(lo.item() - loss.item()) <= 1e-6 * abs(loss.item())
    check_oracle("syncnet", ref_g, {n: osd[n].grad for n in ref_g})
    keep = ["face_encoder.0.conv_block.0.weight", "face_encoder.0.conv_block.1.weight", "face_encoder.16.conv_block.1.bias",
            "audio_encoder.0.conv_block.0.weight", "audio_encoder.13.conv_block.0.bias", "face_encoder.5.conv_block.0.bias"]
    names, norms, full = grad_report(ref_g, keep)
    out["sync_loss"] = np.float32(loss.item())
    out["sync_grad_names"] = np.array(names)
    out["sync_grad_norms"] = norms
    for n, t in full.items():
        out["sync_grad/" + n] = t
    o64 = to64(sds)
    a64, v64 = models_ref.syncnet_graph(o64, mel.double(), x.double(), training=True)
    F.binary_cross_entropy(F.cosine_similarity(a64, v64).unsqueeze(1), y.double()).backward()
    out["sync_grad_norms64"] = norms64(o64, names)
    out["sync_a"] = a.detach().numpy()
    out["sync_v"] = v.detach().numpy()
    out["sync_running_mean/face_encoder.0"] = S.state_dict()["face_encoder.0.conv_block.1.running_mean"].numpy().copy()
    out["sync_running_var/face_encoder.0"] = S.state_dict()["face_encoder.0.conv_block.1.running_var"].numpy().copy()
    print("syncnet loss", loss.item())

    # ------------------------------------------------------------ generator train step (B=4, T=5)
    G = rm.Wav2Lip().train()
    sdg = load(G, seed=0)
    S2 = rm.SyncNet_color()          # stays in train mode, parameters frozen (wav2lip_train.py:187-190)
    sds2 = load(S2, seed=2)
    for p in S2.parameters():
        p.requires_grad = False
    B, T = 4, 5
    r = rng(21)
    gt = torch.from_numpy(r.uniform(0, 1, (B, 3, T, 96, 96)).astype(np.float32))
    wrong = torch.from_numpy(r.uniform(0, 1, (B, 3, T, 96, 96)).astype(np.float32))
    masked = gt.clone()
    masked[:, :, :, 48:] = 0.
    xin = torch.cat([masked, wrong], dim=1)
    indiv = torch.from_numpy(r.uniform(-4, 4, (B, T, 1, 80, 16)).astype(np.float32))
    melw = torch.from_numpy(r.uniform(-4, 4, (B, 1, 80, 16)).astype(np.float32))
    wt = 0.03
    g = G(indiv, xin)
    gl = g[:, :, :, g.size(3) // 2:]
    gl = torch.cat([gl[:, :, i] for i in range(T)], dim=1)
    a, v = S2(melw, gl)
    sync = F.binary_cross_entropy(F.cosine_similarity(a, v).unsqueeze(1), torch.ones(B, 1))
    l1 = F.l1_loss(g, gt)
    loss = wt * sync + (1 - wt) * l1
    loss.backward()
    ref_g = {n: p.grad.detach() for n, p in G.named_parameters()}
    osd = oracle_sd(sdg)
    oss = oracle_sd(sds2, requires_grad=False)
    go = models_ref.wav2lip_graph(osd, indiv, xin, training=True)
    so = models_ref.get_sync_loss(oss, melw, go, training=True)
    lo = wt * so + (1 - wt) * F.l1_loss(go, gt)
    lo.backward()
    assert abs(lo.item() - loss.item()) <= 1e-6 * abs(loss.item()), (lo.item(), loss.item())
    check_oracle("generator", ref_g, {n: osd[n].grad for n in ref_g})
    keep = ["face_encoder_blocks.0.0.conv_block.0.weight", "face_encoder_blocks.0.0.conv_block.1.weight",
            "audio_encoder.0.conv_block.0.weight", "output_block.1.weight", "output_block.1.bias",
            "face_decoder_blocks.6.0.conv_block.1.bias", "face_decoder_blocks.0.0.conv_block.0.bias",
            "face_encoder_blocks.1.1.conv_block.0.weight"]
    names, norms, full = grad_report(ref_g, keep)
    o64, s64 = to64(sdg), to64(sds2, requires_grad=False)
    g64 = models_ref.wav2lip_graph(o64, indiv.double(), xin.double(), training=True)
    gl64 = g64[:, :, :, 48:]
    gl64 = torch.cat([gl64[:, :, i] for i in range(T)], dim=1)
    a64, v64 = models_ref.syncnet_graph(s64, melw.double(), gl64, training=True)
    l64 = wt * F.binary_cross_entropy(F.cosine_similarity(a64, v64).unsqueeze(1), torch.ones(B, 1, dtype=torch.float64)) + \
        (1 - wt) * F.l1_loss(g64, gt.double())
    l64.backward()
    out["gen_grad_norms64"] = norms64(o64, names)
    out["gen_loss64"] = np.float64(l64.item())
    out["gen_inputs_seed"] = np.int64(21)
    out["gen_loss"] = np.float32(loss.item())
    out["gen_l1"] = np.float32(l1.item())
    out["gen_sync"] = np.float32(sync.item())
    out["gen_grad_names"] = np.array(names)
    out["gen_grad_norms"] = norms
    for n, t in full.items():
        out["gen_grad/" + n] = t
    out["gen_out_mean"] = np.float64(g.detach().double().mean().item())
    out["gen_out_t0"] = g.detach()[:, :, 0, ::8, ::8].numpy().copy()
    out["gen_running_var/output_block.0"] = G.state_dict()["output_block.0.conv_block.1.running_var"].numpy().copy()
    print("generator loss", loss.item(), "l1", l1.item(), "sync", sync.item())

    # ------------------------------------------------------------ discriminator (B=1, T=5)
    D = rm.Wav2Lip_disc_qual().train()
    sdd = load(D, seed=4)
    fake = torch.from_numpy(synth.disc_frames(1, 5, seed=31)).requires_grad_(True)
    real = torch.from_numpy(synth.disc_frames(1, 5, seed=32))
    perc = F.binary_cross_entropy(D(fake), torch.ones(5, 1))     # perceptual_forward without its hard-coded .cuda()
    perc.backward()
    out["disc_perceptual"] = np.float32(perc.item())
    out["disc_perceptual_dfake"] = fake.grad[:, :, :, 48::4, ::4].numpy().copy()
    out["disc_perceptual_dfake_norm"] = np.float64(fake.grad.double().norm().item())
    assert float(fake.grad[:, :, :, :48].abs().max()) == 0.0
    osd = oracle_sd(sdd)
    fo = fake.detach().clone().requires_grad_(True)
    po = models_ref.perceptual_loss(osd, fo)
    po.backward()
    assert (fo.grad - fake.grad).abs().max().item() <= 1e-5 * fake.grad.abs().max().item()
    D.zero_grad()
    lr = F.binary_cross_entropy(D(real), torch.ones(5, 1))
    lr.backward()
    lf = F.binary_cross_entropy(D(fake.detach()), torch.zeros(5, 1))
    lf.backward()
    ref_g = {n: p.grad.detach() for n, p in D.named_parameters()}
    osd = oracle_sd(sdd)
    l1o = F.binary_cross_entropy(models_ref.disc_graph(osd, real), torch.ones(5, 1))
    l2o = F.binary_cross_entropy(models_ref.disc_graph(osd, fake.detach()), torch.zeros(5, 1))
    (l1o + l2o).backward()
    check_oracle("disc", ref_g, {n: osd[n].grad for n in ref_g})
    keep = ["face_encoder_blocks.0.0.conv_block.0.weight", "binary_pred.0.weight", "binary_pred.0.bias",
            "face_encoder_blocks.6.1.conv_block.0.bias"]
    names, norms, full = grad_report(ref_g, keep)
    o64 = to64(sdd)
    (F.binary_cross_entropy(models_ref.disc_graph(o64, real.double()), torch.ones(5, 1, dtype=torch.float64)) +
     F.binary_cross_entropy(models_ref.disc_graph(o64, fake.detach().double()), torch.zeros(5, 1, dtype=torch.float64))).backward()
    out["disc_grad_norms64"] = norms64(o64, names)
    out["disc_real_loss"] = np.float32(lr.item())
    out["disc_fake_loss"] = np.float32(lf.item())
    out["disc_grad_names"] = np.array(names)
    out["disc_grad_norms"] = norms
    for n, t in full.items():
        out["disc_grad/" + n] = t
    print("disc perceptual", perc.item(), "real", lr.item(), "fake", lf.item())

    # ------------------------------------------------------------ torch.optim.Adam, three steps
    r = rng(41)
    p0 = [torch.from_numpy(r.normal(0, 1, s).astype(np.float32)) for s in ((17,), (4, 3, 3, 3), (20000,))]
    params = [torch.nn.Parameter(p.clone()) for p in p0]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.5, 0.999))
    grads = []
    for step in range(3):
        gs = [torch.from_numpy(r.normal(0, 0.1, tuple(p.shape)).astype(np.float32)) for p in params]
        grads.append(gs)
        for p, gg in zip(params, gs):
            p.grad = gg.clone()
        opt.step()
    for i, p in enumerate(p0):
        out["adam_p0/%d" % i] = p.numpy()
        out["adam_p3/%d" % i] = params[i].detach().numpy().copy()
        for s in range(3):
            out["adam_g%d/%d" % (s, i)] = grads[s][i].numpy()

    path = os.path.join(HERE, "golden_train_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
