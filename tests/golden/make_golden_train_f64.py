#!/usr/bin/env python
"""fp64 companions of the gradient TENSORS tests/golden/golden_train_v1.npz keeps in fp32 (make_golden_train.py stores fp64
only for the gradient norms): written to tests/golden/golden_train_f64_v1.npz as `<tag>_grad64/<parameter>` (tags sync, gen, disc:
the same three steps, same seeds, evaluated in float64 through oracle/models_ref.py's graphs, which make_golden_train.py pins to the
real reference's autograd within 1e-5) and `disc_perceptual_dfake64` (hq_wav2lip_train.py:233: the perceptual-loss gradient
w.r.t. the fake frames, evaluated by the REAL reference module in float64).

Why: these steps run train-mode BatchNorm over 4-20 samples, where the reference's own fp32 gradients are inexact - the
perceptual gradient (magnitude 4e-5) is 2.9e-2 (L-inf, relative) away from its fp64 value - so a tight bound against an fp32
golden pins a summation order, not the math.  tests/test_train_gpu.py measures the HIP path's distance to these fp64 tensors with
the reference's own fp32 distance as the yardstick.  golden_train_v1.npz is NOT regenerated; the script prints, per tensor, the
committed fp32 golden's distance to fp64.

    python tests/golden/make_golden_train_f64.py        (needs /root/reference; build container only)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden_train as mg  # noqa: E402  (puts the repo root on sys.path)
from oracle import models_ref  # noqa: E402
from wav2lip_amd import synthetic as synth  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def main():
    torch.set_num_threads(8)
    rm = mg.ref_models()
    gold = np.load(os.path.join(HERE, "golden_train_v1.npz"))
    out = {}

    def keep(tag, o64):
        for key in gold.files:
            if key.startswith(tag + "_grad/"):
                n = key[len(tag) + 6:]
                out["%s_grad64/%s" % (tag, n)] = o64[n].grad.numpy().copy()
                print("%-5s %-52s fp32 golden vs fp64: %.3e" % (tag, n, rel(gold[key].astype(np.float64), out["%s_grad64/%s" % (tag, n)])))
        names = [str(n) for n in gold[tag + "_grad_names"]]
        assert np.allclose(mg.norms64(o64, names), gold[tag + "_grad_norms64"], rtol=1e-9, atol=0), tag + ": not the committed fp64 graph"

    # SyncNet train step (make_golden_train.py, first section)
    S = rm.SyncNet_color()
    sds = mg.load(S, seed=2)
    x = torch.from_numpy(synth.sync_faces(4, seed=11))
    mel = torch.from_numpy(synth.mel_windows(4, seed=11)).unsqueeze(1)
    y = torch.tensor([[1.], [0.], [1.], [0.]])
    o64 = mg.to64(sds)
    a64, v64 = models_ref.syncnet_graph(o64, mel.double(), x.double(), training=True)
    F.binary_cross_entropy(F.cosine_similarity(a64, v64).unsqueeze(1), y.double()).backward()
    keep("sync", o64)

    # generator train step (B=4, T=5) with the frozen train-mode SyncNet
    G = rm.Wav2Lip()
    sdg = mg.load(G, seed=0)
    B, T, wt = 4, 5, 0.03
    r = mg.rng(21)
    gt = torch.from_numpy(r.uniform(0, 1, (B, 3, T, 96, 96)).astype(np.float32))
    wrong = torch.from_numpy(r.uniform(0, 1, (B, 3, T, 96, 96)).astype(np.float32))
    masked = gt.clone()
    masked[:, :, :, 48:] = 0.
    xin = torch.cat([masked, wrong], dim=1)
    indiv = torch.from_numpy(r.uniform(-4, 4, (B, T, 1, 80, 16)).astype(np.float32))
    melw = torch.from_numpy(r.uniform(-4, 4, (B, 1, 80, 16)).astype(np.float32))
    o64, s64 = mg.to64(sdg), mg.to64(sds, requires_grad=False)
    g64 = models_ref.wav2lip_graph(o64, indiv.double(), xin.double(), training=True)
    gl64 = g64[:, :, :, 48:]
    gl64 = torch.cat([gl64[:, :, i] for i in range(T)], dim=1)
    a64, v64 = models_ref.syncnet_graph(s64, melw.double(), gl64, training=True)
    (wt * F.binary_cross_entropy(F.cosine_similarity(a64, v64).unsqueeze(1), torch.ones(B, 1, dtype=torch.float64)) +
     (1 - wt) * F.l1_loss(g64, gt.double())).backward()
    keep("gen", o64)

    # discriminator (B=1, T=5): D(real) / D(fake) parameter gradients, and the perceptual gradient w.r.t. the fake frames
    D = rm.Wav2Lip_disc_qual().train()
    sdd = mg.load(D, seed=4)
    fake = torch.from_numpy(synth.disc_frames(1, 5, seed=31))
    real = torch.from_numpy(synth.disc_frames(1, 5, seed=32))
    o64 = mg.to64(sdd)
    (F.binary_cross_entropy(models_ref.disc_graph(o64, real.double()), torch.ones(5, 1, dtype=torch.float64)) +
     F.binary_cross_entropy(models_ref.disc_graph(o64, fake.double()), torch.zeros(5, 1, dtype=torch.float64))).backward()
    keep("disc", o64)
    D = D.double()
    f64 = fake.double().requires_grad_(True)
    F.binary_cross_entropy(D(f64), torch.ones(5, 1, dtype=torch.float64)).backward()     # perceptual_forward without its .cuda()
    out["disc_perceptual_dfake64"] = f64.grad[:, :, :, 48::4, ::4].numpy().copy()
    print("disc  %-52s fp32 golden vs fp64: %.3e" % ("perceptual d/d(fake frames)",
                                                      rel(gold["disc_perceptual_dfake"].astype(np.float64), out["disc_perceptual_dfake64"])))
    path = os.path.join(HERE, "golden_train_f64_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
