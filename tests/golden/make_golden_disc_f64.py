#!/usr/bin/env python
"""fp64 evaluation of ONE tensor of tests/golden/golden_train_v1.npz: the perceptual-loss gradient w.r.t. the fake frames
(hq_wav2lip_train.py:233 through Wav2Lip_disc_qual in train mode, B=1, T=5 - five-sample BatchNorm statistics), written to
tests/golden/disc_perceptual_dfake64_v1.npy.

Why: that gradient has magnitude 4e-5 and the REAL reference's own fp32 result sits 2.9e-2 (L-inf, relative) away from its fp64
result - measured by this script, printed below - so a bound of 1e-2 against the fp32 golden held a summation order, not the
math.  tests/test_train_gpu.py::test_disc_steps_match_reference_golden holds the HIP path to the fp64 tensor with the
reference's own fp32 distance as the yardstick.  golden_train_v1.npz itself is not regenerated (its fp32 values depend on the
host's thread count in the last bits; everything else stays anchored to the committed file).

    python tests/golden/make_golden_disc_f64.py        (needs /root/reference; build container only)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden_train as mg  # noqa: E402  (puts the repo root on sys.path)
from wav2lip_amd import synthetic as synth  # noqa: E402


def grad(dtype):
    rm = mg.ref_models()
    D = rm.Wav2Lip_disc_qual().train()
    mg.load(D, seed=4)
    D = D.to(dtype)
    fake = torch.from_numpy(synth.disc_frames(1, 5, seed=31)).to(dtype).requires_grad_(True)
    perc = F.binary_cross_entropy(D(fake), torch.ones(5, 1, dtype=dtype))     # perceptual_forward without its hard-coded .cuda()
    perc.backward()
    return fake.grad[:, :, :, 48::4, ::4].double().numpy().copy()


def main():
    g64, g32 = grad(torch.float64), grad(torch.float32)
    committed = np.load(os.path.join(HERE, "golden_train_v1.npz"))["disc_perceptual_dfake"].astype(np.float64)
    scale = np.abs(g64).max()
    print("scale %.3e; reference fp32 (this host) vs fp64: %.3e; committed fp32 golden vs fp64: %.3e"
          % (scale, np.abs(g32 - g64).max() / scale, np.abs(committed - g64).max() / scale))
    np.save(os.path.join(HERE, "disc_perceptual_dfake64_v1.npy"), g64)


if __name__ == "__main__":
    main()
