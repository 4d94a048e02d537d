"""Generates the committed golden fixture of the three training steps AT THE BASELINE LAUNCH SHAPES from the REAL reference
(its nn.Modules and loss expressions under torch autograd on CPU, fp32).  Runs only in the build container (needs
/root/reference, ~45 GB of host memory for the fp64 passes, tens of minutes on 8 cores); the fixture travels, the reference
does not.

    python tests/golden/make_golden_train_baseline.py [--cfg 3 4 5] [--noise-seeds 2]

  cfg3  color_syncnet_train.py:155-165   SyncNet_color.train(), cosine_loss, backward                         B = 512
  cfg4  wav2lip_train.py:220-231         Wav2Lip.train(), frozen train-mode SyncNet, 0.03 sync + 0.97 L1      B = 64, T = 5
  cfg5  hq_wav2lip_train.py:221-256      cfg4 + 0.07 perceptual through Wav2Lip_disc_qual; D(real) / D(fake)  B = 64, T = 5

Weights: wav2lip_amd/synthetic.py (seeds 2 / 0 / 4 as in the small-batch goldens); inputs: synthetic.train_batch(cfg, B, seed 5).
Per case the file keeps
  * the losses of the reference step (fp32) and of the fp64 evaluation of the oracle graph (oracle/models_ref.py *_graph, which
    this script first checks against the reference: loss and every gradient),
  * for EVERY parameter gradient its L2 norm and four "sketches" <g, r_k> with fixed +-1 vectors (synthetic.sketch_vectors) -
    a 36 M-parameter gradient does not fit a fixture, its norm alone does not see a wrong direction - from the fp32 reference
    and from the fp64 graph,
  * a subsampled slice of the network outputs,
  * the same norms / sketches / losses from `--noise-seeds` fp64 evaluations under the bf16-storage error model
    (oracle/error_models.py): the measured yardstick the bf16 path is held to.
The GPU tests (tests/test_train_baseline_gpu.py) and tools/train_bench.py's in-run parity read it; no CPU graph runs there.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import models_ref  # noqa: E402
from oracle.error_models import bf16_storage_noise  # noqa: E402
from wav2lip_amd import synthetic as synth  # noqa: E402
from make_golden import ref_models  # noqa: E402
from make_golden_train import check_oracle, load, oracle_sd, to64  # noqa: E402

OUT = os.path.join(HERE, "golden_train_baseline_v1.npz")
SEED_IN = 5
SYNC_WT, DISC_WT = 0.03, 0.07


def tens(d, dt=torch.float32):
    return {k: torch.from_numpy(v).to(dt) for k, v in d.items()}


def summarize(named_grads):
    """{name: grad} -> (names, norms [P], sketches [P, 4]) in float64"""
    names = sorted(named_grads)
    norms = np.zeros(len(names))
    sk = np.zeros((len(names), 4))
    for i, n in enumerate(names):
        g = named_grads[n].detach().double().reshape(-1)
        norms[i] = float(g.norm())
        r = torch.from_numpy(synth.sketch_vectors(n, g.numel()).astype(np.float64))
        sk[i] = (r @ g).numpy()
    return names, norms, sk


def grads_of(osd):
    return {k: t.grad for k, t in osd.items() if t.requires_grad and t.grad is not None}


# ------------------------------------------------------------------ the three graphs on the oracle (any dtype)
def sync_graph(sds, b, dt):
    osd = to64(sds) if dt == torch.float64 else oracle_sd(sds)
    a, v = models_ref.syncnet_graph(osd, b["mel"].to(dt), b["x"].to(dt), training=True)
    loss = models_ref.cosine_loss(a, v, b["y"].to(dt))
    loss.backward()
    return {"loss": loss.item()}, {"S": grads_of(osd)}, torch.cat([a, v], 1).detach()


def gen_graph(sdg, sds, sdd, b, dt, hq):
    f64 = dt == torch.float64
    osd = to64(sdg) if f64 else oracle_sd(sdg)
    oss = to64(sds, requires_grad=False) if f64 else oracle_sd(sds, requires_grad=False)
    g = models_ref.wav2lip_graph(osd, b["indiv_mels"].to(dt), b["x"].to(dt), training=True)
    gl = g[:, :, :, g.size(3) // 2:]
    gl = torch.cat([gl[:, :, i] for i in range(5)], dim=1)
    a, v = models_ref.syncnet_graph(oss, b["mel"].to(dt), gl, training=True)
    sync = models_ref.cosine_loss(a, v, torch.ones(g.size(0), 1, dtype=dt))
    l1 = F.l1_loss(g, b["gt"].to(dt))
    losses = {"sync": sync.item(), "l1": l1.item()}
    if not hq:
        loss = SYNC_WT * sync + (1 - SYNC_WT) * l1
        loss.backward()
        losses["loss"] = loss.item()
        return losses, {"G": grads_of(osd)}, g.detach()
    osdd = to64(sdd) if f64 else oracle_sd(sdd)
    p = models_ref.disc_graph(osdd, g)
    perc = F.binary_cross_entropy(p, torch.ones(len(p), 1, dtype=dt))
    loss = SYNC_WT * sync + DISC_WT * perc + (1. - SYNC_WT - DISC_WT) * l1
    loss.backward()
    losses.update(perceptual=perc.item(), loss=loss.item())
    gg = grads_of(osd)
    for t in osdd.values():                 # disc_optimizer.zero_grad() (hq_wav2lip_train.py:245)
        t.grad = None
    pr = models_ref.disc_graph(osdd, b["gt"].to(dt))
    real = F.binary_cross_entropy(pr, torch.ones(len(pr), 1, dtype=dt))
    real.backward()
    pf = models_ref.disc_graph(osdd, g.detach())
    fake = F.binary_cross_entropy(pf, torch.zeros(len(pf), 1, dtype=dt))
    fake.backward()
    losses.update(disc_real=real.item(), disc_fake=fake.item())
    return losses, {"G": gg, "D": grads_of(osdd)}, g.detach()


# ------------------------------------------------------------------ the reference steps (fp32, the real modules)
def ref_sync_step(rm, b):
    S = rm.SyncNet_color().train()
    sds = load(S, seed=2)
    a, v = S(b["mel"], b["x"])
    loss = F.binary_cross_entropy(F.cosine_similarity(a, v).unsqueeze(1), b["y"])      # color_syncnet_train.py:133-139
    loss.backward()
    return sds, {"loss": loss.item()}, {"S": {n: p.grad.detach() for n, p in S.named_parameters()}}, torch.cat([a, v], 1).detach()


def ref_gen_step(rm, b, hq):
    G = rm.Wav2Lip().train()
    sdg = load(G, seed=0)
    S = rm.SyncNet_color()                       # never put in eval mode by the reference; parameters frozen
    sds = load(S, seed=2)
    for p in S.parameters():
        p.requires_grad = False
    D, sdd = None, None
    if hq:
        D = rm.Wav2Lip_disc_qual().train()
        sdd = load(D, seed=4)
    B = b["x"].shape[0]
    g = G(b["indiv_mels"], b["x"])
    gl = g[:, :, :, g.size(3) // 2:]
    gl = torch.cat([gl[:, :, i] for i in range(5)], dim=1)
    a, v = S(b["mel"], gl)
    sync = F.binary_cross_entropy(F.cosine_similarity(a, v).unsqueeze(1), torch.ones(B, 1))
    l1 = F.l1_loss(g, b["gt"])
    losses = {"sync": sync.item(), "l1": l1.item()}
    if not hq:
        loss = SYNC_WT * sync + (1 - SYNC_WT) * l1
        loss.backward()
        losses["loss"] = loss.item()
        return sdg, sds, sdd, losses, {"G": {n: p.grad.detach() for n, p in G.named_parameters()}}, g.detach()
    p = D(g)
    perc = F.binary_cross_entropy(p, torch.ones(len(p), 1))       # perceptual_forward without its hard-coded .cuda()
    loss = SYNC_WT * sync + DISC_WT * perc + (1. - SYNC_WT - DISC_WT) * l1
    loss.backward()
    losses.update(perceptual=perc.item(), loss=loss.item())
    gg = {n: p_.grad.detach().clone() for n, p_ in G.named_parameters()}
    D.zero_grad()
    pr = D(b["gt"])
    real = F.binary_cross_entropy(pr, torch.ones(len(pr), 1))
    real.backward()
    pf = D(g.detach())
    fake = F.binary_cross_entropy(pf, torch.zeros(len(pf), 1))
    fake.backward()
    losses.update(disc_real=real.item(), disc_fake=fake.item())
    return sdg, sds, sdd, losses, {"G": gg, "D": {n: p_.grad.detach() for n, p_ in D.named_parameters()}}, g.detach()


def record(out, tag, suffix, losses, grads):
    for k, v in losses.items():
        out["%s_%s%s" % (tag, k, suffix)] = np.float64(v)
    for net, gr in grads.items():
        names, norms, sk = summarize(gr)
        out["%s_%s_names" % (tag, net)] = np.array(names)
        out["%s_%s_norms%s" % (tag, net, suffix)] = norms
        out["%s_%s_sketch%s" % (tag, net, suffix)] = sk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, nargs="+", default=[3, 4, 5])
    ap.add_argument("--noise-seeds", type=int, default=2)
    ap.add_argument("--batch3", type=int, default=512)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    rm = ref_models()
    out = dict(np.load(args.out)) if os.path.exists(args.out) else {}
    out["torch_version"] = np.array(torch.__version__)
    out["input_seed"] = np.int64(SEED_IN)
    for cfg in args.cfg:
        tag = "cfg%d" % cfg
        B = args.batch3 if cfg == 3 else args.batch
        b = tens(synth.train_batch(cfg, B, SEED_IN))
        t0 = time.time()
        if cfg == 3:
            sds, losses, grads, outs = ref_sync_step(rm, b)
            run = lambda dt: sync_graph(sds, b, dt)                                   # noqa: E731
        else:
            sdg, sds, sdd, losses, grads, outs = ref_gen_step(rm, b, cfg == 5)
            run = lambda dt: gen_graph(sdg, sds, sdd, b, dt, cfg == 5)                # noqa: E731
        print("%s: reference step %.0f s, losses %s" % (tag, time.time() - t0, losses), flush=True)
        # the oracle graph reproduces the reference (fp32): loss and every gradient
        t0 = time.time()
        lo, go, oo = run(torch.float32)
        for k, v in losses.items():
            assert abs(lo[k] - v) <= 2e-6 * abs(v) + 1e-9, (tag, k, lo[k], v)
        for net in grads:
            check_oracle("%s/%s" % (tag, net), grads[net], go[net])
        assert float((oo - outs).abs().max()) <= 1e-6
        del go, oo
        print("%s: oracle graph fp32 %.0f s" % (tag, time.time() - t0), flush=True)
        out[tag + "_batch"] = np.int64(B)
        record(out, tag, "", losses, grads)
        out[tag + "_out_slice"] = (outs[::7, ::3] if cfg == 3 else outs[::9, :, ::2, ::12, ::12]).numpy().copy()
        del grads, outs
        t0 = time.time()
        l64, g64, o64 = run(torch.float64)
        print("%s: oracle graph fp64 %.0f s, losses %s" % (tag, time.time() - t0, l64), flush=True)
        record(out, tag, "64", l64, g64)
        out[tag + "_out_slice64"] = (o64[::7, ::3] if cfg == 3 else o64[::9, :, ::2, ::12, ::12]).numpy().copy()
        del g64, o64
        for s in range(args.noise_seeds):
            t0 = time.time()
            with bf16_storage_noise(100 * cfg + s):
                ln, gn, _ = run(torch.float64)
            print("%s: bf16 error model seed %d %.0f s, losses %s" % (tag, s, time.time() - t0, ln), flush=True)
            record(out, tag, "_noise%d" % s, ln, gn)
            del gn
        out[tag + "_noise_seeds"] = np.int64(args.noise_seeds)
        np.savez_compressed(args.out, **out)
        print("wrote", args.out, "(%.0f kB)" % (os.path.getsize(args.out) / 1e3), flush=True)


if __name__ == "__main__":
    main()
