"""Two ranks of the product's multi-GPU path on ONE MI355X (the GPU box has one): both processes bind cuda:0 (LOCAL_RANK=0) and
exchange through gloo, whose collectives are staged through the host by wav2lip_amd/sharding.py - the HIP kernels, the
GradReducer inside backward, the weight broadcast, the shard / gather of frames are the code an 8-GPU RCCL job runs; only the wire
differs.  Compared with the single-process result of the same work (BASELINE configs[3]/[4] "DDP over 8 x MI355X")."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wav2lip_amd import synthetic as synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(target, world, *args, timeout=420):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


def _env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0")


def _load(cls, seed, dev):
    m = cls()
    m.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed))
    return m.to(dev)


def _disc_steps(D, opt, frames, steps):
    """hq_wav2lip_train.py:243-250, the real half of the discriminator step, `steps` times; returns the first step's gradients"""
    from wav2lip_amd import losses
    first = None
    for _ in range(steps):
        opt.zero_grad()
        pred = D(frames)
        losses.bce_mean(pred, torch.ones((len(pred), 1), device=pred.device)).backward()
        if first is None:
            first = [p.grad.detach().clone() for p in D.parameters()]
        opt.step()
    return first


def _train_worker(rank, world, port, q, tmp):
    _env(rank, world, port)
    from wav2lip_amd import models, optim, sharding
    ranks = sharding.init_from_env("gloo")
    assert ranks.device == torch.device("cuda", 0) and ranks.dist.get_backend() == "gloo"
    D = _load(models.Wav2Lip_disc_qual, 40 + rank, ranks.device).train()      # every rank starts from its OWN weights ...
    n = sharding.broadcast_state(ranks.dist, D)                                # ... and takes rank 0's
    sharding.GradReducer(ranks.dist, bucket_bytes=8 << 20).attach(D)          # several buckets: the discriminator holds ~36 MB
    opt = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
    frames = torch.from_numpy(synth.disc_frames(4, 2, seed=8)[2 * rank:2 * rank + 2]).to(ranks.device)
    grads = _disc_steps(D, opt, frames, 2)
    torch.save(dict(grads=[g.cpu() for g in grads], params=[p.detach().cpu() for p in D.parameters()], n_bcast=n),
               os.path.join(tmp, "rank%d.pt" % rank))
    q.put((rank, True))
    ranks.close()


def test_two_ranks_on_one_gpu_train_the_discriminator_like_one_process_on_the_global_batch(cuda, tmp_path):
    """The discriminator has no BatchNorm, so data-parallel training on two half batches IS the single-process step on the whole
    batch: the averaged gradient of the first step within fp32 summation-order noise, and the SAME weights on both ranks after two
    Adam steps, bit for bit (both ranks apply the same averaged gradients to the same broadcast weights)."""
    from wav2lip_amd import models, optim
    tmp = str(tmp_path)
    _launch(_train_worker, 2, tmp)
    r0, r1 = (torch.load(os.path.join(tmp, "rank%d.pt" % r)) for r in range(2))
    assert r0["n_bcast"] >= 1
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)
    D = _load(models.Wav2Lip_disc_qual, 40, cuda).train()
    opt = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
    ref = _disc_steps(D, opt, torch.from_numpy(synth.disc_frames(4, 2, seed=8)).to(cuda), 1)
    for g, r in zip(r0["grads"], ref):
        r = r.cpu()
        assert (g - r).abs().max().item() <= 2e-5 * r.abs().max().item() + 1e-9


def _infer_worker(rank, world, port, q, tmp):
    _env(rank, world, port)
    from wav2lip_amd import inference, models, sharding
    ranks = sharding.init_from_env("gloo")
    G = _load(models.Wav2Lip, 0, ranks.device).eval()
    frames = list(synth.face_crops_u8(9, seed=4))
    out = inference.lipsync(G, frames, synth.sine_wav(1.5), fps=25., batch_size=8, ranks=ranks)
    if out is not None:
        np.save(os.path.join(tmp, "out_rank%d.npy" % rank), np.stack(out))
    q.put((rank, None if out is None else len(out)))
    ranks.close()


def test_two_ranks_on_one_gpu_lipsync_a_clip_like_one_process(cuda, tmp_path):
    """inference.lipsync with the mel chunks sharded over two ranks (ragged: batches of 8 over an odd number of chunks) against the
    single-process call: same frames in the same order on rank 0, nothing on rank 1.  A rank's batches hold other frames than the
    single process's batches, and launch configurations are a function of the batch size: identical up to one uint8 level on a
    handful of pixels."""
    from wav2lip_amd import inference, models
    tmp = str(tmp_path)
    res = _launch(_infer_worker, 2, tmp)
    G = _load(models.Wav2Lip, 0, cuda).eval()
    ref = np.stack(inference.lipsync(G, list(synth.face_crops_u8(9, seed=4)), synth.sine_wav(1.5), fps=25., batch_size=8))
    assert res[0][1] == len(ref) and res[1][1] is None and len(ref) > 16
    got = np.load(os.path.join(tmp, "out_rank0.npy"))
    assert got.shape == ref.shape
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-3
