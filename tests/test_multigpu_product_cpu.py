"""The product's own entry points under a multi-rank launch, on CPU with gloo: `trainer.main_wav2lip_train`
(wav2lip_train.py:351-374) and `inference.lipsync` / `inference.main` (inference.py:181-277) run as N processes with the launch
environment torch.distributed.run provides.  The HIP kernels are replaced at the seams the host logic calls them through
(the networks' constructors, the two loss functions, the optimiser class, the runner, the mel) by torch-CPU stand-ins that keep the
protocol (one autograd node per network that hands its gradients to the attached reducer); everything between those seams is the
product's code: init_from_env, the broadcast of rank 0's weights, GradReducer inside backward, one writer, the shard / gather of
frames in order."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(target, world, *args, timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


def _env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))


# ---------------------------------------------------------------- training
class StubNet(torch.nn.Module):
    """a mirrored network without HIP: ONE autograd node whose backward walks its blocks last to first and hands each block's fresh
    gradients to the reducer attached to `_train_graphs` (autograd.TrainGraph.backward's protocol), keyed by data_ptr as there"""

    def __init__(self, dims, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.ws = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(a, b, generator=g) * 0.3) for a, b in zip(dims, dims[1:])])
        self.register_buffer("calls", torch.zeros((), dtype=torch.int64))
        self._train_graphs = SimpleNamespace(reducer=None)

    def _run(self, x):
        net = self

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, *ws):
                acts = [x]
                for w in ws:
                    acts.append(torch.tanh(acts[-1] @ w))
                ctx.acts, ctx.ws = acts, ws
                return acts[-1]

            @staticmethod
            def backward(ctx, gy):
                red = net._train_graphs.reducer
                grads = {}
                for i in reversed(range(len(ctx.ws))):
                    gz = gy * (1 - ctx.acts[i + 1] ** 2)
                    fresh = {ctx.ws[i].data_ptr(): ctx.acts[i].t() @ gz}
                    if red is not None:
                        red.on_grads(fresh)
                    else:
                        grads.update(fresh)
                    gy = gz @ ctx.ws[i].t()
                if red is not None:
                    grads = red.finalize()
                return (gy,) + tuple(grads[w.data_ptr()] for w in ctx.ws)
        return Fn.apply(x, *self.ws)


class StubGenerator(StubNet):
    def __init__(self):
        super().__init__([6, 7, 5], seed=int(os.environ.get("STUB_SEED", "0")))

    def forward(self, indiv_mels, x):
        return self._run(x + indiv_mels.sum(dim=1, keepdim=True) * 0.01)


class StubSync(StubNet):
    def __init__(self):
        super().__init__([5, 3], seed=77)


def _batches(n_batches, per, seed):
    g = torch.Generator().manual_seed(seed)
    # (x, indiv_mels, mel, gt); mel is wide so that the evaluation's averaged sync loss stays above the 0.75 switch
    return [(torch.randn(per, 6, generator=g), torch.randn(per, 2, generator=g), torch.randn(per, 3, generator=g) * 3,
             torch.randn(per, 5, generator=g)) for _ in range(n_batches)]


def _patch_training(batches_for_rank, samples_log):
    from wav2lip_amd import losses, models, optim, trainer
    models.Wav2Lip = StubGenerator
    models.SyncNet_color = StubSync
    optim.Adam = torch.optim.Adam
    losses.l1_loss = lambda g, gt: (g - gt).abs().mean()
    losses.get_sync_loss = lambda syncnet, mel, g: (syncnet._run(g) - mel).square().mean()

    class Loader(list):
        pass
    trainer._loaders = lambda data_root, device, bs, kind, rng=None: (Loader(batches_for_rank), Loader(batches_for_rank[:1]))
    trainer.save_sample_images = lambda x, g, gt, step, d: samples_log.append(step)
    return trainer


def _train_worker(rank, world, port, q, tmp):
    _env(rank, world, port)
    os.environ["STUB_SEED"] = str(100 + rank)          # every rank's constructor draws its OWN initial weights
    samples = []
    glob = _batches(2, 4 * world, seed=9)               # two global batches; rank r trains on rows [4r, 4r+4) of each
    mine = [tuple(t[4 * rank:4 * rank + 4] for t in b) for b in glob]
    trainer = _patch_training(mine, samples)
    from wav2lip_amd.hparams import hparams
    hparams.set_hparam("syncnet_wt", 0.03)
    run = trainer.main_wav2lip_train(["--data_root", tmp, "--checkpoint_dir", os.path.join(tmp, "ckpt"),
                                      "--syncnet_checkpoint_path", os.path.join(tmp, "sync.pth")], max_steps=2, backend="gloo")
    import torch.distributed as dist
    state = torch.load(os.path.join(tmp, "ckpt", "checkpoint_step%09d.pth" % 1), weights_only=False) if rank == 0 else None
    q.put((rank, dict(step=run.global_step, samples=samples, group_closed=not dist.is_initialized(),
                      ckpt_keys=sorted(state["state_dict"]) if state else None, params=None)))
    # the weights after two steps travel through a file (queues and tensors of a dying process do not mix well)
    torch.save([p.detach().clone() for p in run_model[0].parameters()], os.path.join(tmp, "final_rank%d.pt" % rank))


run_model = []


def _train_worker_entry(rank, world, port, q, tmp):
    # main_* builds the model inside; keep a handle on it through the constructor
    orig_init = StubGenerator.__init__

    def init(self):
        orig_init(self)
        run_model.append(self)
    StubGenerator.__init__ = init
    _train_worker(rank, world, port, q, tmp)


def test_main_wav2lip_train_on_two_ranks_broadcasts_reduces_in_backward_and_writes_once(tmp_path):
    tmp = str(tmp_path)
    torch.save({"state_dict": StubSync().state_dict(), "optimizer": None, "global_step": 0, "global_epoch": 0},
               os.path.join(tmp, "sync.pth"))
    world = 2
    res = _launch(_train_worker_entry, world, tmp)
    assert [r[1]["step"] for r in res] == [2, 2]
    assert all(r[1]["group_closed"] for r in res)
    # one writer: rank 0 wrote the sample images of step 0 and the checkpoint of step 1; rank 1 wrote nothing
    assert res[0][1]["samples"] == [0] and res[1][1]["samples"] == []
    assert sorted(os.listdir(os.path.join(tmp, "ckpt"))) == ["checkpoint_step000000001.pth"]
    assert res[0][1]["ckpt_keys"] == ["calls", "ws.0", "ws.1"]
    finals = [torch.load(os.path.join(tmp, "final_rank%d.pt" % r)) for r in range(world)]
    for a, b in zip(*finals):
        assert torch.equal(a, b)                        # same weights everywhere after two steps
    # ... and they are what ONE process computes from rank 0's initial weights on the global batches (mean losses: the mean of
    # the per-shard gradients is the gradient of the global batch)
    os.environ["STUB_SEED"] = "100"
    G, S = StubGenerator(), StubSync()
    opt = torch.optim.Adam(G.parameters(), lr=1e-4)
    from wav2lip_amd.hparams import hparams
    assert hparams.initial_learning_rate == 1e-4
    for x, indiv, mel, gt in _batches(2, 4 * world, seed=9):
        opt.zero_grad()
        g = G(indiv, x)
        loss = 0.03 * (S._run(g) - mel).square().mean() + 0.97 * (g - gt).abs().mean()
        loss.backward()
        opt.step()
    for a, b in zip(finals[0], G.parameters()):
        assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()


# ---------------------------------------------------------------- inference
class FakeRunner:
    """stands where inference.PipelinedRunner stands: same submit / result / depth surface, frames = a pure function of the batch"""

    def __init__(self, model, batch_size=128, depth=2):
        self.depth = depth

    def submit(self, faces_u8, mel_windows=None, mel=None, starts=None, frames=None, frame_idx=None, boxes=None):
        if frames is not None:                                  # run_frames route: paste a starts-coded patch into each frame
            out = frames[torch.tensor(frame_idx)].clone()
            for k, (y1, y2, x1, x2) in enumerate(boxes):
                out[k, y1:y2, x1:x2] = (out[k, y1:y2, x1:x2].to(torch.int32) + int(starts[k]) * 7 + 1).to(torch.uint8)
            return out
        return (faces_u8.to(torch.int32) + starts.view(-1, 1, 1, 1) * 7 + 1).to(torch.uint8)

    @staticmethod
    def result(ticket):
        return ticket


def _patch_inference(monkeypatch=None):
    """in a worker process: plain assignment; in the pytest process: through `monkeypatch`, undone after the test"""
    from wav2lip_amd import audio, inference
    put = setattr if monkeypatch is None else monkeypatch.setattr
    put(inference, "PipelinedRunner", FakeRunner)
    put(audio, "melspectrogram_device", lambda wav, dev: torch.arange(80 * 61, dtype=torch.float32).view(80, 61))
    put(inference, "load_model", lambda path, dev: torch.nn.Linear(1, 1))
    return inference


def _faces(n):
    r = np.random.default_rng(3)
    return [r.integers(0, 255, (96, 96, 3), dtype=np.uint8) for _ in range(n)]


def _lipsync_worker(rank, world, port, q, fps, box, tmp):
    _env(rank, world, port)
    inference = _patch_inference()
    from wav2lip_amd import sharding
    ranks = sharding.init_from_env("gloo")
    model = torch.nn.Linear(1, 1)
    frames = _faces(7) if box is None else [np.pad(f, ((8, 8), (16, 16), (0, 0))) for f in _faces(7)]
    out = inference.lipsync(model, frames, np.zeros(16000, np.float32), fps=fps, batch_size=3, box=box, ranks=ranks)
    if out is not None:
        np.save(os.path.join(tmp, "out_rank%d.npy" % rank), np.stack(out))
    q.put((rank, None if out is None else len(out)))
    ranks.close()


@pytest.mark.parametrize("world,fps,box", [(3, 25., None), (4, 3., None), (2, 25., (8, 90, 16, 100))])
def test_lipsync_shards_the_mel_chunks_and_rank_0_gets_every_frame_in_order(tmp_path, monkeypatch, world, fps, box):
    """61 mel columns: 16 chunks at 25 fps (shards 6/6/4 over three ranks in batches of 3: ragged rounds), 3 chunks at 3 fps over
    four ranks (shards 1/1/1/0: an empty shard still joins every collective); the third case takes the frames-on-device route"""
    tmp = str(tmp_path)
    res = _launch(_lipsync_worker, world, fps, box, tmp)
    inference = _patch_inference(monkeypatch)
    frames = _faces(7) if box is None else [np.pad(f, ((8, 8), (16, 16), (0, 0))) for f in _faces(7)]
    ref = np.stack(inference.lipsync(torch.nn.Linear(1, 1), frames, np.zeros(16000, np.float32), fps=fps, batch_size=3, box=box))
    assert res[0][1] == len(ref) and all(r[1] is None for r in res[1:])
    assert sorted(os.listdir(tmp)) == ["out_rank0.npy"]
    assert np.array_equal(np.load(os.path.join(tmp, "out_rank0.npy")), ref)


def _main_worker(rank, world, port, q, tmp):
    _env(rank, world, port)
    inference = _patch_inference()
    out = inference.main(["--checkpoint_path", "x", "--face", os.path.join(tmp, "face.png"), "--audio", os.path.join(tmp, "a.wav"),
                          "--outfile", os.path.join(tmp, "out", "result_rank%d.avi" % 0), "--box", "4", "60", "8", "72",
                          "--wav2lip_batch_size", "4"], backend="gloo")
    q.put((rank, None if out is None else len(out)))


def test_inference_main_on_three_ranks_writes_one_file_from_rank_0(tmp_path, monkeypatch):
    from PIL import Image
    from scipy.io import wavfile
    tmp = str(tmp_path)
    Image.fromarray(_faces(1)[0]).save(os.path.join(tmp, "face.png"))
    wavfile.write(os.path.join(tmp, "a.wav"), 16000, (np.sin(np.arange(16000) * 0.05) * 8000).astype(np.int16))
    res = _launch(_main_worker, 3, tmp)
    assert res[0][1] == 16 and res[1][1] is None and res[2][1] is None        # 61 mel columns at 25 fps: 16 chunks
    assert os.listdir(os.path.join(tmp, "out")) == ["result_rank0.avi"]
    from wav2lip_amd import container
    clip = container.read_avi(os.path.join(tmp, "out", "result_rank0.avi"))
    # the single-process run of the same command line writes the same frames
    inference = _patch_inference(monkeypatch)
    monkeypatch.setattr(inference, "args", inference.args)        # main() replaces the module-level args
    ref = inference.main(["--checkpoint_path", "x", "--face", os.path.join(tmp, "face.png"), "--audio", os.path.join(tmp, "a.wav"),
                          "--outfile", os.path.join(tmp, "ref.avi"), "--box", "4", "60", "8", "72", "--wav2lip_batch_size", "4"],
                         backend="gloo")
    assert len(clip["frames"]) == len(ref) == 16
    assert all(np.array_equal(a, b) for a, b in zip(clip["frames"], ref))


# ---------------------------------------------------------------- the pieces
def _pieces_worker(rank, world, port, q):
    _env(rank, world, port)
    from wav2lip_amd import sharding
    ranks = sharding.init_from_env("gloo")
    ok = ranks.world == world and ranks.rank == rank and ranks.writer == (rank == 0) and ranks.device.type == "cpu"
    torch.manual_seed(rank)
    m = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4))
    with torch.no_grad():
        m[1].running_mean.fill_(float(rank))
        m[1].num_batches_tracked.fill_(rank + 5)
    v0 = m[0].weight._version
    n = sharding.broadcast_state(ranks.dist, m, m)          # a module listed twice is broadcast once
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4))
    ok = ok and n == 2                                        # one fp32 bucket, one int64 bucket
    ok = ok and torch.equal(m[0].weight, ref[0].weight) and torch.equal(m[0].bias, ref[0].bias)
    ok = ok and float(m[1].running_mean[0]) == 0.0 and int(m[1].num_batches_tracked) == 5
    ok = ok and m[0].weight._version > v0                    # written in place: packed device weights are rebuilt
    # rects of a sharded detection come back for every frame, in frame order, on every rank
    from wav2lip_amd import inference

    class Det:
        def get_detections_for_batch(self, images):
            return [(int(im[0, 0, 0]), 1, 2, 3) for im in images]
    images = [np.full((4, 4, 3), i, dtype=np.uint8) for i in range(7)]
    rects = inference._detect_rects(images, Det(), 2, ranks)
    ok = ok and rects == [(i, 1, 2, 3) for i in range(7)]
    q.put((rank, bool(ok)))
    ranks.close()


def test_init_from_env_broadcast_state_and_sharded_detection_on_three_ranks():
    assert [r[1] for r in _launch(_pieces_worker, 3)] == [True, True, True]


def test_init_from_env_without_a_launcher_is_the_single_process_run():
    from wav2lip_amd import sharding
    r = sharding.init_from_env("gloo", env={})
    assert r.dist is None and r.world == 1 and r.rank == 0 and r.writer
    assert sharding.broadcast_state(None, torch.nn.Linear(2, 2)) == 0
    r.close()
    with pytest.raises(RuntimeError, match="outside WORLD_SIZE"):
        sharding.init_from_env("gloo", env={"WORLD_SIZE": "2", "RANK": "5", "LOCAL_RANK": "0"})
