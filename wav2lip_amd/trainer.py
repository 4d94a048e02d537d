"""The three training scripts of the reference as loops over the HIP step bodies (wav2lip_amd/train.py):

    wav2lip_train.py        train :200-269, eval_model :272-303, save_sample_images :166-177, checkpoints :305-349, main :351-374
    hq_wav2lip_train.py     train :202-298, eval_model :300-351, checkpoints :354-398, main :400-443
    color_syncnet_train.py  train :140-178, eval_model :180-206, checkpoints :208-247, main :249-281

What the scripts keep in module globals (`global_step`, `global_epoch`, the frozen `syncnet`) lives in a `Run` object; the
loops themselves keep the reference's order of events and its quirks, because resumed runs and their checkpoints depend on
them: the step counter is incremented BEFORE the checkpoint / evaluation tests, step 1 always checkpoints (and, for
wav2lip_train, evaluates), sample images are written on the PRE-increment counter, checkpoint files are named after the
global counter, an evaluation that averages a sync loss below 0.75 raises `hparams.syncnet_wt` to 0.01 (0.03 in the hq
script) for the rest of the run, the frozen expert is loaded with `reset_optimizer=True, overwrite_global_states=False` and
is never put in eval mode, evaluation runs 700 / 300 / 1400 steps under `torch.no_grad()`.

A data loader is any iterable of batches in the reference's layout - the reference's own `torch.utils.data.DataLoader` over
its `Dataset`, or `ClipLoader` below over the device-resident `data.ClipStore`.

Multi-GPU (BASELINE configs[3]/[4]): `python -m torch.distributed.run --nproc-per-node 8 -m wav2lip_amd.trainer
wav2lip_train ...` - every `main_*` reads WORLD_SIZE / RANK / LOCAL_RANK, binds cuda:LOCAL_RANK, joins the RCCL group, broadcasts
rank 0's parameters and buffers once (after the checkpoints are loaded), attaches ONE `sharding.GradReducer` to the trained
networks (bucketed all-reduce overlapped with backward) and hands `dist` to the loop: one writer of checkpoints / sample images,
one shared evaluation average, per-rank data.  The loops also accept `dist=torch.distributed` WITHOUT a reducer attached
(gradients averaged after backward).
"""
import argparse
import os

import numpy as np
import torch

from . import losses, train
from .checkpoint import _load, strip_module_prefix
from .hparams import hparams


class Run:
    """the module globals of a training script: counters, checkpoint directory, the frozen expert"""

    def __init__(self, checkpoint_dir, syncnet=None):
        self.global_step = 0
        self.global_epoch = 0
        self.checkpoint_dir = checkpoint_dir
        self.syncnet = syncnet
        self.log = print


# ---------------------------------------------------------------- checkpoints
def save_checkpoint(run, model, optimizer, step, checkpoint_dir, epoch, prefix=''):
    """wav2lip_train.py:305-317 / hq_wav2lip_train.py:354-365: the FILE NAME carries the run's global step, the payload the
    `step` / `epoch` arguments (the reference passes the same values)"""
    path = os.path.join(checkpoint_dir, "{}checkpoint_step{:09d}.pth".format(prefix, run.global_step))
    optimizer_state = optimizer.state_dict() if hparams.save_optimizer_state else None
    torch.save({"state_dict": model.state_dict(), "optimizer": optimizer_state, "global_step": step, "global_epoch": epoch}, path)
    run.log("Saved checkpoint:", path)
    return path


def load_checkpoint(run, path, model, optimizer, reset_optimizer=False, overwrite_global_states=True, strip_module=True):
    """wav2lip_train.py:327-349 (`strip_module=False`: color_syncnet_train.py:225-241, which loads the keys as they are)"""
    run.log("Load checkpoint from: {}".format(path))
    checkpoint = _load(path)
    s = checkpoint["state_dict"]
    model.load_state_dict(strip_module_prefix(s) if strip_module else s)
    if not reset_optimizer:
        optimizer_state = checkpoint["optimizer"]
        if optimizer_state is not None:
            run.log("Load optimizer state from {}".format(path))
            optimizer.load_state_dict(checkpoint["optimizer"])
    if overwrite_global_states:
        run.global_step = checkpoint["global_step"]
        run.global_epoch = checkpoint["global_epoch"]
    return model


def save_sample_images(x, g, gt, global_step, checkpoint_dir):
    """wav2lip_train.py:166-177: collage [reference | masked input | generated | ground truth] per (sample, t) as JPEG files.
    Same uint8 arithmetic (x255., truncation); written with PIL instead of cv2.imwrite (BGR -> RGB for the encoder)."""
    from PIL import Image
    x = (x.detach().cpu().numpy().transpose(0, 2, 3, 4, 1) * 255.).astype(np.uint8)
    g = (g.detach().cpu().numpy().transpose(0, 2, 3, 4, 1) * 255.).astype(np.uint8)
    gt = (gt.detach().cpu().numpy().transpose(0, 2, 3, 4, 1) * 255.).astype(np.uint8)
    refs, inps = x[..., 3:], x[..., :3]
    folder = os.path.join(checkpoint_dir, "samples_step{:09d}".format(global_step))
    if not os.path.exists(folder):
        os.mkdir(folder)
    collage = np.concatenate((refs, inps, g, gt), axis=-2)
    for batch_idx, c in enumerate(collage):
        for t in range(len(c)):
            Image.fromarray(np.ascontiguousarray(c[t][:, :, ::-1])).save('{}/{}_{}.jpg'.format(folder, batch_idx, t))
    return folder


def _is_writer(dist):
    """checkpoints and sample images are written by ONE rank: all ranks hold the same weights after the gradient all-reduce,
    and concurrent torch.save calls to one path would corrupt it (the reference is single-process and has no such case)"""
    return dist is None or dist.get_rank() == 0


def _rank_mean(dist, value, device):
    """the evaluation average over all ranks (each rank evaluates its own validation batches): the `< .75 -> set syncnet_wt`
    switch (wav2lip_train.py:258-260, hq_wav2lip_train.py:285-287) must flip on every rank in the same step"""
    if dist is None or dist.get_world_size() == 1:
        return value
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()


def _grad_sync(dist, *modules):
    """what a step body gets as its `dist`: None when a sharding.GradReducer attached to every one of `modules` already
    averages the gradients INSIDE backward (overlapped with it), else `dist` (bucketed all-reduce after backward).  `dist`
    itself stays the loops' control plane either way: one writer, one shared evaluation average."""
    if dist is None:
        return None
    if all(getattr(getattr(m, "_train_graphs", None), "reducer", None) is not None for m in modules):
        return None
    return dist


def _to(device, *tensors):
    return tuple(t.to(device) for t in tensors)


def _val(v):
    return v.item() if torch.is_tensor(v) else float(v)


# ---------------------------------------------------------------- wav2lip_train.py
def eval_wav2lip(run, test_data_loader, device, model, eval_steps=700):
    """wav2lip_train.py:272-303; returns the averaged sync loss"""
    run.log('Evaluating for {} steps'.format(eval_steps))
    sync_losses, recon_losses = [], []
    step = 0
    while 1:
        for x, indiv_mels, mel, gt in test_data_loader:
            step += 1
            model.eval()
            x, gt, indiv_mels, mel = _to(device, x, gt, indiv_mels, mel)
            g = model(indiv_mels, x)
            sync_losses.append(losses.get_sync_loss(run.syncnet, mel, g).item())
            recon_losses.append(losses.l1_loss(g, gt).item())
            if step > eval_steps:
                averaged_sync_loss = sum(sync_losses) / len(sync_losses)
                run.log('L1: {}, Sync loss: {}'.format(sum(recon_losses) / len(recon_losses), averaged_sync_loss))
                return averaged_sync_loss


def train_wav2lip(run, device, model, train_data_loader, test_data_loader, optimizer, checkpoint_dir=None,
                  checkpoint_interval=None, nepochs=None, eval_steps=700, dist=None, max_steps=None):
    """wav2lip_train.py:200-269.  `max_steps` (not in the reference) stops after that many steps of THIS session."""
    checkpoint_dir = checkpoint_dir or run.checkpoint_dir
    checkpoint_interval = hparams.checkpoint_interval if checkpoint_interval is None else checkpoint_interval
    nepochs = hparams.nepochs if nepochs is None else nepochs
    resumed_step = run.global_step
    while run.global_epoch < nepochs:
        run.log('Starting Epoch: {}'.format(run.global_epoch))
        running_sync_loss, running_l1_loss = 0., 0.
        for step, (x, indiv_mels, mel, gt) in enumerate(train_data_loader):
            x, mel, indiv_mels, gt = _to(device, x, mel, indiv_mels, gt)
            loss, l1loss, sync_loss, g = train.wav2lip_train_step(model, run.syncnet, optimizer, x, indiv_mels, mel, gt,
                                                                  dist=_grad_sync(dist, model), return_generated=True)
            if run.global_step % checkpoint_interval == 0 and _is_writer(dist):
                save_sample_images(x, g, gt, run.global_step, checkpoint_dir)
            run.global_step += 1
            running_l1_loss += l1loss.item()
            running_sync_loss += _val(sync_loss) if hparams.syncnet_wt > 0. else 0.
            if (run.global_step == 1 or run.global_step % checkpoint_interval == 0) and _is_writer(dist):
                save_checkpoint(run, model, optimizer, run.global_step, checkpoint_dir, run.global_epoch)
            if run.global_step == 1 or run.global_step % hparams.eval_interval == 0:
                with torch.no_grad():
                    average_sync_loss = _rank_mean(dist, eval_wav2lip(run, test_data_loader, device, model, eval_steps), device)
                    if average_sync_loss < .75:
                        hparams.set_hparam('syncnet_wt', 0.01)   # without image GAN a lesser weight is sufficient
            run.last_description = 'L1: {}, Sync Loss: {}'.format(running_l1_loss / (step + 1), running_sync_loss / (step + 1))
            if max_steps is not None and run.global_step - resumed_step >= max_steps:
                return run
        run.global_epoch += 1
    return run


# ---------------------------------------------------------------- hq_wav2lip_train.py
def eval_hq(run, test_data_loader, device, model, disc, eval_steps=300):
    """hq_wav2lip_train.py:300-351: ONE pass over the loader (at most eval_steps + 2 batches); returns the averaged sync loss"""
    run.log('Evaluating for {} steps'.format(eval_steps))
    r_sync, r_l1, r_real, r_fake, r_perc = [], [], [], [], []
    for step, (x, indiv_mels, mel, gt) in enumerate(test_data_loader):
        model.eval()
        disc.eval()
        x, mel, indiv_mels, gt = _to(device, x, mel, indiv_mels, gt)
        pred = disc(gt)
        r_real.append(losses.bce_mean(pred, torch.ones((len(pred), 1), device=pred.device)).item())
        g = model(indiv_mels, x)
        pred = disc(g)
        r_fake.append(losses.bce_mean(pred, torch.zeros((len(pred), 1), device=pred.device)).item())
        r_sync.append(losses.get_sync_loss(run.syncnet, mel, g).item())
        r_perc.append(disc.perceptual_forward(g).item() if hparams.disc_wt > 0. else 0.)
        r_l1.append(losses.l1_loss(g, gt).item())
        if step > eval_steps:
            break
    run.log('L1: {}, Sync: {}, Percep: {} | Fake: {}, Real: {}'.format(
        sum(r_l1) / len(r_l1), sum(r_sync) / len(r_sync), sum(r_perc) / len(r_perc), sum(r_fake) / len(r_fake),
        sum(r_real) / len(r_real)))
    return sum(r_sync) / len(r_sync)


def train_hq(run, device, model, disc, train_data_loader, test_data_loader, optimizer, disc_optimizer, checkpoint_dir=None,
             checkpoint_interval=None, nepochs=None, eval_steps=300, dist=None, max_steps=None, gather_frames=None):
    """hq_wav2lip_train.py:202-298.  `gather_frames=dist`: the discriminator trains on the all-gathered global batch (BASELINE
    configs[4], train.hq_train_step)."""
    checkpoint_dir = checkpoint_dir or run.checkpoint_dir
    checkpoint_interval = hparams.checkpoint_interval if checkpoint_interval is None else checkpoint_interval
    nepochs = hparams.nepochs if nepochs is None else nepochs
    resumed_step = run.global_step
    while run.global_epoch < nepochs:
        run.log('Starting Epoch: {}'.format(run.global_epoch))
        tot = dict(l1=0., sync=0., perceptual=0., disc_real=0., disc_fake=0.)
        for step, (x, indiv_mels, mel, gt) in enumerate(train_data_loader):
            x, mel, indiv_mels, gt = _to(device, x, mel, indiv_mels, gt)
            out = train.hq_train_step(model, disc, run.syncnet, optimizer, disc_optimizer, x, indiv_mels, mel, gt,
                                      dist=_grad_sync(dist, model, disc), return_generated=True, gather_frames=gather_frames)
            if run.global_step % checkpoint_interval == 0 and _is_writer(dist):
                save_sample_images(x, out["g"], gt, run.global_step, checkpoint_dir)
            run.global_step += 1
            for k in tot:
                tot[k] += _val(out[k])
            if (run.global_step == 1 or run.global_step % checkpoint_interval == 0) and _is_writer(dist):
                save_checkpoint(run, model, optimizer, run.global_step, checkpoint_dir, run.global_epoch)
                save_checkpoint(run, disc, disc_optimizer, run.global_step, checkpoint_dir, run.global_epoch, prefix='disc_')
            if run.global_step % hparams.eval_interval == 0:
                with torch.no_grad():
                    average_sync_loss = _rank_mean(dist, eval_hq(run, test_data_loader, device, model, disc, eval_steps), device)
                    if average_sync_loss < .75:
                        hparams.set_hparam('syncnet_wt', 0.03)
            run.last_description = 'L1: {}, Sync: {}, Percep: {} | Fake: {}, Real: {}'.format(
                tot["l1"] / (step + 1), tot["sync"] / (step + 1), tot["perceptual"] / (step + 1), tot["disc_fake"] / (step + 1),
                tot["disc_real"] / (step + 1))
            if max_steps is not None and run.global_step - resumed_step >= max_steps:
                return run
        run.global_epoch += 1
    return run


# ---------------------------------------------------------------- color_syncnet_train.py
def eval_syncnet(run, test_data_loader, device, model, eval_steps=1400):
    """color_syncnet_train.py:180-206: one pass over the loader (at most eval_steps + 2 batches); returns the averaged loss"""
    run.log('Evaluating for {} steps'.format(eval_steps))
    vals = []
    for step, (x, mel, y) in enumerate(test_data_loader):
        model.eval()
        x, mel, y = _to(device, x, mel, y)
        a, v = model(mel, x)
        vals.append(losses.cosine_loss(a, v, y).item())
        if step > eval_steps:
            break
    averaged_loss = sum(vals) / len(vals)
    run.log(averaged_loss)
    return averaged_loss


def train_syncnet(run, device, model, train_data_loader, test_data_loader, optimizer, checkpoint_dir=None,
                  checkpoint_interval=None, nepochs=None, eval_steps=1400, dist=None, max_steps=None):
    """color_syncnet_train.py:140-178"""
    checkpoint_dir = checkpoint_dir or run.checkpoint_dir
    checkpoint_interval = hparams.syncnet_checkpoint_interval if checkpoint_interval is None else checkpoint_interval
    nepochs = hparams.nepochs if nepochs is None else nepochs
    resumed_step = run.global_step
    while run.global_epoch < nepochs:
        running_loss = 0.
        for step, (x, mel, y) in enumerate(train_data_loader):
            x, mel, y = _to(device, x, mel, y)
            loss = train.syncnet_train_step(model, optimizer, x, mel, y, dist=_grad_sync(dist, model))
            run.global_step += 1
            running_loss += loss.item()
            if (run.global_step == 1 or run.global_step % checkpoint_interval == 0) and _is_writer(dist):
                save_checkpoint(run, model, optimizer, run.global_step, checkpoint_dir, run.global_epoch)
            if run.global_step % hparams.syncnet_eval_interval == 0:
                with torch.no_grad():
                    eval_syncnet(run, test_data_loader, device, model, eval_steps)
            run.last_description = 'Loss: {}'.format(running_loss / (step + 1))
            if max_steps is not None and run.global_step - resumed_step >= max_steps:
                return run
        run.global_epoch += 1
    return run


# ---------------------------------------------------------------- device-resident loader
class ClipLoader:
    """Iterable over `data.ClipStore` with the DataLoader's epoch length (ceil(clips / batch_size): the reference's Dataset has
    one item per clip, hparams.py:4-13 + wav2lip_train.py:108-109): every batch is drawn by the store's sampling loop (the
    reference's `__getitem__` rules) and assembled on the device.  kind: "generator" -> (x, indiv_mels, mel, gt);
    "syncnet" -> (x, mel, y)."""

    def __init__(self, store, batch_size, kind="generator", rng=None):
        import random
        self.store, self.batch_size, self.kind = store, int(batch_size), kind
        self.rng = rng or random
        if kind not in ("generator", "syncnet"):
            raise ValueError("kind must be 'generator' or 'syncnet'")

    def __len__(self):
        return (len(self.store) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for _ in range(len(self)):
            if self.kind == "generator":
                yield self.store.sample_generator_batch(self.batch_size, self.rng)[:4]
            else:
                yield self.store.sample_syncnet_batch(self.batch_size, self.rng)[:3]


# ---------------------------------------------------------------- the scripts' command lines
def _common_parser(description, syncnet_ckpt, disc_ckpt=False, typed_root=True):
    p = argparse.ArgumentParser(description=description)
    if typed_root:
        p.add_argument("--data_root", help="Root folder of the preprocessed LRS2 dataset", required=True, type=str)
    else:       # color_syncnet_train.py:21 declares the flag without a type
        p.add_argument("--data_root", help="Root folder of the preprocessed LRS2 dataset", required=True)
    p.add_argument('--checkpoint_dir', help='Save checkpoints to this directory', required=True, type=str)
    if syncnet_ckpt:
        p.add_argument('--syncnet_checkpoint_path', help='Load the pre-trained Expert discriminator', required=True, type=str)
    p.add_argument('--checkpoint_path', help='Resume from this checkpoint', default=None, type=str)
    if disc_ckpt:
        p.add_argument('--disc_checkpoint_path', help='Resume quality disc from this checkpoint', default=None, type=str)
    return p


def wav2lip_train_parser():
    return _common_parser('Code to train the Wav2Lip model without the visual quality discriminator', True)


def hq_wav2lip_train_parser():
    return _common_parser('Code to train the Wav2Lip model WITH the visual quality discriminator', True, True)


def color_syncnet_train_parser():
    return _common_parser('Code to train the expert lip-sync discriminator', False, typed_root=False)


def _loader_rng(ranks):
    """the sampler of a rank.  One process: the `random` module itself, as the reference's Dataset uses it.  Several ranks must
    draw DIFFERENT samples, reproducibly when asked: W2L_DATA_SEED=<int> seeds rank r with seed + r (without it every process's
    `random` is seeded from OS entropy, as the reference's DataLoader workers are)."""
    import random
    seed = os.environ.get("W2L_DATA_SEED")
    if seed is None:
        return random
    return random.Random(int(seed) + ranks.rank)


def _loaders(data_root, device, batch_size, kind, rng=None):
    from .data import ClipStore
    train_store = ClipStore.from_directory(data_root, 'train', device)
    val_store = ClipStore.from_directory(data_root, 'val', device)
    return ClipLoader(train_store, batch_size, kind, rng=rng), ClipLoader(val_store, batch_size, kind, rng=rng)


def _start_job(backend):
    """one process per GPU (sharding.init_from_env): binds cuda:LOCAL_RANK and joins the RCCL group when launched under
    torch.distributed.run; a plain `python -m wav2lip_amd.trainer ...` is the reference's single-process run"""
    from . import sharding
    ranks = sharding.init_from_env(backend)
    if ranks.dist is not None:
        print("rank {} of {} on {}".format(ranks.rank, ranks.world, ranks.device))
    return ranks


def _make_data_parallel(ranks, trained, frozen=(), bucket_mb=32):
    """After the checkpoints are loaded: every rank takes rank 0's parameters and buffers (each constructor drew its own
    initialisation), then ONE GradReducer is attached to the trained networks so that their backward passes return gradients
    already averaged over the ranks, the all-reduces of late layers' buckets running while earlier layers' kernels execute.
    Returns the reducer (None in a single-process run)."""
    if ranks.dist is None:
        return None
    from . import sharding
    sharding.broadcast_state(ranks.dist, *trained, *frozen)
    reducer = sharding.GradReducer(ranks.dist, bucket_bytes=bucket_mb << 20)
    reducer.attach(*trained)
    return reducer


def _log_for(ranks):
    """progress lines come from the writer rank only (N copies of every line help nobody)"""
    return print if ranks.writer else (lambda *a, **k: None)


def _make_checkpoint_dir(ranks, path):
    if ranks.writer and not os.path.exists(path):
        os.mkdir(path)
    if ranks.dist is not None:
        ranks.dist.barrier()       # nobody trains before the directory exists


def main_wav2lip_train(argv=None, max_steps=None, backend="nccl"):
    """wav2lip_train.py:19-29 (flags) + :351-374; under torch.distributed.run: BASELINE configs[3], one rank per GPU, batch
    `hparams.batch_size` PER RANK, gradients averaged over RCCL inside backward"""
    from . import models, optim
    a = wav2lip_train_parser().parse_args(argv)
    ranks = _start_job(backend)
    device = ranks.device
    tr, te = _loaders(a.data_root, device, hparams.batch_size, "generator", _loader_rng(ranks))
    model = models.Wav2Lip().to(device)
    syncnet = models.SyncNet_color().to(device)
    for p in syncnet.parameters():
        p.requires_grad = False
    run = Run(a.checkpoint_dir, syncnet)
    run.log = _log_for(ranks)
    run.log('total trainable params {}'.format(sum(p.numel() for p in model.parameters() if p.requires_grad)))
    optimizer = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=hparams.initial_learning_rate)
    if a.checkpoint_path is not None:
        load_checkpoint(run, a.checkpoint_path, model, optimizer, reset_optimizer=False)
    load_checkpoint(run, a.syncnet_checkpoint_path, syncnet, None, reset_optimizer=True, overwrite_global_states=False)
    _make_checkpoint_dir(ranks, a.checkpoint_dir)
    _make_data_parallel(ranks, [model], [syncnet])
    try:
        return train_wav2lip(run, device, model, tr, te, optimizer, checkpoint_dir=a.checkpoint_dir,
                             checkpoint_interval=hparams.checkpoint_interval, nepochs=hparams.nepochs, max_steps=max_steps,
                             dist=ranks.dist)
    finally:
        ranks.close()


def main_hq_wav2lip_train(argv=None, max_steps=None, backend="nccl"):
    """hq_wav2lip_train.py:19-30 (flags) + :400-443; under torch.distributed.run: BASELINE configs[4].  W2L_GATHER_FRAMES=1: the
    discriminator trains on the all-gathered global batch of real / generated frames (train.hq_train_step)."""
    from . import models, optim
    a = hq_wav2lip_train_parser().parse_args(argv)
    ranks = _start_job(backend)
    device = ranks.device
    tr, te = _loaders(a.data_root, device, hparams.batch_size, "generator", _loader_rng(ranks))
    model = models.Wav2Lip().to(device)
    disc = models.Wav2Lip_disc_qual().to(device)
    syncnet = models.SyncNet_color().to(device)
    for p in syncnet.parameters():
        p.requires_grad = False
    run = Run(a.checkpoint_dir, syncnet)
    run.log = _log_for(ranks)
    run.log('total trainable params {}'.format(sum(p.numel() for p in model.parameters() if p.requires_grad)))
    run.log('total DISC trainable params {}'.format(sum(p.numel() for p in disc.parameters() if p.requires_grad)))
    optimizer = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=hparams.initial_learning_rate, betas=(0.5, 0.999))
    disc_optimizer = optim.Adam([p for p in disc.parameters() if p.requires_grad], lr=hparams.disc_initial_learning_rate,
                                betas=(0.5, 0.999))
    if a.checkpoint_path is not None:
        load_checkpoint(run, a.checkpoint_path, model, optimizer, reset_optimizer=False)
    if a.disc_checkpoint_path is not None:
        load_checkpoint(run, a.disc_checkpoint_path, disc, disc_optimizer, reset_optimizer=False, overwrite_global_states=False)
    load_checkpoint(run, a.syncnet_checkpoint_path, syncnet, None, reset_optimizer=True, overwrite_global_states=False)
    _make_checkpoint_dir(ranks, a.checkpoint_dir)
    _make_data_parallel(ranks, [model, disc], [syncnet])
    gather = ranks.dist if (ranks.dist is not None and os.environ.get("W2L_GATHER_FRAMES", "0") == "1") else None
    try:
        return train_hq(run, device, model, disc, tr, te, optimizer, disc_optimizer, checkpoint_dir=a.checkpoint_dir,
                        checkpoint_interval=hparams.checkpoint_interval, nepochs=hparams.nepochs, max_steps=max_steps,
                        dist=ranks.dist, gather_frames=gather)
    finally:
        ranks.close()


def main_color_syncnet_train(argv=None, max_steps=None, backend="nccl"):
    """color_syncnet_train.py:19-27 (flags) + :249-281; under torch.distributed.run: data-parallel pairs, `syncnet_batch_size`
    per rank"""
    from . import models, optim
    a = color_syncnet_train_parser().parse_args(argv)
    ranks = _start_job(backend)
    _make_checkpoint_dir(ranks, a.checkpoint_dir)
    device = ranks.device
    tr, te = _loaders(a.data_root, device, hparams.syncnet_batch_size, "syncnet", _loader_rng(ranks))
    model = models.SyncNet_color().to(device)
    run = Run(a.checkpoint_dir)
    run.log = _log_for(ranks)
    run.log('total trainable params {}'.format(sum(p.numel() for p in model.parameters() if p.requires_grad)))
    optimizer = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=hparams.syncnet_lr)
    if a.checkpoint_path is not None:
        load_checkpoint(run, a.checkpoint_path, model, optimizer, reset_optimizer=False, strip_module=False)
    _make_data_parallel(ranks, [model])
    try:
        return train_syncnet(run, device, model, tr, te, optimizer, checkpoint_dir=a.checkpoint_dir,
                             checkpoint_interval=hparams.syncnet_checkpoint_interval, nepochs=hparams.nepochs, max_steps=max_steps,
                             dist=ranks.dist)
    finally:
        ranks.close()


if __name__ == "__main__":
    import sys
    which = {"wav2lip_train": main_wav2lip_train, "hq_wav2lip_train": main_hq_wav2lip_train,
             "color_syncnet_train": main_color_syncnet_train}
    if len(sys.argv) < 2 or sys.argv[1] not in which:
        sys.exit("usage: python -m wav2lip_amd.trainer {wav2lip_train|hq_wav2lip_train|color_syncnet_train} <the script's flags>")
    which[sys.argv[1]](sys.argv[2:])
