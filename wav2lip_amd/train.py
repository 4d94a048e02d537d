"""The bodies of the reference's three training loops on the HIP path, plus the sample-window index arithmetic of
their `Dataset` classes (host integer/double math, bit-exact by construction).

    syncnet_train_step   color_syncnet_train.py:149-165   a, v = model(mel, x); cosine_loss; backward; Adam
    wav2lip_train_step   wav2lip_train.py:210-230         g = model(indiv_mels, x); sync + L1; backward; Adam
    hq_train_step        hq_wav2lip_train.py:212-257      + disc.perceptual_forward(g); then D(real)/D(fake) BCE, 2nd Adam

Every network call is one autograd node whose forward/backward are HIP launch sequences (wav2lip_amd/autograd.py); the
losses are HIP (wav2lip_amd/losses.py); the optimiser is one fused HIP launch (wav2lip_amd/optim.py).  torch's autograd
engine only chains those nodes together (plus the view/cat plumbing of get_sync_loss).

Multi-GPU (BASELINE configs 4/5): one process per GPU, each with its own batch shard; BatchNorm statistics stay local to
a rank (what nn.DataParallel gave the authors); gradients are averaged with bucketed all-reduces
(wav2lip_amd/sharding.py:allreduce_gradients) between backward() and step().
"""
import numpy as np
import torch

from . import losses
from .hparams import hparams

syncnet_T = 5
syncnet_mel_step_size = 16


# ---------------------------------------------------------------- Dataset index math (wav2lip_train.py:48-106)
def get_frame_id(frame):
    """wav2lip_train.py:48-49: '<dir>/<id>.jpg' -> id"""
    import os
    return int(os.path.basename(frame).split('.')[0])


def window_frame_ids(start_id):
    """wav2lip_train.py:51-61: the T consecutive frame ids of a window"""
    return list(range(start_id, start_id + syncnet_T))


def audio_window_start(start_frame_num, fps=None):
    """wav2lip_train.py:80 / color_syncnet_train.py:79: int(80. * (frame / float(fps))) — double divide, double multiply,
    truncation"""
    fps = hparams.fps if fps is None else fps
    return int(80. * (start_frame_num / float(fps)))


def crop_audio_window(spec, start_frame_num, fps=None):
    """spec: mel transposed to (T, 80) as the Dataset holds it (wav2lip_train.py:75-84)"""
    start_idx = audio_window_start(start_frame_num, fps)
    return spec[start_idx:start_idx + syncnet_mel_step_size, :]


def get_segmented_mels(spec, start_frame_id, fps=None):
    """wav2lip_train.py:86-99: five (80,16) windows at frames id-1 .. id+3, or None when the clip is too short"""
    mels = []
    start_frame_num = start_frame_id + 1
    if start_frame_num - 2 < 0:
        return None
    for i in range(start_frame_num, start_frame_num + syncnet_T):
        m = crop_audio_window(spec, i - 2, fps)
        if m.shape[0] != syncnet_mel_step_size:
            return None
        mels.append(m.T)
    return np.asarray(mels)


def prepare_window(window):
    """wav2lip_train.py:101-106: list of T HxWx3 uint8 frames -> float64 (3, T, H, W) in [0,1]"""
    x = np.asarray(window) / 255.
    return np.transpose(x, (3, 0, 1, 2))


def make_generator_sample(window, wrong_window, orig_mel_T, frame_id, fps=None):
    """wav2lip_train.py:142-164 after the file reads: returns (x (6,T,H,W), indiv_mels (T,1,80,16), mel (1,80,16),
    y (3,T,H,W)) as float32 torch tensors, or None where the reference `continue`s"""
    mel = crop_audio_window(orig_mel_T.copy(), frame_id, fps)
    if mel.shape[0] != syncnet_mel_step_size:
        return None
    indiv_mels = get_segmented_mels(orig_mel_T.copy(), frame_id, fps)
    if indiv_mels is None:
        return None
    window = prepare_window(window)
    y = window.copy()
    window[:, :, window.shape[2] // 2:] = 0.
    wrong_window = prepare_window(wrong_window)
    x = np.concatenate([window, wrong_window], axis=0)
    return (torch.FloatTensor(x), torch.FloatTensor(indiv_mels).unsqueeze(1), torch.FloatTensor(mel.T).unsqueeze(0),
            torch.FloatTensor(y))


def make_syncnet_sample(window, orig_mel_T, frame_id, fps=None):
    """color_syncnet_train.py:111-131: lower halves of T frames stacked on channels (t-major) + one mel window"""
    x = np.concatenate([np.asarray(f) / 255. for f in window], axis=2)   # H x W x 3T
    x = x.transpose(2, 0, 1)
    x = x[:, x.shape[1] // 2:]
    mel = crop_audio_window(orig_mel_T.copy(), frame_id, fps)
    if mel.shape[0] != syncnet_mel_step_size:
        return None
    return torch.FloatTensor(x), torch.FloatTensor(mel.T).unsqueeze(0)


# ---------------------------------------------------------------- the three step bodies
def _sync_grads(params, dist):
    """`dist` = torch.distributed: average the gradients after backward (sharding.allreduce_gradients).  With a
    sharding.GradReducer attached to the module the averaging already happened INSIDE backward (overlapped with it): pass
    dist=None then."""
    if dist is not None:
        from .sharding import allreduce_gradients
        allreduce_gradients(dist, params)


# hq step: the generator's perceptual loss runs through the discriminator WITHOUT producing the discriminator's own parameter
# gradients (see _perceptual_loss); W2L_PERCEPTUAL_DISC_GRADS=1 computes them as the reference does (A/B switch)
PERCEPTUAL_DISC_GRADS = [__import__("os").environ.get("W2L_PERCEPTUAL_DISC_GRADS", "0") == "1"]


DISC_ONE_PASS = [__import__("os").environ.get("W2L_DISC_ONE_PASS", "1") != "0"]


def _perceptual_loss(disc, g):
    """hq_wav2lip_train.py:233 `disc.perceptual_forward(g)` inside the generator's loss.  In the reference its backward also
    accumulates gradients into the DISCRIMINATOR's parameters, which nothing ever reads: `disc_optimizer.zero_grad()`
    (hq_wav2lip_train.py:245) clears them before the discriminator's own backward passes.  Here the discriminator's parameters are
    marked as not requiring a gradient for the duration of this one forward call, so the backward pass computes the gradient
    with respect to the generated frames only (the data gradients) and skips the discriminator's weight gradients.  Every
    parameter, optimiser state and loss after the step is what the reference's step produces; the one observable difference is
    that `p.grad` of the discriminator's parameters stays None between `loss.backward()` and `disc_optimizer.zero_grad()`."""
    if PERCEPTUAL_DISC_GRADS[0]:
        return disc.perceptual_forward(g)
    ps = [p for p in disc.parameters() if p.requires_grad]
    for p in ps:
        p.requires_grad_(False)
    try:
        return disc.perceptual_forward(g)
    finally:
        for p in ps:
            p.requires_grad_(True)


def syncnet_train_step(model, optimizer, x, mel, y, dist=None):
    """color_syncnet_train.py:149-165"""
    model.train()
    optimizer.zero_grad()
    a, v = model(mel, x)
    loss = losses.cosine_loss(a, v, y)
    loss.backward()
    _sync_grads([p for p in model.parameters() if p.requires_grad], dist)
    optimizer.step()
    return loss


def wav2lip_train_step(model, syncnet, optimizer, x, indiv_mels, mel, gt, syncnet_wt=None, dist=None, return_generated=False):
    """wav2lip_train.py:210-230; returns (loss, l1loss, sync_loss) [+ the generated window `g` when asked: the training loop's
    sample images, wav2lip_train.py:233-234]"""
    syncnet_wt = hparams.syncnet_wt if syncnet_wt is None else syncnet_wt
    model.train()
    optimizer.zero_grad()
    g = model(indiv_mels, x)
    sync_loss = losses.get_sync_loss(syncnet, mel, g) if syncnet_wt > 0. else 0.
    l1loss = losses.l1_loss(g, gt)
    loss = syncnet_wt * sync_loss + (1 - syncnet_wt) * l1loss
    loss.backward()
    _sync_grads([p for p in model.parameters() if p.requires_grad], dist)
    optimizer.step()
    if return_generated:
        return loss, l1loss, sync_loss, g.detach()
    return loss, l1loss, sync_loss


def hq_train_step(model, disc, syncnet, optimizer, disc_optimizer, x, indiv_mels, mel, gt, syncnet_wt=None, disc_wt=None,
                  dist=None, return_generated=False, gather_frames=None):
    """hq_wav2lip_train.py:212-257; returns dict of the five scalar losses (+ "g", the generated window, when asked).

    `gather_frames=torch.distributed` (BASELINE configs[4]: "frames all-gathered over xGMI", SURVEY.md 8e - optional): the
    discriminator's real / fake batches are the GLOBAL batch - every rank all-gathers `gt` and the detached `g` (one collective
    each; no gradient crosses it, the fake batch is detached as in the reference) and trains the discriminator on all of them,
    so that its BCE is the mean over world x B x T frames as it would be under nn.DataParallel on one process.  The generator
    side (perceptual loss through D) stays on the local shard."""
    syncnet_wt = hparams.syncnet_wt if syncnet_wt is None else syncnet_wt
    disc_wt = hparams.disc_wt if disc_wt is None else disc_wt
    disc.train()
    model.train()
    optimizer.zero_grad()
    disc_optimizer.zero_grad()
    g = model(indiv_mels, x)
    sync_loss = losses.get_sync_loss(syncnet, mel, g) if syncnet_wt > 0. else 0.
    perceptual_loss = _perceptual_loss(disc, g) if disc_wt > 0. else 0.
    l1loss = losses.l1_loss(g, gt)
    loss = syncnet_wt * sync_loss + disc_wt * perceptual_loss + (1. - syncnet_wt - disc_wt) * l1loss
    loss.backward()
    _sync_grads([p for p in model.parameters() if p.requires_grad], dist)
    optimizer.step()

    disc_optimizer.zero_grad()
    real, fake = gt, g.detach()
    if gather_frames is not None and gather_frames.get_world_size() > 1:
        from .sharding import all_gather_batch
        real, fake = all_gather_batch(gather_frames, real), all_gather_batch(gather_frames, fake)
    if DISC_ONE_PASS[0]:
        # hq_wav2lip_train.py:247-254 runs D(real) and D(fake) as two forward / backward passes whose parameter gradients ADD.  The
        # discriminator has no BatchNorm, so one pass over [real; fake] with the two targets is the same computation - the same two
        # losses, the sum of the same gradients (in another fp32 summation order) - at half the launches and twice the rows per
        # launch.  W2L_DISC_ONE_PASS=0 runs the two passes (A/B switch).
        # The discriminator stacks the T frames of a window along the batch axis, time-major (models/wav2lip.py:155-161 `to_2d`):
        # the rows of D([real; fake]) are ordered (t, real | fake, sample).
        nr, nf, T = len(real), len(fake), real.shape[2]
        pred = disc(torch.cat([real, fake], dim=0)).view(T, nr + nf, 1)
        pred_real = pred[:, :nr].reshape(-1, 1)
        pred_fake = pred[:, nr:].reshape(-1, 1)
        disc_real_loss = losses.bce_mean(pred_real, torch.ones((len(pred_real), 1), device=pred.device))
        disc_fake_loss = losses.bce_mean(pred_fake, torch.zeros((len(pred_fake), 1), device=pred.device))
        (disc_real_loss + disc_fake_loss).backward()
    else:
        pred = disc(real)
        disc_real_loss = losses.bce_mean(pred, torch.ones((len(pred), 1), device=pred.device))
        disc_real_loss.backward()
        pred = disc(fake)
        disc_fake_loss = losses.bce_mean(pred, torch.zeros((len(pred), 1), device=pred.device))
        disc_fake_loss.backward()
    _sync_grads([p for p in disc.parameters() if p.requires_grad], dist)
    disc_optimizer.step()
    out = dict(loss=loss, l1=l1loss, sync=sync_loss, perceptual=perceptual_loss, disc_real=disc_real_loss,
               disc_fake=disc_fake_loss)
    if return_generated:
        out["g"] = g.detach()
    return out


# ---------------------------------------------------------------- device-resident mel bank
class MelBank:
    """The reference's Dataset recomputes the WHOLE clip's mel spectrogram in every `__getitem__`
    (wav2lip_train.py:138-141, color_syncnet_train.py:117-120: 16 worker processes redoing the STFT per sample).  Here every
    clip's spectrogram is computed once by the HIP mel kernel and kept in HBM as one [80, sum T] bank; a batch of windows
    is one gather launch (w2l_mel_gather) over host-computed start columns (the reference's index expression, bit-exact)."""

    def __init__(self, device):
        from . import audio
        self._audio = audio
        self.device = torch.device(device)
        self._parts, self.offsets, self.lengths = [], [], []
        self.bank = None

    def add(self, wav):
        """wav: float32 samples of one clip (audio.load_wav output) -> clip id"""
        mel = self._audio.melspectrogram_device(wav, self.device)
        self.offsets.append(sum(self.lengths))
        self.lengths.append(int(mel.shape[1]))
        self._parts.append(mel)
        self.bank = None
        return len(self.lengths) - 1

    def _bank(self):
        if self.bank is None:
            self.bank = torch.cat(self._parts, dim=1).contiguous()
        return self.bank

    def window_start(self, clip, frame_num, fps=None):
        """absolute start column of the 16-frame window of `frame_num` in clip `clip`, or None where the reference rejects
        the sample (window runs past the clip: wav2lip_train.py:147-148)"""
        s = audio_window_start(frame_num, fps)
        if s < 0 or s + syncnet_mel_step_size > self.lengths[clip]:
            return None
        return self.offsets[clip] + s

    def _gather(self, starts):
        from ._lib import check, current_stream, load, ptr
        bank = self._bank()
        st = torch.tensor(starts, dtype=torch.int32, device=self.device)
        out = torch.empty((len(starts), 1, 80, syncnet_mel_step_size), device=self.device, dtype=torch.float32)
        check(load().w2l_mel_gather(current_stream(), ptr(bank), bank.shape[1], ptr(st), len(starts), ptr(out), 1, 1),
              "mel_gather")
        return out

    def windows(self, clips, frame_ids, fps=None):
        """`mel` of a batch: [B,1,80,16] (wav2lip_train.py:145,162); raises where the reference would `continue`"""
        starts = [self.window_start(c, f, fps) for c, f in zip(clips, frame_ids)]
        if any(s is None for s in starts):
            raise ValueError("a mel window runs past its clip (the reference's Dataset resamples such items)")
        return self._gather(starts)

    def segmented(self, clips, frame_ids, fps=None):
        """`indiv_mels` of a batch: [B,5,1,80,16], windows at frames id-1 .. id+3 (wav2lip_train.py:86-99,150,163)"""
        starts = []
        for c, f in zip(clips, frame_ids):
            if f + 1 - 2 < 0:
                raise ValueError("frame id %d has no preceding frame (wav2lip_train.py:90)" % f)
            for i in range(f + 1, f + 1 + syncnet_T):
                s = self.window_start(c, i - 2, fps)
                if s is None:
                    raise ValueError("a segmented mel window runs past its clip (wav2lip_train.py:93-94)")
                starts.append(s)
        return self._gather(starts).view(len(frame_ids), syncnet_T, 1, 80, syncnet_mel_step_size)
