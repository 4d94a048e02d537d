"""Device-resident training set: what the reference's `Dataset.__getitem__` + 16 DataLoader workers do per
sample — glob a clip directory, `cv2.imread` + `cv2.resize` ten frames, recompute the WHOLE clip's mel spectrogram, cut windows
(wav2lip_train.py:108-165, color_syncnet_train.py:68-131) — done once per clip instead: every frame is decoded and resized to
96x96 once (`w2l_crop_resize_u8`, the cv2.resize-exact kernel) and stays in HBM as uint8 (27 kB per frame: 100 000 frames are
2.7 GB of the 288), every clip's spectrogram is computed once (`train.MelBank`), and a batch is a few gathers:

    frames u8 [F,96,96,3] --index_select--> windows --w2l_datagen_pack--> /255, lower half masked --> x, y
    mel bank [80, sum T] --w2l_mel_gather--> mel, indiv_mels

The SAMPLING is the reference's: random clip, random frame and "wrong" frame, the same rejection rules in the same order
(clip with <= 3*T frames, incomplete 5-frame windows, audio windows that run past the clip, frame 0 for the segmented mels), drawn
from Python's `random` like the reference.  `picks` records what was drawn so that tests can rebuild the same samples through the
host path (`train.make_generator_sample`).

Image files are decoded with PIL (cv2 is not in the image); JPEG decoding is a libjpeg matter on both sides and is not pinned.
"""
import os
import random

import numpy as np
import torch

from . import _lib, train
from ._lib import check, current_stream, ptr
from .hparams import hparams

T = train.syncnet_T
IMG = 96


class ClipStore:
    def __init__(self, device):
        if not torch.cuda.is_available() or "cuda" not in str(device):
            raise RuntimeError("wav2lip_amd.data.ClipStore needs a HIP device (no CPU path)")
        self.device = torch.device(device)
        self.lib = _lib.load()
        self.mels = train.MelBank(self.device)
        self.names = []
        self.frame_ids = []          # per clip: the ids of its <id>.jpg frames, in the order they were added
        self.first = []              # per clip: row of its first frame in the store
        self.row_of = []             # per clip: {frame id: row}
        self._parts = []
        self._frames = None
        self.n_rows = 0

    # ---------------------------------------------------------------- building
    def _resize(self, frame):
        """`cv2.resize(img, (96, 96))` of wav2lip_train.py:68 / color_syncnet_train.py:101 on the device"""
        f = torch.from_numpy(np.ascontiguousarray(frame)).to(self.device).unsqueeze(0)
        h, w = int(f.shape[1]), int(f.shape[2])
        idx = torch.zeros(1, dtype=torch.int32, device=self.device)
        box = torch.tensor([[0, h, 0, w]], dtype=torch.int32, device=self.device)       # (y1, y2, x1, x2): the whole image
        out = torch.empty((1, IMG, IMG, 3), dtype=torch.uint8, device=self.device)
        check(self.lib.w2l_crop_resize_u8(current_stream(), 1, ptr(f), h, w, ptr(idx), ptr(box), IMG, ptr(out)), "crop_resize_u8")
        return out

    def add_clip(self, frames, frame_ids, wav, name=None):
        """frames: uint8 BGR images [h, w, 3] (any sizes), frame_ids: the integer of each `<id>.jpg`, wav: float32 16 kHz samples
        of the clip's audio.wav.  Returns the clip index."""
        if len(frames) != len(frame_ids) or not len(frames):
            raise ValueError("add_clip: need one frame id per frame")
        rows = []
        for f in frames:
            f = np.asarray(f)
            if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
                raise ValueError("frames must be uint8 [h, w, 3] (BGR), got %s %s" % (f.dtype, f.shape))
            rows.append(self._resize(f) if f.shape[:2] != (IMG, IMG) else torch.from_numpy(np.ascontiguousarray(f)).to(self.device)[None])
        self._parts.append(torch.cat(rows, dim=0))
        self._frames = None
        self.first.append(self.n_rows)
        ids = [int(i) for i in frame_ids]
        self.row_of.append({fid: self.n_rows + k for k, fid in enumerate(ids)})
        self.frame_ids.append(ids)
        self.n_rows += len(ids)
        self.mels.add(np.asarray(wav, dtype=np.float32))
        self.names.append(name if name is not None else "clip%d" % len(self.names))
        return len(self.names) - 1

    @classmethod
    def from_directory(cls, data_root, split, device, filelist_dir="filelists"):
        """the on-disk layout the reference trains from: `filelists/<split>.txt` names clip directories under `data_root`, each
        holding `<id>.jpg` face crops and `audio.wav` (hparams.get_image_list, wav2lip_train.py:42,119,137)"""
        from PIL import Image
        from . import audio
        store = cls(device)
        with open(os.path.join(filelist_dir, "%s.txt" % split)) as fh:
            clips = [line.split()[0] for line in (l.strip() for l in fh) if line]
        for rel in clips:
            d = os.path.join(data_root, rel)
            names = sorted((n for n in os.listdir(d) if n.endswith(".jpg")), key=lambda n: int(n.split(".")[0]))
            frames = [np.asarray(Image.open(os.path.join(d, n)).convert("RGB"))[:, :, ::-1] for n in names]   # cv2.imread order: BGR
            wav = audio.load_wav(os.path.join(d, "audio.wav"), hparams.sample_rate)
            store.add_clip(frames, [int(n.split(".")[0]) for n in names], wav, name=rel)
        return store

    def frames(self):
        if self._frames is None:
            self._frames = torch.cat(self._parts, dim=0).contiguous()
            self._parts = [self._frames]
        return self._frames

    def __len__(self):
        return len(self.names)

    # ---------------------------------------------------------------- the reference's sampling loop
    def _window_rows(self, clip, start_id):
        """rows of frames start_id .. start_id+T-1, or None when one is missing (get_window, wav2lip_train.py:51-61)"""
        rows = [self.row_of[clip].get(start_id + t) for t in range(T)]
        return None if any(r is None for r in rows) else rows

    def _draw(self, rng):
        """one pass of the `while 1:` body up to the window checks: (clip, img id, wrong id) or None where the reference `continue`s"""
        clip = rng.randint(0, len(self.names) - 1)
        ids = self.frame_ids[clip]
        if len(ids) <= 3 * T:
            return None
        img = rng.choice(ids)
        wrong = rng.choice(ids)
        while wrong == img:
            wrong = rng.choice(ids)
        return clip, img, wrong

    def _segmented_starts(self, clip, img, fps):
        """bank columns of the five windows at frames img-1 .. img+3, or None (get_segmented_mels, wav2lip_train.py:86-99)"""
        if img + 1 - 2 < 0:
            return None
        starts = [self.mels.window_start(clip, i - 2, fps) for i in range(img + 1, img + 1 + T)]
        return None if any(s is None for s in starts) else starts

    def _pack(self, rows):
        """uint8 frames of `rows` -> fp32 [n, 96, 96, 8]: channels 0-2 the frame / 255 with rows 48.. zeroed, 3-5 the frame / 255
        (the arithmetic of prepare_window + the masking of wav2lip_train.py:153-158, by the inference path's pack kernel)"""
        idx = torch.tensor(rows, dtype=torch.int64, device=self.device)
        faces = self.frames().index_select(0, idx).contiguous()
        out = torch.empty((len(rows), IMG, IMG, 8), dtype=torch.float32, device=self.device)
        check(self.lib.w2l_datagen_pack(current_stream(), len(rows), IMG, ptr(faces), ptr(out), 8, 8), "datagen_pack")
        return out

    def _generator_rows(self, clip, img, wrong, fps):
        """(window rows, wrong-window rows, mel start, segmented starts) of one pick, or None where the reference `continue`s
        (wav2lip_train.py:124-151, in its order)"""
        wr, wwr = self._window_rows(clip, img), self._window_rows(clip, wrong)
        if wr is None or wwr is None:
            return None
        ms = self.mels.window_start(clip, img, fps)
        if ms is None:
            return None
        ss = self._segmented_starts(clip, img, fps)
        if ss is None:
            return None
        return wr, wwr, ms, ss

    def generator_batch(self, picks, fps=None):
        """the batch of explicit picks [(clip, img id, wrong id)]: x [B,6,T,96,96], indiv_mels [B,T,1,80,16], mel [B,1,80,16],
        y [B,3,T,96,96] (device, fp32) - what wav2lip_train.py's Dataset returns for those choices; raises where it would
        have rejected one"""
        B = len(picks)
        win_rows, wrong_rows, mel_starts, seg_starts = [], [], [], []
        for clip, img, wrong in picks:
            r = self._generator_rows(clip, img, wrong, fps)
            if r is None:
                raise ValueError("pick (clip %d, frame %d, wrong %d) is one the reference's Dataset rejects" % (clip, img, wrong))
            win_rows += r[0]
            wrong_rows += r[1]
            mel_starts.append(r[2])
            seg_starts += r[3]
        win = self._pack(win_rows).view(B, T, IMG, IMG, 8)
        wrong = self._pack(wrong_rows).view(B, T, IMG, IMG, 8)
        x = torch.cat([win[..., 0:3], wrong[..., 3:6]], dim=-1).permute(0, 4, 1, 2, 3).contiguous()
        y = win[..., 3:6].permute(0, 4, 1, 2, 3).contiguous()
        mel = self.mels._gather(mel_starts)
        indiv = self.mels._gather(seg_starts).view(B, T, 1, 80, train.syncnet_mel_step_size)
        return x, indiv, mel, y

    def sample_generator_batch(self, B, rng=random, fps=None):
        """B samples of wav2lip_train.py's Dataset: x [B,6,T,96,96], indiv_mels [B,T,1,80,16], mel [B,1,80,16], y [B,3,T,96,96]
        (device, fp32) and the picks [(clip, img id, wrong id)]"""
        picks = []
        while len(picks) < B:
            d = self._draw(rng)
            if d is None or self._generator_rows(*d, fps) is None:
                continue
            picks.append(d)
        return self.generator_batch(picks, fps) + (picks,)

    def syncnet_batch(self, picks, fps=None):
        """the batch of explicit picks [(clip, img id, wrong id, in_sync)]: x [B,15,48,96] (lower halves, frames stacked t-major
        on channels), mel [B,1,80,16] (always the TRUE frame's audio), y [B,1] - color_syncnet_train.py's Dataset"""
        B = len(picks)
        rows, mel_starts, labels = [], [], []
        for clip, img, wrong, in_sync in picks:
            wr = self._window_rows(clip, img if in_sync else wrong)
            ms = self.mels.window_start(clip, img, fps)
            if wr is None or ms is None:
                raise ValueError("pick (clip %d, frame %d, wrong %d) is one the reference's Dataset rejects" % (clip, img, wrong))
            rows += wr
            mel_starts.append(ms)
            labels.append(1.0 if in_sync else 0.0)
        full = self._pack(rows).view(B, T, IMG, IMG, 8)[..., 3:6]                 # [B,T,96,96,3] / 255
        x = full[:, :, IMG // 2:].permute(0, 1, 4, 2, 3).reshape(B, 3 * T, IMG // 2, IMG).contiguous()
        mel = self.mels._gather(mel_starts)
        y = torch.tensor(labels, dtype=torch.float32, device=self.device).view(B, 1)
        return x, mel, y

    def sample_syncnet_batch(self, B, rng=random, fps=None):
        """B samples of color_syncnet_train.py's Dataset and the picks [(clip, img id, wrong id, in_sync)]"""
        picks = []
        while len(picks) < B:
            d = self._draw(rng)
            if d is None:
                continue
            clip, img, wrong = d
            in_sync = rng.choice([True, False])
            if self._window_rows(clip, img if in_sync else wrong) is None or self.mels.window_start(clip, img, fps) is None:
                continue
            picks.append((clip, img, wrong, in_sync))
        return self.syncnet_batch(picks, fps) + (picks,)
