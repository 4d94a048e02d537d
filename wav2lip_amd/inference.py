"""The per-batch inference loop of the reference (inference.py:108-154 `datagen`, :231-240 mel chunking,
:249-272 forward + uint8 frames) on the HIP path, with everything between "uint8 face crops + mel" and
"uint8 generated crops" resident on the device:

    faces u8 [B,96,96,3] --w2l_datagen_pack--> x fp32 NHWC8 --+
    mel [80,T] + starts --w2l_mel_gather----> m fp32 NHWC4 ---+--> generator plan --> w2l_frames_to_u8 --> u8 [B,96,96,3]

Out of scope here: video decode, face detection, cv2.resize of non-96x96 crops, paste-back and the
ffmpeg mux; `--box`-style pre-cropped 96x96 faces are the supported input.
"""
import numpy as np
import torch

from . import _lib, audio
from ._lib import check, current_stream, ptr

mel_step_size = 16   # inference.py:156
img_size = 96


def mel_chunk_starts(n_mel_frames, fps):
    """inference.py:231-240: start column of each 16-frame window; the tail window is re-anchored at the end.
    Pure host integer arithmetic (double multiply + truncation), bit-exact by construction."""
    mel_idx_multiplier = 80. / fps
    starts = []
    i = 0
    while True:
        start_idx = int(i * mel_idx_multiplier)
        if start_idx + mel_step_size > n_mel_frames:
            starts.append(n_mel_frames - mel_step_size)
            return starts
        starts.append(start_idx)
        i += 1


class Wav2LipRunner:
    """Device-resident replacement of the body of the reference's batch loop for one model and batch size."""

    def __init__(self, model, batch_size=128):
        self.model = model
        self.batch_size = batch_size
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("wav2lip_amd.inference: the model must be on a HIP device (no CPU path)")
        self.lib = _lib.load()
        self._out_u8 = {}

    def _graph(self, n):
        return self.model.graph(n, img_size, img_size, self.device)

    def run_batch(self, faces_u8, mel_windows=None, mel=None, starts=None):
        """faces_u8: torch uint8 [n,96,96,3] on the device.  Audio either as ready windows `mel_windows`
        float32 [n,80,16], or as the full spectrogram `mel` [80,T] plus int32 `starts` [n] (device tensors).
        Returns torch uint8 [n,96,96,3] (BGR order preserved), valid until the next call."""
        n = faces_u8.shape[0]
        g = self._graph(n)
        s = current_stream()
        faces_u8 = faces_u8.contiguous()
        check(self.lib.w2l_datagen_pack(s, n, img_size, ptr(faces_u8), ptr(g.x_in), 8, 8), "datagen_pack")
        if mel_windows is not None:
            mw = mel_windows.contiguous().float().view(n, 1, 80, 16)
            check(self.lib.w2l_nchw_to_nhwc(s, n, 1, 80, 16, ptr(mw), ptr(g.mel_in), 4, 4), "nchw_to_nhwc")
        else:
            check(self.lib.w2l_mel_gather(s, ptr(mel), mel.shape[1], ptr(starts), n, ptr(g.mel_in), 4, 4),
                  "mel_gather")
        g.run()
        out = self._out_u8.get(n)
        if out is None:
            out = torch.empty((n, img_size, img_size, 3), dtype=torch.uint8, device=self.device)
            self._out_u8[n] = out
        check(self.lib.w2l_frames_to_u8(s, n, img_size, img_size, g.out.ptr, g.out.cs, ptr(out)), "frames_to_u8")
        self._last = g
        return out

    def last_pred_nchw(self):
        return self._last.output_nchw()


def datagen(frames, mels, batch_size=128, static=False, box=None):
    """inference.py:108-154 for pre-cropped faces: yields (faces_u8 [b,96,96,3], mel_windows [b,80,16],
    frame_batch, coords_batch).  `frames` are HxWx3 uint8 images; with `box=(y1,y2,x1,x2)` the face is the box
    crop, which must already be 96x96 (cv2.resize of other sizes is outside the hot path)."""
    img_batch, mel_batch, frame_batch, coords_batch = [], [], [], []
    for i, m in enumerate(mels):
        idx = 0 if static else i % len(frames)
        frame = frames[idx]
        if box is not None:
            y1, y2, x1, x2 = box
        else:
            y1, y2, x1, x2 = 0, frame.shape[0], 0, frame.shape[1]
        face = frame[y1:y2, x1:x2]
        if face.shape[:2] != (img_size, img_size):
            raise NotImplementedError("face crop is %s; only %dx%d crops are supported" %
                                      (face.shape[:2], img_size, img_size))
        img_batch.append(face)
        mel_batch.append(m)
        frame_batch.append(frame.copy())
        coords_batch.append((y1, y2, x1, x2))
        if len(img_batch) >= batch_size:
            yield np.asarray(img_batch), np.asarray(mel_batch), frame_batch, coords_batch
            img_batch, mel_batch, frame_batch, coords_batch = [], [], [], []
    if img_batch:
        yield np.asarray(img_batch), np.asarray(mel_batch), frame_batch, coords_batch


def lipsync(model, frames, wav, fps=25., batch_size=128, static=False, box=None):
    """End-to-end body of inference.py:main for in-memory inputs: returns the list of output frames (uint8)."""
    dev = next(model.parameters()).device
    mel = audio.melspectrogram_device(wav, dev)
    if bool(torch.isnan(mel).any()):
        raise ValueError("Mel contains nan! Using a TTS voice? Add a small epsilon noise to the wav file and try again")
    starts = mel_chunk_starts(mel.shape[1], fps)
    frames = frames[:len(starts)] if not static else frames
    runner = Wav2LipRunner(model, batch_size)
    out_frames = []
    pos = 0
    starts_dev = torch.tensor(starts, dtype=torch.int32, device=dev)
    for faces, _, frame_batch, coords in datagen(frames, starts, batch_size, static, box):
        n = len(faces)
        u8 = runner.run_batch(torch.from_numpy(faces).to(dev), mel=mel, starts=starts_dev[pos:pos + n].contiguous())
        pos += n
        u8 = u8.cpu().numpy()
        for p, f, c in zip(u8, frame_batch, coords):
            y1, y2, x1, x2 = c
            f[y1:y2, x1:x2] = p
            out_frames.append(f)
    return out_frames
