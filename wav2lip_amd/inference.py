"""The per-batch inference loop of the reference (inference.py:108-154 `datagen`, :231-240 mel chunking,
:249-272 forward + uint8 frames) on the HIP path, with everything between "uint8 face crops + mel" and
"uint8 generated crops" resident on the device:

    faces u8 [B,96,96,3] --w2l_datagen_pack--> x fp32 NHWC8 --+
    mel [80,T] + starts --w2l_mel_gather----> m fp32 NHWC4 ---+--> generator plan --> w2l_frames_to_u8 --> u8 [B,96,96,3]

Either side of that, also on the device: the face crop + `cv2.resize(face, (96, 96))` of
inference.py:121-126 (w2l_crop_resize_u8) and the `cv2.resize` to the box size + paste-back of :270-271
(w2l_resize_paste_u8), so that full uint8 frames go in and full uint8 frames come out (`Wav2LipRunner.run_frames`).
The output side of the loop (`cv2.VideoWriter` + the ffmpeg mux, inference.py:256-257,272-277) is `write_result` below, on top
of wav2lip_amd/container.py (uncompressed AVI with the PCM16 audio interleaved).  Out of scope: decoding compressed video
(`cv2.VideoCapture` of mp4 input), audio extraction from non-WAV containers (the reference's first ffmpeg call).
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

    def __init__(self, model, batch_size=128, lane=0):
        self.model = model
        self.batch_size = batch_size
        self.lane = lane
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("wav2lip_amd.inference: the model must be on a HIP device (no CPU path)")
        self.lib = _lib.load()
        self._out_u8 = {}

    def _graph(self, n):
        return self.model.graph(n, img_size, img_size, self.device, lane=self.lane)

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

    def run_frames(self, frames, frame_idx, boxes, mel_windows=None, mel=None, starts=None):
        """Full-frame variant of `run_batch` (inference.py:121-126 + :259-271).  frames: torch uint8 [F,H,W,3] on the
        device; frame_idx: n frame numbers; boxes: n (y1, y2, x1, x2) face boxes.  Crops and resizes the faces to 96x96,
        runs the batch, resizes each generated crop to its box and pastes it into a copy of its frame.
        Returns torch uint8 [n,H,W,3] (a fresh tensor)."""
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3 or not frames.is_cuda:
            raise RuntimeError("run_frames: frames must be a uint8 [F,H,W,3] tensor on the HIP device")
        frames = frames.contiguous()
        F_, H, W = frames.shape[:3]
        boxes = validate_boxes(boxes, H, W)
        n = len(boxes)
        idx = [int(i) for i in frame_idx]
        if len(idx) != n or any(i < 0 or i >= F_ for i in idx):
            raise ValueError("run_frames: frame_idx must hold one valid frame number per box")
        s = current_stream()
        boxes_dev = torch.tensor(boxes, dtype=torch.int32, device=self.device)
        idx_dev = torch.tensor(idx, dtype=torch.int32, device=self.device)
        faces = torch.empty((n, img_size, img_size, 3), dtype=torch.uint8, device=self.device)
        check(self.lib.w2l_crop_resize_u8(s, n, ptr(frames), H, W, ptr(idx_dev), ptr(boxes_dev), img_size, ptr(faces)),
              "crop_resize_u8")
        pred = self.run_batch(faces, mel_windows=mel_windows, mel=mel, starts=starts)
        out = frames.index_select(0, idx_dev.long())            # frame_batch copies (inference.py:131)
        max_px = max((y2 - y1) * (x2 - x1) for y1, y2, x1, x2 in boxes)
        check(self.lib.w2l_resize_paste_u8(s, n, ptr(pred), img_size, ptr(boxes_dev), None, ptr(out), H, W, max_px),
              "resize_paste_u8")
        return out


class PipelinedRunner:
    """`depth` Wav2LipRunner lanes, each with its own generator buffers and HIP stream: successive batches alternate between
    the lanes, so the low-occupancy layers of one batch overlap the chip-filling layers of the other (+7 % frames/s at
    depth 2 on MI355X, bench.py).  `submit` enqueues a batch and returns a ticket; `result(ticket)` makes the caller's
    stream wait for that batch and returns its uint8 frames (valid until the lane is reused `depth` submits later)."""

    def __init__(self, model, batch_size=128, depth=2):
        self.lanes = [Wav2LipRunner(model, batch_size, lane=k) for k in range(depth)]
        dev = self.lanes[0].device
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.n = 0

    def submit(self, faces_u8, mel_windows=None, mel=None, starts=None, frames=None, frame_idx=None, boxes=None):
        k = self.n % len(self.lanes)
        self.n += 1
        st = self.streams[k]
        st.wait_stream(torch.cuda.current_stream())       # inputs were produced on the caller's stream
        for t in (faces_u8, mel_windows, mel, starts, frames):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)                        # the allocator must not recycle them while the lane still reads
        with torch.cuda.stream(st):
            if frames is not None:
                out = self.lanes[k].run_frames(frames, frame_idx, boxes, mel_windows=mel_windows, mel=mel, starts=starts)
            else:
                out = self.lanes[k].run_batch(faces_u8, mel_windows=mel_windows, mel=mel, starts=starts)
            done = torch.cuda.Event()
            done.record(st)
        return (out, done)

    @staticmethod
    def result(ticket):
        out, done = ticket
        torch.cuda.current_stream().wait_event(done)
        return out


def validate_boxes(boxes, H, W):
    """(y1, y2, x1, x2) per item, as the reference slices frames (inference.py:87,121): must be non-empty and inside the
    frame — numpy would silently clip an overhanging slice and the later paste would then fail on the shape mismatch"""
    out = []
    for b in boxes:
        y1, y2, x1, x2 = (int(v) for v in b)
        if not (0 <= y1 < y2 <= H and 0 <= x1 < x2 <= W):
            raise ValueError("face box (y1=%d, y2=%d, x1=%d, x2=%d) is empty or outside the %dx%d frame" % (y1, y2, x1, x2, H, W))
        out.append((y1, y2, x1, x2))
    if not out:
        raise ValueError("no face boxes")
    return out


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
    runner = PipelinedRunner(model, batch_size, depth=2)     # batch i+1 is enqueued before batch i is collected
    out_frames = []
    pos = 0
    starts_dev = torch.tensor(starts, dtype=torch.int32, device=dev)
    shapes = {tuple(f.shape) for f in frames}
    if len(shapes) == 1 and box is not None and (box[1] - box[0], box[3] - box[2]) != (img_size, img_size):
        # faces that need resizing: keep the frames on the device, crop/resize/paste there (inference.py:121-126, 270-271)
        frames_dev = torch.from_numpy(np.stack(frames)).to(dev)
        pending = None
        for lo in range(0, len(starts), batch_size):
            n = min(batch_size, len(starts) - lo)
            idx = [0 if static else (lo + j) % len(frames) for j in range(n)]
            ticket = runner.submit(None, mel=mel, starts=starts_dev[lo:lo + n].contiguous(), frames=frames_dev, frame_idx=idx,
                                   boxes=[box] * n)
            if pending is not None:
                out_frames += list(runner.result(pending).cpu().numpy())
            pending = ticket
        if pending is not None:
            out_frames += list(runner.result(pending).cpu().numpy())
        return out_frames
    pending = None

    def collect(item):
        ticket, frame_batch, coords = item
        u8 = runner.result(ticket).cpu().numpy()
        for p, f, c in zip(u8, frame_batch, coords):
            y1, y2, x1, x2 = c
            f[y1:y2, x1:x2] = p
            out_frames.append(f)

    for faces, _, frame_batch, coords in datagen(frames, starts, batch_size, static, box):
        n = len(faces)
        ticket = runner.submit(torch.from_numpy(faces).to(dev), mel=mel, starts=starts_dev[pos:pos + n].contiguous())
        pos += n
        if pending is not None:
            collect(pending)
        pending = (ticket, frame_batch, coords)
    if pending is not None:
        collect(pending)
    return out_frames


def write_result(outfile, frames, fps, audio_path=None):
    """inference.py:256-257 (`cv2.VideoWriter('temp/result.avi', DIVX, fps, (w, h))`), :272 (`out.write(f)`), :274 and the mux of
    :276-277 (`ffmpeg -y -i <audio> -i temp/result.avi <outfile>`) in one step: the generated frames and the driving audio go into
    one AVI (lossless BGR video, PCM16 audio).  `audio_path`: the WAV that drove the lips (any rate; stored as it is)."""
    from . import container
    pcm, sr = None, 16000
    if audio_path is not None:
        from scipy.io import wavfile
        sr, pcm = wavfile.read(audio_path)
        if pcm.dtype != np.int16:
            pcm = np.clip(np.round(audio._pcm_to_float32(pcm) * 32768.0), -32768, 32767).astype(np.int16)
    container.write_avi(outfile, frames, fps, audio=pcm, audio_sr=sr)
    return outfile


# ---------------------------------------------------------------- face detection front end (inference.py:59-104)
def get_smoothened_boxes(boxes, T):
    """inference.py:59-66, in place (on the integer array face_detect builds: means are truncated on assignment)"""
    for i in range(len(boxes)):
        if i + T > len(boxes):
            window = boxes[len(boxes) - T:]
        else:
            window = boxes[i: i + T]
        boxes[i] = np.mean(window, axis=0)
    return boxes


def face_detect(images, detector, pads=(0, 10, 0, 0), nosmooth=False, batch_size=16):
    """inference.py:68-104: S3FD boxes per frame (HIP detector), padding, temporal smoothing; returns
    [[face crop, (y1, y2, x1, x2)], ...].  `detector` is a wav2lip_amd.face_detection.FaceAlignment."""
    while 1:
        predictions = []
        try:
            for i in range(0, len(images), batch_size):
                predictions.extend(detector.get_detections_for_batch(np.array(images[i:i + batch_size])))
        except RuntimeError:
            if batch_size == 1:
                raise RuntimeError('Image too big to run face detection on GPU. Please use the --resize_factor argument')
            batch_size //= 2
            continue
        break
    results = []
    pady1, pady2, padx1, padx2 = pads
    for rect, image in zip(predictions, images):
        if rect is None:
            raise ValueError('Face not detected! Ensure the video contains a face in all the frames.')
        y1 = max(0, rect[1] - pady1)
        y2 = min(image.shape[0], rect[3] + pady2)
        x1 = max(0, rect[0] - padx1)
        x2 = min(image.shape[1], rect[2] + padx2)
        results.append([x1, y1, x2, y2])
    boxes = np.array(results)
    if not nosmooth:
        boxes = get_smoothened_boxes(boxes, T=5)
    return [[image[y1: y2, x1:x2], (y1, y2, x1, x2)] for image, (x1, y1, x2, y2) in zip(images, boxes)]
