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
import argparse
import os

import numpy as np
import torch

from . import _lib, audio
from ._lib import check, current_stream, ptr
from .checkpoint import _load, load_model  # noqa: F401  (inference.py:160-179: same names, same behaviour)

mel_step_size = 16   # inference.py:156
img_size = 96


# ---------------------------------------------------------------- the command-line surface (inference.py:11-57)
def build_parser():
    """the reference's flags, names, types and defaults (inference.py:11-51); `python -m wav2lip_amd.inference ...` takes the
    command line the reference's `python inference.py ...` takes"""
    parser = argparse.ArgumentParser(description='Inference code to lip-sync videos in the wild using Wav2Lip models')
    parser.add_argument('--checkpoint_path', type=str, help='Name of saved checkpoint to load weights from', required=True)
    parser.add_argument('--face', type=str, help='Filepath of video/image that contains faces to use', required=True)
    parser.add_argument('--audio', type=str, help='Filepath of video/audio file to use as raw audio source', required=True)
    parser.add_argument('--outfile', type=str, help='Video path to save result. See default for an e.g.',
                        default='results/result_voice.mp4')
    parser.add_argument('--static', type=bool, help='If True, then use only first video frame for inference', default=False)
    parser.add_argument('--fps', type=float, help='Can be specified only if input is a static image (default: 25)',
                        default=25., required=False)
    parser.add_argument('--pads', nargs='+', type=int, default=[0, 10, 0, 0],
                        help='Padding (top, bottom, left, right). Please adjust to include chin at least')
    parser.add_argument('--face_det_batch_size', type=int, help='Batch size for face detection', default=16)
    parser.add_argument('--wav2lip_batch_size', type=int, help='Batch size for Wav2Lip model(s)', default=128)
    parser.add_argument('--resize_factor', default=1, type=int,
                        help='Reduce the resolution by this factor. Sometimes, best results are obtained at 480p or 720p')
    parser.add_argument('--crop', nargs='+', type=int, default=[0, -1, 0, -1],
                        help='Crop video to a smaller region (top, bottom, left, right). Applied after resize_factor and rotate arg. '
                             'Useful if multiple face present. -1 implies the value will be auto-inferred based on height, width')
    parser.add_argument('--box', nargs='+', type=int, default=[-1, -1, -1, -1],
                        help='Specify a constant bounding box for the face. Use only as a last resort if the face is not detected.'
                             'Also, might work only if the face is not moving around much. Syntax: (top, bottom, left, right).')
    parser.add_argument('--rotate', default=False, action='store_true',
                        help='Sometimes videos taken from a phone can be flipped 90deg. If true, will flip video right by 90deg.'
                             'Use if you get a flipped result, despite feeding a normal looking video')
    parser.add_argument('--nosmooth', default=False, action='store_true',
                        help='Prevent smoothing face detections over a short temporal window')
    return parser


parser = build_parser()


def parse_args(argv=None):
    """inference.py:53-57: parse, then `img_size = 96` and `static = True` for an image input"""
    a = parser.parse_args(argv)
    a.img_size = img_size
    if os.path.isfile(a.face) and a.face.split('.')[1] in ['jpg', 'png', 'jpeg']:
        a.static = True
    return a


# The reference parses sys.argv at import time into a module global that `datagen` / `face_detect` read (inference.py:53).  A
# library cannot do that; `args` holds the defaults until `main()` (or a caller) replaces it.
args = parser.parse_args(['--checkpoint_path', '', '--face', '', '--audio', ''])
args.img_size = img_size
device = 'cuda'      # inference.py:157 (`'cuda' if torch.cuda.is_available() else 'cpu'`): this engine has no CPU path


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

    def run_batch(self, faces_u8, mel_windows=None, mel=None, starts=None, out=None):
        """faces_u8: torch uint8 [n,96,96,3] on the device.  Audio either as ready windows `mel_windows`
        float32 [n,80,16], or as the full spectrogram `mel` [80,T] plus int32 `starts` [n] (device tensors).
        Returns torch uint8 [n,96,96,3] (BGR order preserved), valid until the next call; `out` (a contiguous uint8
        [n,96,96,3] device tensor, e.g. the send slot of the multi-GPU frame gatherer) receives the frames instead."""
        n = faces_u8.shape[0]
        cap = self.model.MAX_PLAN_BATCH
        if n > cap:     # one NHWC buffer must stay below 2 GiB: run equal chunks and concatenate the uint8 frames
            parts = []
            for lo in range(0, n, cap):
                hi = min(n, lo + cap)
                parts.append(self.run_batch(faces_u8[lo:hi], None if mel_windows is None else mel_windows[lo:hi], mel,
                                            None if starts is None else starts[lo:hi].contiguous()).clone())
            allf = torch.cat(parts, dim=0)
            if out is not None:
                out.copy_(allf)
                return out
            return allf
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
        if out is not None:
            if out.dtype != torch.uint8 or tuple(out.shape) != (n, img_size, img_size, 3) or not out.is_cuda or not out.is_contiguous():
                raise RuntimeError("run_batch: `out` must be a contiguous uint8 [%d,%d,%d,3] tensor on the HIP device" % (n, img_size, img_size))
        else:
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
    the lanes, so the low-occupancy layers of one batch overlap the chip-filling layers of the others.  This is the loop
    bench.py times (`--pipeline` = depth; `lipsync` / `main` run LIPSYNC_DEPTH).  `submit` enqueues a batch and returns a
    ticket; `result(ticket)` makes the caller's stream wait for that batch and returns its uint8 frames (valid until the lane is
    reused `depth` submits later)."""

    def __init__(self, model, batch_size=128, depth=2):
        self.lanes = [Wav2LipRunner(model, batch_size, lane=k) for k in range(depth)]
        dev = self.lanes[0].device
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.depth = depth
        self.n = 0

    def submit(self, faces_u8, mel_windows=None, mel=None, starts=None, frames=None, frame_idx=None, boxes=None, out=None):
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
                out = self.lanes[k].run_batch(faces_u8, mel_windows=mel_windows, mel=mel, starts=starts, out=out)
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


def resize_faces_u8(faces, size=img_size):
    """`cv2.resize(face, (size, size))` of inference.py:126 for a list of uint8 BGR crops of ANY sizes, on the device
    (w2l_crop_resize_u8: OpenCV's fixed-point INTER_LINEAR, csrc/resize.hip).  Returns numpy uint8 [n, size, size, 3]."""
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    out = np.empty((len(faces), size, size, 3), dtype=np.uint8)
    by_shape = {}
    for i, f in enumerate(faces):
        f = np.asarray(f)
        if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3 or f.shape[0] < 1 or f.shape[1] < 1:
            raise ValueError("face crops must be non-empty uint8 [h, w, 3] arrays, got %s %s" % (f.dtype, f.shape))
        if f.shape[:2] == (size, size):
            out[i] = f
        else:
            by_shape.setdefault(f.shape[:2], []).append(i)
    for (h, w), idxs in by_shape.items():          # one launch per distinct crop size (a video's boxes share few sizes)
        src = torch.from_numpy(np.stack([np.ascontiguousarray(faces[i]) for i in idxs])).to(dev)
        boxes = torch.tensor([[0, h, 0, w]] * len(idxs), dtype=torch.int32, device=dev)
        dst = torch.empty((len(idxs), size, size, 3), dtype=torch.uint8, device=dev)
        check(lib.w2l_crop_resize_u8(current_stream(), len(idxs), ptr(src), h, w, None, ptr(boxes), size, ptr(dst)),
              "crop_resize_u8")
        out[idxs] = dst.cpu().numpy()
    return out


def datagen(frames, mels):
    """The reference's HOST-format batch generator (inference.py:108-154), written from its contract: called as
    `datagen(full_frames.copy(), mel_chunks)` and driven by the module-level `args` (box / static / img_size / wav2lip_batch_size,
    and through `face_detect` pads / nosmooth / face_det_batch_size), it yields per batch of at most `wav2lip_batch_size` chunks

        img_batch     float64 [b, img_size, img_size, 6]   channels 0-2: the resized face with its lower half zeroed, 3-5: the face; / 255.
        mel_batch     float32 [b, 80, 16, 1]
        frame_batch   list of b frame copies               chunk i pairs with frame i % len(frames) (frame 0 when args.static)
        coords_batch  list of b (y1, y2, x1, x2)

    Faces come from `face_detect` (all frames, or the first one when static) or are the `--box` crop of every frame; the
    `cv2.resize(face, (img_size, img_size))` of every face is the device kernel (w2l_crop_resize_u8).  float64 images cannot come out
    of the fp32 device pack, so this generator is for callers that consume the reference's numpy batches; `main()` / `lipsync()` keep
    uint8 crops on the device (`datagen_u8` + `Wav2LipRunner`), which is the measured path."""
    a = args
    if a.box[0] == -1:
        detections = face_detect(frames if not a.static else [frames[0]])
    else:
        print('Using the specified bounding box instead of face detection...')
        y1, y2, x1, x2 = a.box
        detections = [[f[y1: y2, x1:x2], (y1, y2, x1, x2)] for f in frames]
    size, per_batch = a.img_size, a.wav2lip_batch_size
    for lo in range(0, len(mels), per_batch):
        chunk = mels[lo:lo + per_batch]
        which = [0 if a.static else i % len(frames) for i in range(lo, lo + len(chunk))]
        faces = resize_faces_u8([detections[j][0] for j in which], size)          # uint8 [b, size, size, 3]
        masked = faces.copy()
        masked[:, size // 2:] = 0
        mel_batch = np.asarray(chunk)
        yield (np.concatenate((masked, faces), axis=3) / 255., mel_batch.reshape(mel_batch.shape + (1,)),
               [frames[j].copy() for j in which], [detections[j][1] for j in which])


def datagen_u8(frames, mels, batch_size=128, static=False, box=None, first=0):
    """the device-path batching of inference.py:108-154 for faces that are already 96x96: yields (faces_u8 [b,96,96,3],
    mel items [b], frame_batch, coords_batch); the masking / concat / `/255.` happen in w2l_datagen_pack on the device.
    `frames` are HxWx3 uint8 images; with `box=(y1,y2,x1,x2)` the face is the box crop.  Other crop sizes take the
    frames-on-device route of `lipsync` / `Wav2LipRunner.run_frames` (crop + resize + paste-back on the device).
    `first`: the clip position of mels[0] (a rank's shard of the mel chunks starts there; frame i of the clip pairs with
    frames[i % len(frames)] as in the reference)."""
    img_batch, mel_batch, frame_batch, coords_batch = [], [], [], []
    for i, m in enumerate(mels, start=first):
        idx = 0 if static else i % len(frames)
        frame = frames[idx]
        if box is not None:
            y1, y2, x1, x2 = box
        else:
            y1, y2, x1, x2 = 0, frame.shape[0], 0, frame.shape[1]
        face = frame[y1:y2, x1:x2]
        if face.shape[:2] != (img_size, img_size):
            raise ValueError("datagen_u8 batches pre-sized %dx%d faces (got %s): use lipsync() / Wav2LipRunner.run_frames, which "
                             "crop and resize on the device" % (img_size, img_size, face.shape[:2]))
        img_batch.append(face)
        mel_batch.append(m)
        frame_batch.append(frame.copy())
        coords_batch.append((y1, y2, x1, x2))
        if len(img_batch) >= batch_size:
            yield np.asarray(img_batch), np.asarray(mel_batch), frame_batch, coords_batch
            img_batch, mel_batch, frame_batch, coords_batch = [], [], [], []
    if img_batch:
        yield np.asarray(img_batch), np.asarray(mel_batch), frame_batch, coords_batch


LIPSYNC_DEPTH = 4     # batches in flight in lipsync() / main(): the depth bench.py measures (bench.py --pipeline)


def lipsync(model, frames, wav, fps=25., batch_size=128, static=False, box=None, ranks=None, depth=None):
    """End-to-end body of inference.py:main for in-memory inputs: returns the list of output frames (uint8).

    `ranks` (sharding.init_from_env(), one process per GPU): the mel chunks are cut into contiguous per-rank shards
    (sharding.shard_range), every rank runs ITS chunks through its own runner with the replicated weights, and the generated
    frames are all-gathered in frame order to rank 0 (sharding.gather_shards_to_writer), which alone returns them - every other
    rank returns None (the reference's loop, inference.py:249-272, is single-process)."""
    dev = next(model.parameters()).device
    mel = audio.melspectrogram_device(wav, dev)
    if bool(torch.isnan(mel).any()):
        raise ValueError("Mel contains nan! Using a TTS voice? Add a small epsilon noise to the wav file and try again")
    starts = mel_chunk_starts(mel.shape[1], fps)
    frames = frames[:len(starts)] if not static else frames
    world = 1 if ranks is None or ranks.dist is None else ranks.world
    rank = 0 if world == 1 else ranks.rank
    from . import sharding
    first, last = sharding.shard_range(len(starts), rank, world)
    runner = PipelinedRunner(model, batch_size, depth=depth or LIPSYNC_DEPTH)   # later batches are enqueued before batch i is collected
    out_frames = []
    starts_dev = torch.tensor(starts, dtype=torch.int32, device=dev)
    shapes = {tuple(f.shape) for f in frames}
    if world > 1 and len(shapes) != 1:
        raise ValueError("sharded inference gathers frames of ONE shape (a video); got %d shapes" % len(shapes))
    pending = []
    if len(shapes) == 1 and box is not None and (box[1] - box[0], box[3] - box[2]) != (img_size, img_size):
        # faces that need resizing: keep the frames on the device, crop/resize/paste there (inference.py:121-126, 270-271)
        frames_dev = torch.from_numpy(np.stack(frames)).to(dev)
        for lo in range(first, last, batch_size):
            n = min(batch_size, last - lo)
            idx = [0 if static else (lo + j) % len(frames) for j in range(n)]
            pending.append(runner.submit(None, mel=mel, starts=starts_dev[lo:lo + n].contiguous(), frames=frames_dev, frame_idx=idx,
                                         boxes=[box] * n))
            if len(pending) >= runner.depth:
                out_frames += list(runner.result(pending.pop(0)).cpu().numpy())
        while pending:
            out_frames += list(runner.result(pending.pop(0)).cpu().numpy())
    else:
        def collect(item):
            ticket, frame_batch, coords = item
            u8 = runner.result(ticket).cpu().numpy()
            for p, f, c in zip(u8, frame_batch, coords):
                y1, y2, x1, x2 = c
                f[y1:y2, x1:x2] = p
                out_frames.append(f)

        pos = first
        for faces, _, frame_batch, coords in datagen_u8(frames, starts[first:last], batch_size, static, box, first=first):
            n = len(faces)
            ticket = runner.submit(torch.from_numpy(faces).to(dev), mel=mel, starts=starts_dev[pos:pos + n].contiguous())
            pos += n
            pending.append((ticket, frame_batch, coords))
            if len(pending) >= runner.depth:
                collect(pending.pop(0))
        while pending:
            collect(pending.pop(0))
    if world == 1:
        return out_frames
    local = np.stack(out_frames) if out_frames else None
    allf = sharding.gather_shards_to_writer(ranks.dist, local, len(starts), rank, world, device=dev, chunk=batch_size)
    return list(allf) if allf is not None else None


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
    """inference.py:59-66: box i becomes the mean of boxes i .. i+T-1, the last T-1 boxes the mean of the last T; done in
    place and in order on the integer array face_detect builds, so a window sees the boxes already smoothed before it and
    every mean is truncated on assignment (both are the reference's behaviour, kept)."""
    n = len(boxes)
    for i in range(n):
        window = slice(i, i + T) if i + T <= n else slice(n - T, None)      # n < T: a negative start wraps, as there
        boxes[i] = boxes[window].mean(axis=0)
    return boxes


def _detect_rects(images, detector, batch_size, ranks=None):
    """inference.py:75-88: one rect (or None) per frame; a RuntimeError of the detector (out of device memory on a large
    frame) halves the detection batch and starts over, down to single frames.  With `ranks` every rank detects the frames of its
    contiguous shard and the per-frame rects (four numbers each) are all-gathered: the box smoothing that follows needs them
    all, on every rank, in frame order."""
    if ranks is not None and ranks.dist is not None and ranks.world > 1:
        from . import sharding
        lo, hi = sharding.shard_range(len(images), ranks.rank, ranks.world)
        mine = _detect_rects(images[lo:hi], detector, batch_size) if hi > lo else []
        parts = [None] * ranks.world
        ranks.dist.all_gather_object(parts, mine)
        return [r for part in parts for r in part]
    while True:
        try:
            rects = []
            for lo in range(0, len(images), batch_size):
                rects += detector.get_detections_for_batch(np.array(images[lo:lo + batch_size]))
            return rects
        except RuntimeError:
            if batch_size == 1:
                raise RuntimeError('Image too big to run face detection on GPU. Please use the --resize_factor argument')
            batch_size //= 2
            print('Recovering from OOM error; New batch size: {}'.format(batch_size))


def face_detect(images, detector=None, pads=None, nosmooth=None, batch_size=None, ranks=None):
    """inference.py:68-104: S3FD boxes per frame (HIP detector), padding, temporal smoothing; returns
    [[face crop, (y1, y2, x1, x2)], ...].  Called as the reference calls it - `face_detect(images)` - pads / nosmooth /
    face_det_batch_size come from the module-level `args` and the detector is built as inference.py:69-70 does
    (`face_detection.FaceAlignment(LandmarksType._2D, flip_input=False, device=device)`, weights `face_detection/s3fd.pth`);
    a ready `wav2lip_amd.face_detection.FaceAlignment` may be passed instead."""
    if detector is None:
        from . import face_detection
        detector = face_detection.FaceAlignment(face_detection.LandmarksType._2D, flip_input=False, device=device)
    pads = args.pads if pads is None else pads
    nosmooth = args.nosmooth if nosmooth is None else nosmooth
    batch_size = args.face_det_batch_size if batch_size is None else batch_size
    rects = _detect_rects(images, detector, batch_size, ranks)
    if any(r is None for r in rects):
        raise ValueError('Face not detected! Ensure the video contains a face in all the frames.')
    top, bottom, left, right = pads
    # (x1, y1, x2, y2) grown by the pads; the near edges are clipped at 0 and the far edges at the frame size ONLY (inference.py:91-98:
    # `max(0, rect[1] - pady1)`, `min(image.shape[0], rect[3] + pady2)`, ...) - a negative pad or a rect beyond the frame is not
    # pulled back from the other side, exactly as there
    boxes = np.array([[max(0, r[0] - left), max(0, r[1] - top), min(im.shape[1], r[2] + right), min(im.shape[0], r[3] + bottom)]
                      for r, im in zip(rects, images)])
    if not nosmooth:
        boxes = get_smoothened_boxes(boxes, T=5)
    return [[im[y1:y2, x1:x2], (y1, y2, x1, x2)] for im, (x1, y1, x2, y2) in zip(images, boxes)]


# ---------------------------------------------------------------- main (inference.py:181-277)
def read_image_bgr(path):
    """`cv2.imread(path)`: uint8 [h, w, 3] in BGR order (decoded with PIL; JPEG decoding is a libjpeg matter on both sides)"""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


def resize_frames_u8(frames, wh):
    """`cv2.resize(frame, (w, h))` of whole frames on the device (w2l_resize_u8); frames: list of equal-sized uint8 [H,W,3]"""
    w, h = int(wh[0]), int(wh[1])
    if w < 1 or h < 1:
        raise ValueError("resize target %dx%d is empty" % (w, h))
    dev = torch.device("cuda", torch.cuda.current_device())
    src = torch.from_numpy(np.stack(frames)).to(dev)
    dst = torch.empty((len(frames), h, w, 3), dtype=torch.uint8, device=dev)
    check(_lib.load().w2l_resize_u8(current_stream(), len(frames), ptr(src), src.shape[1], src.shape[2], ptr(dst), h, w),
          "resize_u8")
    return list(dst.cpu().numpy())


def read_frames(a):
    """inference.py:185-213: the frames of `--face` and the frame rate.  Images through PIL; video from the uncompressed-AVI
    container this package writes (wav2lip_amd/container.py) - compressed video needs a codec (cv2.VideoCapture / ffmpeg),
    which this image does not have and which is not arithmetic of the path."""
    if not os.path.isfile(a.face):
        raise ValueError('--face argument must be a valid path to video/image file')
    if a.face.split('.')[1] in ['jpg', 'png', 'jpeg']:
        return [read_image_bgr(a.face)], a.fps
    from . import container
    try:
        clip = container.read_avi(a.face)
        frames, fps = clip["frames"], clip["fps"]
    except Exception as e:      # noqa: BLE001
        raise ValueError("--face %s: only images (jpg/png/jpeg) and uncompressed BGR AVI video can be read here (no video "
                         "codecs in this build): %s" % (a.face, e))
    print('Reading video frames...')
    frames = list(frames)
    if a.resize_factor > 1 and frames:
        h, w = frames[0].shape[:2]
        frames = resize_frames_u8(frames, (w // a.resize_factor, h // a.resize_factor))
    full_frames = []
    for frame in frames:
        if a.rotate:
            frame = np.ascontiguousarray(np.rot90(frame, k=-1))     # cv2.ROTATE_90_CLOCKWISE: pure data movement
        y1, y2, x1, x2 = a.crop
        if x2 == -1:
            x2 = frame.shape[1]
        if y2 == -1:
            y2 = frame.shape[0]
        full_frames.append(frame[y1:y2, x1:x2])
    return full_frames, fps


def main(argv=None, keep_frames=True, backend="nccl"):
    """inference.py:181-277 on the HIP path.  Same flags, same steps, same messages; differences, all on the file-format
    side: video input is the uncompressed AVI of wav2lip_amd/container.py (no codecs here), `--audio` must be a WAV (the
    reference shells out to ffmpeg for anything else), and the result - the reference's `temp/result.avi` + ffmpeg mux - is
    written as ONE AVI (BGR video + the driving audio as PCM16) at `--outfile`.

    Like the reference's loop (inference.py:249-274) this one STREAMS: every batch uploads only the frames it pastes into
    (deduplicated, `frame_idx` remapped) and its output frames go to the AVI writer as soon as they are back, so device and host
    memory are bounded by a few batches whatever the clip length.  `keep_frames` (default, what the tests use) additionally
    returns the list of output frames; the command line runs with keep_frames=False.

    Multi-GPU (`python -m torch.distributed.run --nproc-per-node N -m wav2lip_amd.inference ...`): every rank reads
    the inputs, face detection and the mel chunks are cut into contiguous per-rank shards (sharding.shard_range), each rank runs
    its shard on cuda:LOCAL_RANK, the generated frames are all-gathered in frame order in rounds of one batch per rank
    (sharding.gather_shards_to_writer) and rank 0 ALONE writes `--outfile` (and returns the frames; the other ranks return
    None)."""
    global args
    from . import sharding
    args = parse_args(argv)
    ranks = sharding.init_from_env(backend)
    sharded = ranks.dist is not None and ranks.world > 1
    say = print if ranks.writer else (lambda *a, **k: None)
    try:
        full_frames, fps = read_frames(args)
        say("Number of frames available for inference: " + str(len(full_frames)))
        if not args.audio.endswith('.wav'):
            raise ValueError("--audio %s: extracting audio from other containers is the reference's ffmpeg call; pass a .wav" % args.audio)
        wav = audio.load_wav(args.audio, 16000)
        dev = ranks.device
        mel = audio.melspectrogram_device(wav, dev)
        say(tuple(mel.shape))
        if bool(torch.isnan(mel).any()):
            raise ValueError('Mel contains nan! Using a TTS voice? Add a small epsilon noise to the wav file and try again')
        starts = mel_chunk_starts(mel.shape[1], fps)
        say("Length of mel chunks: {}".format(len(starts)))
        full_frames = full_frames[:len(starts)]
        if args.box[0] == -1:
            det = face_detect(full_frames if not args.static else [full_frames[0]], ranks=ranks)
            coords = [c for _, c in det]
        else:
            say('Using the specified bounding box instead of face detection...')
            coords = [tuple(args.box)] * len(full_frames)
        model = load_model(args.checkpoint_path, dev)
        say("Model loaded")
        n = len(starts)
        idx = [0 if args.static else i % len(full_frames) for i in range(n)]
        boxes = [validate_boxes([coords[0 if args.static else j]], *full_frames[j].shape[:2])[0] for j in idx]
        starts_dev = torch.tensor(starts, dtype=torch.int32, device=dev)
        runner = PipelinedRunner(model, args.wav2lip_batch_size, depth=LIPSYNC_DEPTH)
        bs = args.wav2lip_batch_size
        first, last = sharding.shard_range(n, ranks.rank, ranks.world)
        frame_h, frame_w = full_frames[0].shape[:2]
        out_frames, local, pending = [], [], []
        writer = None
        if ranks.writer:
            outdir = os.path.dirname(args.outfile)
            if outdir:
                os.makedirs(outdir, exist_ok=True)
            from . import container
            from scipy.io import wavfile
            sr, pcm = wavfile.read(args.audio)
            if pcm.dtype != np.int16:
                pcm = np.clip(np.round(audio._pcm_to_float32(pcm) * 32768.0), -32768, 32767).astype(np.int16)
            writer = container.AviWriter(args.outfile, fps, (frame_w, frame_h), audio=pcm, audio_sr=sr)
            writer.__enter__()
        try:
            def emit(f):
                writer.write(f)
                if keep_frames:
                    out_frames.append(f)

            def drain(ticket):
                for f in runner.result(ticket).cpu().numpy():
                    if sharded:
                        local.append(f)         # this rank's shard, exchanged below
                    else:
                        emit(f)                 # single process: straight to the writer, as the reference's loop does

            for lo in range(first, last, bs):
                hi = min(last, lo + bs)
                uniq = sorted(set(idx[lo:hi]))                              # a static image: one frame per batch
                remap = {j: k for k, j in enumerate(uniq)}
                frames_dev = torch.from_numpy(np.stack([full_frames[j] for j in uniq])).to(dev)
                pending.append(runner.submit(None, mel=mel, starts=starts_dev[lo:hi].contiguous(), frames=frames_dev,
                                             frame_idx=[remap[j] for j in idx[lo:hi]], boxes=boxes[lo:hi]))
                if len(pending) >= runner.depth:
                    drain(pending.pop(0))
            while pending:
                drain(pending.pop(0))
            if sharded:
                allf = sharding.gather_shards_to_writer(ranks.dist, np.stack(local) if local else None, n, ranks.rank, ranks.world,
                                                        device=dev, chunk=bs)
                if ranks.writer:
                    for f in allf:
                        emit(f)
        finally:
            if writer is not None:
                writer.__exit__(None, None, None)
        if not ranks.writer:
            return None
        return out_frames if keep_frames else None
    finally:
        ranks.close()


if __name__ == '__main__':
    main(keep_frames=False)
