"""The on-disk format behind the inference loop's output: what `cv2.VideoWriter` and the ffmpeg mux do at
inference.py:256-257 (open `temp/result.avi`), :272 (`out.write(f)`), :274 (`out.release()`) and :276-277
(`ffmpeg -y -i <audio> -i temp/result.avi ... <outfile>`), without cv2 or ffmpeg (neither exists offline).

The container is RIFF AVI 1.0 like the reference's intermediate file; the video stream is uncompressed 24-bit BGR ('DIB ',
BI_RGB) instead of the reference's DIVX (MPEG-4 ASP, a lossy encoder that is out of scope): every written frame decodes to exactly
the pixels the generator produced.  The audio stream is PCM16, interleaved with the video one frame's worth at a time — the mux of
:276-277 becomes part of the same file.  `read_avi` reads the files written here (and any BI_RGB-24 / PCM16 AVI), which also gives
the `--face <video>` side (inference.py:189-215, `cv2.VideoCapture`) an input format that works offline.

Host-side byte shuffling only: the frames are already uint8 BGR in host memory when they reach the writer (the paste-back kernel
w2l_resize_paste_u8 wrote them); nothing here is on the GPU path.  Limits: RIFF sizes are 32-bit (files below 4 GiB; the OpenDML
extension is not written), BI_RGB 24-bit video, PCM16 audio.
"""
import struct

import numpy as np

_AVIF_HASINDEX, _AVIF_ISINTERLEAVED = 0x10, 0x100
_AVIIF_KEYFRAME = 0x10
_MAX_RIFF = (1 << 32) - (1 << 20)


def _fps_rational(fps):
    """(rate, scale) with rate / scale = fps to 1e-3, as AVI stream headers store it (25 -> 25000/1000, 29.97 -> 29970/1000)"""
    fps = float(fps)
    if not (fps > 0):
        raise ValueError("fps must be positive (got %r)" % (fps,))
    return int(round(fps * 1000)), 1000


def _chunk(fourcc, payload):
    return fourcc + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")


def _list(kind, payload):
    return b"LIST" + struct.pack("<I", len(payload) + 4) + kind + payload


class AviWriter:
    """`cv2.VideoWriter(path, fourcc, fps, (frame_w, frame_h))` + the audio mux, for the loop at inference.py:256-274:

        out = AviWriter('result.avi', fps, (frame_w, frame_h), audio=pcm16, audio_sr=16000)
        for f in frames: out.write(f)          # uint8 [frame_h, frame_w, 3] BGR, as cv2 hands frames around
        out.release()

    Frames stream to disk as they arrive; the headers and the index are completed by `release()`.  Unlike cv2 (which drops a frame
    of the wrong size silently) a mismatching frame is an error."""

    def __init__(self, path, fps, frame_size, audio=None, audio_sr=16000):
        self.w, self.h = int(frame_size[0]), int(frame_size[1])
        if self.w <= 0 or self.h <= 0:
            raise ValueError("bad frame size %r" % (frame_size,))
        self.rate, self.scale = _fps_rational(fps)
        self.stride = (self.w * 3 + 3) & ~3                # DIB rows are padded to 4 bytes
        self.frame_bytes = self.stride * self.h
        self.audio = None
        if audio is not None:
            a = np.asarray(audio)
            if a.dtype != np.int16:
                raise ValueError("audio must be PCM16 (int16), got %s" % a.dtype)
            self.audio = np.ascontiguousarray(a.reshape(a.shape[0], -1))      # [samples, channels]
            self.audio_sr = int(audio_sr)
            self.block_align = 2 * self.audio.shape[1]
        self.path = path
        self.f = open(path, "wb")
        self.index = []                                    # (fourcc, offset relative to 'movi', size)
        self.n_frames = 0
        self.audio_pos = 0
        self.f.write(self._headers(0, 0))                  # placeholders of the final size
        self.movi_start = self.f.tell()                    # position of the 'movi' fourcc + 4 = first chunk
        self.released = False

    # -- headers ---------------------------------------------------------------------------------------------------------
    def _headers(self, n_frames, movi_bytes):
        n_streams = 2 if self.audio is not None else 1
        audio_rate = self.audio_sr * self.block_align if self.audio is not None else 0
        avih = struct.pack("<14I", int(round(1e6 * self.scale / self.rate)),            # dwMicroSecPerFrame
                           self.frame_bytes * self.rate // self.scale + audio_rate,     # dwMaxBytesPerSec
                           0, _AVIF_HASINDEX | (_AVIF_ISINTERLEAVED if n_streams == 2 else 0), n_frames, 0, n_streams,
                           self.frame_bytes, self.w, self.h, 0, 0, 0, 0)
        strh_v = struct.pack("<4s4sIHHIIIIIIII4h", b"vids", b"DIB ", 0, 0, 0, 0, self.scale, self.rate, 0, n_frames,
                             self.frame_bytes, 0xFFFFFFFF, 0, 0, 0, self.w, self.h)
        strf_v = struct.pack("<IiiHHIIiiII", 40, self.w, self.h, 1, 24, 0, self.frame_bytes, 0, 0, 0, 0)   # BITMAPINFOHEADER, bottom-up
        hdrl = _chunk(b"avih", avih) + _list(b"strl", _chunk(b"strh", strh_v) + _chunk(b"strf", strf_v))
        if self.audio is not None:
            n_samples = self.audio.shape[0]
            strh_a = struct.pack("<4s4sIHHIIIIIIII4h", b"auds", b"\x00\x00\x00\x00", 0, 0, 0, 0, self.block_align, audio_rate, 0,
                                 n_samples, audio_rate, 0xFFFFFFFF, self.block_align, 0, 0, 0, 0)
            strf_a = struct.pack("<HHIIHH", 1, self.audio.shape[1], self.audio_sr, audio_rate, self.block_align, 16)   # WAVEFORMAT PCM
            hdrl += _list(b"strl", _chunk(b"strh", strh_a) + _chunk(b"strf", strf_a))
        hdrl = _list(b"hdrl", hdrl)
        idx_bytes = 8 + 16 * len(self.index)
        riff_size = 4 + len(hdrl) + 12 + movi_bytes + idx_bytes
        return b"RIFF" + struct.pack("<I", riff_size) + b"AVI " + hdrl + b"LIST" + struct.pack("<I", movi_bytes + 4) + b"movi"

    # -- frames ----------------------------------------------------------------------------------------------------------
    def _emit(self, fourcc, payload):
        off = self.f.tell() - self.movi_start + 4          # idx1 offsets count from the 'movi' fourcc
        if self.f.tell() + len(payload) + 16 * (len(self.index) + 2) > _MAX_RIFF:
            raise ValueError("AVI 1.0 files are limited to 4 GiB (%s)" % self.path)
        self.f.write(fourcc + struct.pack("<I", len(payload)))
        self.f.write(payload)
        if len(payload) & 1:
            self.f.write(b"\x00")
        self.index.append((fourcc, off, len(payload)))

    def write(self, frame):
        if self.released:
            raise ValueError("write() after release()")
        frame = np.asarray(frame)
        if frame.dtype != np.uint8 or frame.shape != (self.h, self.w, 3):
            raise ValueError("frame must be uint8 [%d, %d, 3], got %s %s" % (self.h, self.w, frame.dtype, frame.shape))
        rows = frame[::-1].reshape(self.h, self.w * 3)     # DIBs are stored bottom-up; BGR order is the DIB order already
        if self.stride != self.w * 3:
            padded = np.zeros((self.h, self.stride), dtype=np.uint8)
            padded[:, :self.w * 3] = rows
            rows = padded
        self._emit(b"00db", np.ascontiguousarray(rows).tobytes())
        self.n_frames += 1
        if self.audio is not None:                          # this frame's share of the audio, cut on sample boundaries
            end = min(self.audio.shape[0], (self.n_frames * self.audio_sr * self.scale) // self.rate)
            if end > self.audio_pos:
                self._emit(b"01wb", self.audio[self.audio_pos:end].tobytes())
                self.audio_pos = end

    def isOpened(self):
        return not self.released

    def release(self):
        if self.released:
            return
        if self.audio is not None and self.audio_pos < self.audio.shape[0]:    # audio longer than the video: keep the tail
            self._emit(b"01wb", self.audio[self.audio_pos:].tobytes())
            self.audio_pos = self.audio.shape[0]
        movi_bytes = self.f.tell() - self.movi_start
        idx = b"".join(struct.pack("<4sIII", cc, _AVIIF_KEYFRAME, off, size) for cc, off, size in self.index)
        self.f.write(b"idx1" + struct.pack("<I", len(idx)) + idx)
        self.f.seek(0)
        self.f.write(self._headers(self.n_frames, movi_bytes))
        self.f.close()
        self.released = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()


def write_avi(path, frames, fps, audio=None, audio_sr=16000):
    """all frames at once: `frames` uint8 [T, H, W, 3] BGR (or a list of [H, W, 3])"""
    frames = list(frames)
    if not frames:
        raise ValueError("no frames to write")
    h, w = np.asarray(frames[0]).shape[:2]
    with AviWriter(path, fps, (w, h), audio=audio, audio_sr=audio_sr) as out:
        for f in frames:
            out.write(f)


def mux(audio_path, video_path, outfile):
    """inference.py:276-277 (`ffmpeg -y -i audio -i temp/result.avi outfile`) for a WAV and an AVI written here: one AVI with both
    streams.  The audio must be PCM16 (what audio.load_wav's inputs are after the reference's own ffmpeg step, inference.py:217-222)."""
    from scipy.io import wavfile
    sr, pcm = wavfile.read(audio_path)
    if pcm.dtype != np.int16:
        raise ValueError("mux: %s is not PCM16" % audio_path)
    v = read_avi(video_path)
    write_avi(outfile, v["frames"], v["fps"], audio=pcm, audio_sr=sr)


# ---------------------------------------------------------------- reader
def _walk(buf, start, end):
    """(fourcc, payload offset, payload size) of the chunks in buf[start:end]"""
    pos = start
    while pos + 8 <= end:
        cc = bytes(buf[pos:pos + 4])
        size = struct.unpack_from("<I", buf, pos + 4)[0]
        yield cc, pos + 8, size
        pos += 8 + size + (size & 1)


def read_avi(path):
    """-> dict(frames uint8 [T,H,W,3] BGR top-down, fps, audio int16 [n, channels] or None, audio_sr).  Reads uncompressed
    24-bit video and PCM16 audio (the subset AviWriter produces); anything else is an error, not a guess."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    if len(buf) < 12 or bytes(buf[:4]) != b"RIFF" or bytes(buf[8:12]) != b"AVI ":
        raise ValueError("%s is not a RIFF AVI file" % path)
    riff_end = min(len(buf), 8 + struct.unpack_from("<I", buf, 4)[0])
    streams, movi = [], None
    for cc, off, size in _walk(buf, 12, riff_end):
        if cc == b"LIST" and bytes(buf[off:off + 4]) == b"hdrl":
            for cc2, off2, size2 in _walk(buf, off + 4, off + size):
                if cc2 == b"LIST" and bytes(buf[off2:off2 + 4]) == b"strl":
                    st = {}
                    for cc3, off3, size3 in _walk(buf, off2 + 4, off2 + size2):
                        if cc3 == b"strh":
                            st["type"] = bytes(buf[off3:off3 + 4])
                            st["scale"], st["rate"] = struct.unpack_from("<II", buf, off3 + 20)
                        elif cc3 == b"strf":
                            st["strf"] = bytes(buf[off3:off3 + size3])
                    streams.append(st)
        elif cc == b"LIST" and bytes(buf[off:off + 4]) == b"movi":
            movi = (off + 4, off + size)
    if movi is None or not streams:
        raise ValueError("%s: no stream headers / movi list" % path)
    vid = [i for i, s in enumerate(streams) if s.get("type") == b"vids"]
    aud = [i for i, s in enumerate(streams) if s.get("type") == b"auds"]
    if not vid:
        raise ValueError("%s: no video stream" % path)
    vs = streams[vid[0]]
    _, w, h, _, bits, comp = struct.unpack_from("<IiiHHI", vs["strf"], 0)
    if comp != 0 or bits != 24:
        raise ValueError("%s: only uncompressed 24-bit video is supported (compression %#x, %d bits)" % (path, comp, bits))
    top_down = h < 0
    h = abs(h)
    stride = (w * 3 + 3) & ~3
    vtag, atag = b"%02ddb" % vid[0], (b"%02dwb" % aud[0]) if aud else None
    frames, pcm = [], []
    for cc, off, size in _walk(buf, movi[0], movi[1]):
        if cc == vtag or cc == vtag[:2] + b"dc":
            if size != stride * h:
                raise ValueError("%s: video chunk of %d bytes, expected %d" % (path, size, stride * h))
            rows = np.frombuffer(buf, dtype=np.uint8, count=size, offset=off).reshape(h, stride)[:, :w * 3].reshape(h, w, 3)
            frames.append(rows if top_down else rows[::-1])
        elif atag is not None and cc == atag:
            pcm.append(np.frombuffer(buf, dtype=np.uint8, count=size, offset=off))
    out = {"frames": np.ascontiguousarray(np.stack(frames)) if frames else np.zeros((0, h, w, 3), np.uint8),
           "fps": vs["rate"] / float(vs["scale"]), "audio": None, "audio_sr": None}
    if aud:
        fmt, ch, sr, _, _, bps = struct.unpack_from("<HHIIHH", streams[aud[0]]["strf"], 0)
        if fmt != 1 or bps != 16:
            raise ValueError("%s: only PCM16 audio is supported (format %d, %d bits)" % (path, fmt, bps))
        raw = np.concatenate(pcm) if pcm else np.zeros(0, np.uint8)
        out["audio"] = raw.view(np.int16).reshape(-1, ch).copy()
        out["audio_sr"] = sr
    return out
