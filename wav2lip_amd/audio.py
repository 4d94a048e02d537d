"""`audio` module with the reference's surface (audio.py:9-136): `load_wav(path, sr)`, `melspectrogram(wav)`,
module-global `hp`.  The spectrogram runs as one fused HIP kernel (csrc/audio_mel.hip); the host only builds
the constant tables once (Slaney mel filterbank as librosa.filters.mel builds it, periodic Hann window as
scipy.signal.get_window builds it) and moves samples to the device.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr
from .hparams import hparams as hp

_ctx = {}          # device index -> w2l_mel handle
_mel_basis = None


def load_wav(path, sr):
    """audio.py:9-10.  In scope: PCM16 WAV already at `sr` (librosa.load's decode path: int16/32768, mono mean).
    Other containers / sample rates need ffmpeg / resampy, which are outside the hot path."""
    from scipy.io import wavfile
    file_sr, data = wavfile.read(path)
    if file_sr != sr:
        raise ValueError("load_wav: %s is %d Hz; resampling to %d Hz is not implemented" % (path, file_sr, sr))
    if data.dtype != np.int16:
        raise ValueError("load_wav: only PCM16 WAV is supported (got %s)" % data.dtype)
    x = data.astype(np.float32) / np.float32(32768.0)
    if x.ndim > 1:
        x = x.mean(axis=1, dtype=np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def _slaney_hz_to_mel(f):
    f = np.atleast_1d(np.asarray(f, dtype=np.float64))
    lin = f / (200.0 / 3)
    log = 15.0 + np.log(np.maximum(f, 1e-12) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m):
    m = np.atleast_1d(np.asarray(m, dtype=np.float64))
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)


def _build_mel_basis():
    """audio.py:98-101 -> librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax): triangular Slaney filters with
    area normalisation, float32 [num_mels, 1 + n_fft//2]"""
    assert hp.fmax <= hp.sample_rate // 2
    n_bins = 1 + hp.n_fft // 2
    bin_hz = np.linspace(0.0, hp.sample_rate / 2.0, n_bins)
    lo, hi = _slaney_hz_to_mel(hp.fmin)[0], _slaney_hz_to_mel(hp.fmax)[0]
    edges = _slaney_mel_to_hz(np.linspace(lo, hi, hp.num_mels + 2))
    width = np.diff(edges)
    d = edges[:, None] - bin_hz[None, :]                      # [num_mels+2, n_bins]
    rising = -d[:-2] / width[:-1, None]
    falling = d[2:] / width[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling))
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return tri.astype(np.float32)


def _window():
    from scipy import signal
    return np.ascontiguousarray(signal.get_window("hann", hp.win_size, fftbins=True), dtype=np.float64)


def _context(device):
    global _mel_basis
    idx = device.index if device.index is not None else torch.cuda.current_device()
    h = _ctx.get(idx)
    if h is None:
        if (hp.n_fft, hp.hop_size, hp.win_size, hp.num_mels, hp.sample_rate) != (800, 200, 800, 80, 16000) \
                or hp.use_lws or not (hp.signal_normalization and hp.symmetric_mels and hp.preemphasize
                                      and hp.allow_clipping_in_normalization):
            raise RuntimeError("the HIP mel kernel is specialised to the reference's hparams (hparams.py:32-69)")
        if _mel_basis is None:
            _mel_basis = _build_mel_basis()
        win = _window()
        h = C.c_void_p()
        with torch.cuda.device(idx):
            check(_lib.load().w2l_mel_create(_mel_basis.ctypes.data_as(C.c_void_p), win.ctypes.data_as(C.c_void_p),
                                             C.byref(h)), "mel_create")
        _ctx[idx] = h
    return h


def melspectrogram_device(wav, device=None):
    """wav: 1-D float32 numpy array or torch tensor -> torch float32 [80, 1 + N//200] on the HIP device"""
    if not torch.cuda.is_available():
        raise RuntimeError("wav2lip_amd.audio: no HIP device; there is no CPU path")
    if isinstance(wav, np.ndarray):
        wav = torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32))
    device = torch.device(device) if device is not None else (wav.device if wav.is_cuda else torch.device("cuda"))
    w = wav.to(device=device, dtype=torch.float32).contiguous().view(-1)
    lib = _lib.load()
    T = lib.w2l_mel_num_frames(w.numel())
    mel = torch.empty((hp.num_mels, T), device=device, dtype=torch.float32)
    with torch.cuda.device(device):
        check(lib.w2l_melspectrogram(_context(device), _lib.current_stream(), ptr(w), w.numel(), ptr(mel)),
              "melspectrogram")
    return mel


def melspectrogram(wav):
    """audio.py:45-51: float32 numpy [80, T] (computed on the HIP device)"""
    return melspectrogram_device(wav).cpu().numpy()
