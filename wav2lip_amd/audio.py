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


def _pcm_to_float32(data):
    """the float32 conversion librosa.load gets from soundfile for the WAV sample formats scipy.io.wavfile returns"""
    if data.dtype == np.int16:
        return data.astype(np.float32) / np.float32(32768.0)
    if data.dtype == np.int32:
        return (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    if data.dtype == np.uint8:
        return (data.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    if data.dtype in (np.float32, np.float64):
        return data.astype(np.float32)
    raise ValueError("load_wav: unsupported WAV sample format %s" % data.dtype)


def load_wav(path, sr):
    """audio.py:9-10: `librosa.core.load(path, sr=sr)[0]` for WAV containers: decode to float32 (soundfile's scaling), mono mix by
    mean, and — when the file's rate differs from `sr` — librosa's 'kaiser_best' resampling on the HIP device (`resample`).
    Other containers (mp3/mp4 audio tracks) went through ffmpeg in the reference (inference.py:217-222) and stay out of scope."""
    from scipy.io import wavfile
    file_sr, data = wavfile.read(path)
    x = _pcm_to_float32(data)
    if x.ndim > 1:
        x = x.T.mean(axis=0)               # librosa.to_mono: np.mean over the channel axis of the (channels, n) array
    x = np.ascontiguousarray(x, dtype=np.float32)
    if file_sr != sr:
        x = resample(x, file_sr, sr)
    return x


# ---------------------------------------------------------------- librosa.load's resampling (resampy 'kaiser_best')
_KAISER_BEST = None


def _kaiser_best_filter():
    """resampy.filters.sinc_window(num_zeros=64, precision=9, window=kaiser(beta=14.769656459379492),
    rolloff=0.9475937167399596): the half window (float64, 64 * 512 + 1 samples) and the table step 512.  Host-built constant
    table, like the mel basis."""
    global _KAISER_BEST
    if _KAISER_BEST is None:
        from scipy import signal
        num_zeros, num_bits, rolloff, beta = 64, 2 ** 9, 0.9475937167399596, 14.769656459379492
        n = num_bits * num_zeros
        sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
        taper = signal.windows.kaiser(2 * n + 1, beta)[n:]
        _KAISER_BEST = (taper * sinc_win, num_bits)
    return _KAISER_BEST


def resample(y, orig_sr, target_sr, device=None):
    """librosa 0.7.0 `resample(y, orig_sr, target_sr, res_type='kaiser_best', fix=True, scale=False)` for 1-D float32 input:
    resampy's sinc interpolation (csrc/audio_mel.hip: w2l_resample_sinc, one thread per output sample, the reference's term
    order and float32 rounding), then fix_length to ceil(len * ratio).  The host builds the filter table and the
    interpolator's time registers (repeated float64 addition, as the reference loop accumulates them); there is no CPU path."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    if orig_sr == target_sr:
        return y
    if y.ndim != 1:
        raise ValueError("resample: 1-D input expected")
    ratio = float(target_sr) / orig_sr
    n_out = int(y.shape[0] * ratio)
    if n_out < 1:
        raise ValueError("Input signal length=%d is too small to resample from %d->%d" % (y.shape[0], orig_sr, target_sr))
    if not torch.cuda.is_available():
        raise RuntimeError("audio.resample needs a HIP device (no CPU path)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    win, num_table = _kaiser_best_filter()
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    tr = np.zeros(n_out, dtype=np.float64)
    if n_out > 1:
        np.cumsum(np.full(n_out - 1, 1.0 / ratio, dtype=np.float64), out=tr[1:])
    lib = _lib.load()
    with torch.cuda.device(dev):
        xd = torch.from_numpy(y).to(dev)
        trd, wind, deltad = (torch.from_numpy(a).to(dev) for a in (tr, win, delta))
        out = torch.empty(n_out, dtype=torch.float32, device=dev)
        check(lib.w2l_resample_sinc(_lib.current_stream(), ptr(xd), y.shape[0], ptr(trd), n_out, ratio, ptr(wind), ptr(deltad),
                                    win.shape[0], num_table, ptr(out)), "resample_sinc")
        y_hat = out.cpu().numpy()
    n_samples = int(np.ceil(y.shape[0] * ratio))
    if y_hat.shape[0] > n_samples:
        y_hat = y_hat[:n_samples]
    elif y_hat.shape[0] < n_samples:
        y_hat = np.pad(y_hat, (0, n_samples - y_hat.shape[0]), mode="constant")
    return np.ascontiguousarray(y_hat, dtype=np.float32)


_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP       # 14.999999999999998 in float64 (librosa computes it, it does not write 15.0)
_LOGSTEP = np.log(6.4) / 27.0


def _slaney_hz_to_mel(f):
    f = np.atleast_1d(np.asarray(f, dtype=np.float64))
    lin = f / _F_SP
    log = _MIN_LOG_MEL + np.log(np.maximum(f, 1e-12) / _MIN_LOG_HZ) / _LOGSTEP
    return np.where(f >= _MIN_LOG_HZ, log, lin)


def _slaney_mel_to_hz(m):
    m = np.atleast_1d(np.asarray(m, dtype=np.float64))
    return np.where(m >= _MIN_LOG_MEL, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), _F_SP * m)


def _build_mel_basis():
    """audio.py:98-101 -> librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax): triangular Slaney filters with area
    normalisation, float32 [num_mels, 1 + n_fft//2] - with librosa 0.7.0's rounding points: the triangles are rounded to
    float32 first (assignment into its float32 `weights`), THEN scaled by the float64 area norm and rounded again (its
    in-place `weights *= enorm[:, np.newaxis]`)"""
    assert hp.fmax <= hp.sample_rate // 2
    n_bins = 1 + hp.n_fft // 2
    bin_hz = np.linspace(0.0, hp.sample_rate / 2.0, n_bins)
    lo, hi = _slaney_hz_to_mel(hp.fmin)[0], _slaney_hz_to_mel(hp.fmax)[0]
    edges = _slaney_mel_to_hz(np.linspace(lo, hi, hp.num_mels + 2))
    width = np.diff(edges)
    d = edges[:, None] - bin_hz[None, :]                      # [num_mels+2, n_bins]
    rising = -d[:-2] / width[:-1, None]
    falling = d[2:] / width[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling)).astype(np.float32)          # first rounding
    area = 2.0 / (edges[2:] - edges[:-2])                                           # float64
    return (tri.astype(np.float64) * area[:, None]).astype(np.float32)             # second rounding


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
