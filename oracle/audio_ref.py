"""Oracle for audio.melspectrogram in numpy/scipy.  TEST INFRASTRUCTURE.  PINNED to the reference's audio.py
(tests/golden/make_golden_datapath.py runs the real module with a stub librosa that delegates to the three functions below:
bit-equal), PARITY UNPINNED INSIDE librosa: the reference calls librosa==0.7.0 (requirements.txt:1; audio.py:10,61,100), a
third-party dependency that is neither in /root/reference nor installed here, and the reference has no test or golden
vector for it.  stft / mel_basis / load_wav_pcm16 restate librosa 0.7.0's published algorithm for the three calls the
reference makes and are cross-checked against torch.stft and known-answer properties (tests/test_oracle.py).

Dtype of every intermediate, as librosa 0.7.0 / numpy produce it (restated from the published 0.7.0 source, which is not
available offline - stated so that a reader with the source can check line by line):
  load            soundfile read(dtype=float32): int16/32768 -> float32; to_mono = np.mean(axis=0) in float32
  preemphasis     scipy lfilter -> float64
  stft            get_window('hann', 800, fftbins=True) float64, pad_center no-op (win_length == n_fft); np.pad(reflect)
                  float64; util.frame = strided view; fft_window * y_frames float64; numpy FFT in float64; assignment into
                  the pre-allocated complex64 matrix rounds real and imaginary parts once.  (0.7.0 calls its fft lib's
                  `rfft`; a release that calls `fft(...)[:401]` would differ in the last float64 bit before that rounding.)
  np.abs(D)       float32
  mel basis       float32 weights, float64 ramps -> rounded per row, then in-place `*= enorm` (second rounding), see mel_basis()
  np.dot          float32 x float32 -> float32 (BLAS sgemm; summation order is the BLAS's, tolerance 1e-4 in the tests)

Step-by-step (reference line -> restatement):
  audio.py:20-23  preemphasis  scipy.signal.lfilter([1,-0.97],[1],wav)                  -> float64
  audio.py:57-61  _stft        librosa.stft(y, n_fft=800, hop_length=200, win_length=800):
                               periodic Hann (scipy get_window 'hann', fftbins=True), center=True => np.pad reflect
                               by n_fft//2, frames of 800 every 200, FFT in float64, stored as complex64
  audio.py:92-101 mel basis    librosa.filters.mel(16000, 800, n_mels=80, fmin=55, fmax=7600): Slaney scale
                               (htk=False), area ("slaney") normalisation, float32
  audio.py:103-105 _amp_to_db  20*log10(max(1e-5, x))  (float32 under the reference's numpy 1.17 value-based casting)
  audio.py:47,110-116          - ref_level_db(20); clip(8*((S+100)/100) - 4, -4, 4)
"""
import numpy as np
from scipy import signal

SR, N_FFT, HOP, WIN, N_MELS, FMIN, FMAX = 16000, 800, 200, 800, 80, 55, 7600
PREEMPH, MIN_DB, REF_DB, MAX_ABS = 0.97, -100, 20, 4.


def load_wav_pcm16(path):
    """audio.py:9-10 for the configurations in scope: a 16 kHz PCM16 WAV -> float32 in [-1, 1) (int16/32768,
    the soundfile convention librosa.load uses), mono-mixed by mean.  Resampling is out of scope."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if sr != SR:
        raise ValueError("only %d Hz input is in scope (got %d)" % (SR, sr))
    if data.dtype != np.int16:
        raise ValueError("only PCM16 input is in scope")
    x = data.astype(np.float32) / 32768.0
    if x.ndim > 1:
        x = x.mean(axis=1)
    return x.astype(np.float32)


def preemphasis(wav):
    return signal.lfilter([1, -PREEMPH], [1], wav)


def hann_window():
    return signal.get_window("hann", WIN, fftbins=True)


def stft(y):
    y = np.asarray(y, dtype=np.float64)
    win = hann_window()
    yp = np.pad(y, N_FFT // 2, mode="reflect")
    n_frames = 1 + (len(yp) - N_FFT) // HOP
    idx = np.arange(N_FFT)[:, None] + HOP * np.arange(n_frames)[None, :]
    frames = yp[idx]
    return np.fft.rfft(win[:, None] * frames, axis=0).astype(np.complex64)


def _hz_to_mel(frequencies):
    """librosa 0.7.0 core/time_frequency.py hz_to_mel(htk=False), line by line (float64 throughout; the scalar and the array
    branch are both kept because the reference reaches it with Python ints: fmin=55, fmax=7600)"""
    frequencies = np.asanyarray(frequencies)
    f_min = 0.0
    f_sp = 200.0 / 3
    mels = (frequencies - f_min) / f_sp                     # float64
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp               # 14.999999999999998, NOT 15.0
    logstep = np.log(6.4) / 27.0
    if frequencies.ndim:
        log_t = (frequencies >= min_log_hz)
        mels[log_t] = min_log_mel + np.log(frequencies[log_t] / min_log_hz) / logstep
    elif frequencies >= min_log_hz:
        mels = min_log_mel + np.log(frequencies / min_log_hz) / logstep
    return mels


def _mel_to_hz(mels):
    """librosa 0.7.0 mel_to_hz(htk=False), float64"""
    mels = np.asanyarray(mels)
    f_min = 0.0
    f_sp = 200.0 / 3
    freqs = f_min + f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        log_t = (mels >= min_log_mel)
        freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def mel_basis():
    """librosa 0.7.0 filters.mel(sr=16000, n_fft=800, n_mels=80, fmin=55, fmax=7600, htk=False, norm=1, dtype=np.float32)
    with ITS dtypes: `weights` is allocated float32, every triangle row (computed in float64) is rounded once on assignment
    into it, and the Slaney area normalisation is the in-place `weights *= enorm[:, np.newaxis]` - a float64 multiply of the
    already-rounded float32 rows, rounded to float32 a second time.  (Rounds 1-2 built triangle x norm in float64 and cast
    once: 184 of the 739 non-zero entries differed by one ulp.)"""
    fftfreqs = np.linspace(0, float(SR) / 2, int(1 + N_FFT // 2), endpoint=True)          # fft_frequencies, float64
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(FMIN), _hz_to_mel(FMAX), N_MELS + 2))       # mel_frequencies, float64
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)                                             # float64 [82, 401]
    weights = np.zeros((N_MELS, int(1 + N_FFT // 2)), dtype=np.float32)
    for i in range(N_MELS):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))                               # float64 -> float32 (1st rounding)
    enorm = 2.0 / (mel_f[2:N_MELS + 2] - mel_f[:N_MELS])                                   # float64
    weights *= enorm[:, np.newaxis]                                                        # f32 * f64 -> f64 -> float32 (2nd)
    return weights


_BASIS = None


def melspectrogram(wav):
    global _BASIS
    if _BASIS is None:
        _BASIS = mel_basis()
    D = stft(preemphasis(wav))
    S = np.abs(D)                                        # float32
    M = np.dot(_BASIS, S)                                # float32
    min_level = np.float32(np.exp(MIN_DB / 20 * np.log(10)))
    db = (20 * np.log10(np.maximum(min_level, M))).astype(np.float32) - np.float32(REF_DB)
    out = np.clip((2 * MAX_ABS) * ((db - MIN_DB) / (-MIN_DB)) - MAX_ABS, -MAX_ABS, MAX_ABS)
    return out.astype(np.float32)
