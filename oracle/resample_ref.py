"""Oracle for the sample-rate conversion inside `audio.load_wav` (reference audio.py:9-10: `librosa.core.load(path, sr=16000)`).
TEST INFRASTRUCTURE.  PARITY UNPINNED: the arithmetic lives in third-party code that is neither under /root/reference nor
installed here — librosa==0.7.0 (requirements.txt:1), whose `load` calls `resample(..., res_type='kaiser_best')`, which calls
`resampy.resample(y, sr_native, sr, filter='kaiser_best')` (resampy is an unpinned dependency of librosa 0.7.0; 0.2.2 was current
at the pin date).  The reference holds no test or golden vector for it.  This file restates the published algorithm of both
layers; tests/test_oracle.py checks it against known-answer properties (identity at equal rates, band-limited sines against the
analytic signal, the DC gain of the filter, linearity).

librosa 0.7.0 `core.audio.load` (mono=True, dtype=float32):
    y = soundfile read as float32   (int16 / 32768, int32 / 2**31, uint8 (x - 128) / 128, floats as they are)
    y = to_mono(y) = mean over channels
    y = resample(y, sr_native, sr, res_type='kaiser_best')        only when sr_native != sr
librosa 0.7.0 `core.audio.resample` (fix=True, scale=False):
    ratio = float(target_sr) / orig_sr;  n_samples = int(ceil(len(y) * ratio))
    y_hat = resampy.resample(y, orig_sr, target_sr, filter='kaiser_best', axis=-1)
    y_hat = util.fix_length(y_hat, n_samples)                      zero-pad or trim at the end
    return ascontiguousarray(y_hat, dtype=y.dtype)
resampy 0.2.2 `core.resample` / `interpn.resample_f` (band-limited sinc interpolation, J.O. Smith's scheme):
    filter 'kaiser_best' = sinc_window(num_zeros=64, precision=9, window=kaiser(beta=14.769656459379492),
                                        rolloff=0.9475937167399596): a half window of 64 * 512 + 1 float64 samples
    sample_ratio = sr_new / sr_orig;  len(y) = int(len(x) * sample_ratio);  y = zeros(dtype = x.dtype = float32)
    interp_win *= sample_ratio when sample_ratio < 1;  interp_delta[:-1] = diff(interp_win)
    time_register accumulates 1 / sample_ratio per output sample (repeated float64 addition, NOT t * increment);
    each output sample adds the left wing (x[n], x[n-1], ...) then the right wing (x[n+1], ...), every term rounded into the
    float32 accumulator as it is added (numba keeps y's dtype).
"""
import numpy as np
from scipy import signal

NUM_ZEROS, PRECISION = 64, 9
KAISER_BETA = 14.769656459379492
ROLLOFF = 0.9475937167399596


def kaiser_best_filter():
    """resampy.filters.sinc_window(num_zeros=64, precision=9, window=kaiser(beta), rolloff): (half window f64, table step 512)"""
    num_bits = 2 ** PRECISION
    n = num_bits * NUM_ZEROS
    sinc_win = ROLLOFF * np.sinc(ROLLOFF * np.linspace(0, NUM_ZEROS, num=n + 1, endpoint=True))
    taper = signal.windows.kaiser(2 * n + 1, KAISER_BETA)[n:]
    return taper * sinc_win, num_bits


def time_registers(n_out, sample_ratio):
    """the value of resample_f's `time_register` at every output sample: repeated float64 addition of the increment"""
    inc = 1.0 / sample_ratio
    tr = np.empty(n_out, dtype=np.float64)
    if n_out:
        tr[0] = 0.0
        if n_out > 1:
            np.cumsum(np.full(n_out - 1, inc, dtype=np.float64), out=tr[1:])     # sequential: ((inc + inc) + inc) + ...
    return tr


def resampy_resample(x, sr_orig, sr_new):
    """resampy.resample(x, sr_orig, sr_new, filter='kaiser_best') for 1-D float32 x"""
    x = np.asarray(x, dtype=np.float32)
    sample_ratio = float(sr_new) / sr_orig
    n_out = int(x.shape[0] * sample_ratio)
    if n_out < 1:
        raise ValueError("Input signal length=%d is too small to resample from %d->%d" % (x.shape[0], sr_orig, sr_new))
    interp_win, num_table = kaiser_best_filter()
    if sample_ratio < 1:
        interp_win = interp_win * sample_ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, sample_ratio)
    index_step = int(scale * num_table)
    nwin = interp_win.shape[0]
    n_orig = x.shape[0]
    tr = time_registers(n_out, sample_ratio)
    y = np.zeros(n_out, dtype=np.float32)
    x64 = x.astype(np.float64)
    for t in range(n_out):
        time_register = tr[t]
        n = int(time_register)
        acc = np.float32(0.0)
        # left wing: x[n], x[n-1], ...
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        if i_max > 0:
            idx = offset + np.arange(i_max) * index_step
            terms = (interp_win[idx] + eta * interp_delta[idx]) * x64[n - np.arange(i_max)]
            for v in terms:
                acc = np.float32(np.float64(acc) + v)
        # right wing: x[n+1], x[n+2], ...
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        if k_max > 0:
            idx = offset + np.arange(k_max) * index_step
            terms = (interp_win[idx] + eta * interp_delta[idx]) * x64[n + 1 + np.arange(k_max)]
            for v in terms:
                acc = np.float32(np.float64(acc) + v)
        y[t] = acc
    return y


def librosa_resample(y, orig_sr, target_sr):
    """librosa 0.7.0 core.audio.resample(y, orig_sr, target_sr, res_type='kaiser_best', fix=True, scale=False)"""
    y = np.asarray(y, dtype=np.float32)
    if orig_sr == target_sr:
        return y
    ratio = float(target_sr) / orig_sr
    n_samples = int(np.ceil(y.shape[-1] * ratio))
    y_hat = resampy_resample(y, orig_sr, target_sr)
    if y_hat.shape[0] > n_samples:
        y_hat = y_hat[:n_samples]
    elif y_hat.shape[0] < n_samples:
        y_hat = np.pad(y_hat, (0, n_samples - y_hat.shape[0]), mode="constant")
    return np.ascontiguousarray(y_hat, dtype=np.float32)


def pcm_to_float32(data):
    """soundfile's conversion to float32 for the WAV sample formats scipy.io.wavfile returns"""
    if data.dtype == np.int16:
        return data.astype(np.float32) / np.float32(32768.0)
    if data.dtype == np.int32:
        return (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    if data.dtype == np.uint8:
        return (data.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    if data.dtype in (np.float32, np.float64):
        return data.astype(np.float32)
    raise ValueError("unsupported WAV sample format %s" % data.dtype)


def load_wav(path, sr):
    """audio.py:9-10: librosa.core.load(path, sr=sr)[0] for WAV containers"""
    from scipy.io import wavfile
    file_sr, data = wavfile.read(path)
    x = pcm_to_float32(data)
    if x.ndim > 1:
        x = x.T.mean(axis=0)     # to_mono: np.mean(y, axis=0) on the (channels, n) array
    return librosa_resample(np.ascontiguousarray(x, dtype=np.float32), file_sr, sr)
