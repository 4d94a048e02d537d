"""The oracle (CPU restatement) against the committed golden vectors and known answers.  No GPU."""
import numpy as np
import torch

from oracle import audio_ref, datagen_ref, models_ref
from wav2lip_amd import synthetic as synth
from wav2lip_amd import models as amd_models


def _shapes(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def test_generator_oracle_matches_reference_golden(golden):
    sd = synth.synthetic_state_dict(_shapes(amd_models.Wav2Lip()), seed=0)
    img, mel = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(synth.face_crops_u8(2, seed=1),
                                                                    synth.mel_windows(2, seed=1)))
    y = models_ref.wav2lip_forward(sd, torch.from_numpy(mel), torch.from_numpy(img)).numpy()
    # the golden file holds the reference's own output; same torch build => bit-exact, other builds => fp32 noise
    assert np.abs(y - golden["gen_out_b2"]).max() <= 2e-6
    assert 0.05 < y.std() < 0.3, "synthetic weights must not saturate the sigmoid"


def test_generator_oracle_5d_folding(golden):
    sd = synth.synthetic_state_dict(_shapes(amd_models.Wav2Lip()), seed=0)
    img, mel = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(synth.face_crops_u8(2, seed=1),
                                                                    synth.mel_windows(2, seed=1)))
    img5 = torch.from_numpy(img).unsqueeze(0).permute(0, 2, 1, 3, 4).contiguous()
    mel5 = torch.from_numpy(mel).unsqueeze(0)
    y5 = models_ref.wav2lip_forward(sd, mel5, img5).numpy()
    assert y5.shape == (1, 3, 2, 96, 96)
    assert np.abs(y5 - golden["gen_out_5d"]).max() <= 2e-6
    # t-major folding: time step t of the 5-D output is sample t of the 4-D batch
    assert np.abs(y5[0, :, 1] - golden["gen_out_b2"][1]).max() <= 2e-6


def test_syncnet_and_disc_oracle_match_reference_golden(golden):
    sds = synth.synthetic_state_dict(_shapes(amd_models.SyncNet_color()), seed=2)
    a, v = models_ref.syncnet_forward(sds, torch.from_numpy(synth.mel_windows(2, seed=3)).unsqueeze(1),
                                      torch.from_numpy(synth.sync_faces(2, seed=3)))
    assert np.abs(a.numpy() - golden["sync_audio_emb"]).max() <= 1e-6
    assert np.abs(v.numpy() - golden["sync_face_emb"]).max() <= 1e-6
    assert np.allclose(np.linalg.norm(a.numpy(), axis=1), 1.0, atol=1e-5)
    sdd = synth.synthetic_state_dict(_shapes(amd_models.Wav2Lip_disc_qual()), seed=4)
    p = models_ref.disc_forward(sdd, torch.from_numpy(synth.disc_frames(1, 2, seed=5)))
    assert p.shape == (2, 1)
    assert np.abs(p.numpy() - golden["disc_pred"]).max() <= 1e-6


def test_mel_known_answers(golden):
    basis = audio_ref.mel_basis()
    assert basis.shape == (80, 401) and basis.dtype == np.float32
    assert int((basis != 0).sum()) == 739
    assert abs(float(basis.sum()) - 3.999498) < 1e-5
    assert (basis.sum(axis=1) > 0).all()
    # librosa 0.7.0's rounding points (float32 rows, then the in-place float64 area norm): stored known answer of the
    # float32 array, byte for byte; the product's host-side basis (wav2lip_amd/audio.py) must be the same array
    import hashlib
    assert hashlib.sha256(np.ascontiguousarray(basis).tobytes()).hexdigest() == \
        "a8457c06c33e239af4b81bc79320aa0bec753d5050e50e8a7fddee9f42c1b0de"
    from wav2lip_amd import audio as amd_audio
    assert np.array_equal(amd_audio._build_mel_basis(), basis)
    # and it is NOT the single-rounding variant of rounds 1-2 (triangle x norm in float64, one cast)
    fftfreqs = np.linspace(0, 8000.0, 401)
    mel_f = audio_ref._mel_to_hz(np.linspace(audio_ref._hz_to_mel(55), audio_ref._hz_to_mel(7600), 82))
    ramps = np.subtract.outer(mel_f, fftfreqs)
    fd = np.diff(mel_f)
    once = np.stack([np.maximum(0, np.minimum(-ramps[i] / fd[i], ramps[i + 2] / fd[i + 1])) * (2.0 / (mel_f[i + 2] - mel_f[i]))
                     for i in range(80)]).astype(np.float32)
    assert 100 < int((once != basis).sum()) < 300 and np.abs(once - basis).max() < 1e-8
    m = audio_ref.melspectrogram(synth.sine_wav())
    assert m.shape == (80, 241) and m.dtype == np.float32       # 48000 samples -> 1 + 48000//200
    assert not np.isnan(m).any() and m.min() >= -4 and m.max() <= 4
    assert np.abs(m - golden["mel_sine3s"]).max() <= 1e-5
    assert np.abs(audio_ref.melspectrogram(synth.noise_wav(16000, seed=7)) - golden["mel_noise1s"]).max() <= 1e-5
    # a 440 Hz tone peaks in the mel band containing 440 Hz
    band = int(np.argmax(m[:, 100]))
    assert basis[band, 22] > 0                                   # bin 22 = 440 Hz at 20 Hz/bin


def test_stft_against_torch_stft():
    y = audio_ref.preemphasis(synth.noise_wav(8000, seed=11))
    D = audio_ref.stft(y)
    Dt = torch.stft(torch.from_numpy(y), 800, 200, 800, window=torch.from_numpy(audio_ref.hann_window()),
                    center=True, pad_mode="reflect", return_complex=True).numpy()
    assert D.shape == Dt.shape == (401, 41)
    assert np.abs(D.astype(np.complex128) - Dt).max() <= 1e-6 * np.abs(Dt).max()


def test_preemphasis_is_float64_first_order_filter():
    x = synth.noise_wav(1000, seed=5)
    y = audio_ref.preemphasis(x)
    assert y.dtype == np.float64 and y[0] == x[0]
    assert np.array_equal(y[1:], x[1:].astype(np.float64) + (-0.97 * x[:-1].astype(np.float64)))


def test_mel_chunk_indices_bit_exact(golden):
    s = datagen_ref.mel_chunk_starts(241, 25.0)
    assert len(s) == 72
    assert s[:8] == [0, 3, 6, 9, 12, 16, 19, 22] and s[-4:] == [217, 220, 224, 225]
    assert np.array_equal(np.asarray(s, dtype=np.int32), golden["chunk_starts_T241_fps25"])
    assert np.array_equal(np.asarray(datagen_ref.mel_chunk_starts(1000, 30.0), dtype=np.int32),
                          golden["chunk_starts_T1000_fps30"])
    # training-side formula equals the inference-side formula
    for n in range(20000):
        assert datagen_ref.crop_audio_window_start(n) == int(n * (80. / 25.))


def test_datagen_semantics():
    faces = synth.face_crops_u8(3, seed=9)
    mels = synth.mel_windows(3, seed=9)
    img, mel = datagen_ref.datagen_batch(faces, mels)
    assert img.dtype == np.float64 and img.shape == (3, 96, 96, 6) and mel.shape == (3, 80, 16, 1)
    assert (img[:, 48:, :, :3] == 0).all() and np.array_equal(img[..., 3:], faces / 255.)
    assert np.array_equal(img[:, :48, :, :3], faces[:, :48] / 255.)
    x, m = datagen_ref.to_model_inputs(img, mel)
    assert x.dtype == np.float32 and x.shape == (3, 6, 96, 96) and m.shape == (3, 1, 80, 16)
    pred = np.random.default_rng(0).uniform(0, 1, (2, 3, 4, 4)).astype(np.float32)
    u8 = datagen_ref.frames_to_u8(pred)
    assert u8.dtype == np.uint8 and np.array_equal(u8, np.floor(pred.transpose(0, 2, 3, 1) * np.float32(255.)))


# ---------------------------------------------------------------- audio.load_wav's sample-rate conversion (oracle/resample_ref.py)
def test_kaiser_best_filter_known_answers():
    from oracle import resample_ref as R
    w, num_table = R.kaiser_best_filter()
    assert w.shape == (64 * 512 + 1,) and num_table == 512 and w.dtype == np.float64
    assert w[0] == R.ROLLOFF                                   # sinc(0) * kaiser centre (= 1)
    assert abs((2 * w.sum() - w[0]) / num_table - 1.0) < 1e-8  # unit DC gain of the full (two-sided) filter
    assert np.all(np.abs(w[512::512][:10]) < 0.06)             # near its zero crossings one table period apart / rolloff
    tr = R.time_registers(5, 16000 / 44100)
    acc, ref = 0.0, []
    for _ in range(5):
        ref.append(acc)
        acc += 1.0 / (16000 / 44100)
    assert tr.tolist() == ref                                   # repeated addition, not t * increment


def test_resample_oracle_against_the_analytic_signal():
    from oracle import resample_ref as R
    for so in (8000, 48000, 22050, 44100):
        n = int(so * 0.15)
        x = (0.5 * np.sin(2 * np.pi * 440 * np.arange(n) / so)).astype(np.float32)
        y = R.librosa_resample(x, so, 16000)
        assert y.dtype == np.float32 and y.shape[0] == int(np.ceil(n * 16000 / so))
        ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(len(y)) / 16000)
        mid = slice(300, len(y) - 300)
        g = np.dot(y[mid], ref[mid]) / np.dot(ref[mid], ref[mid])
        if so < 16000:      # upsampling: the table is walked in whole steps of 512: exact interpolation of a band-limited signal
            assert abs(g - 1) < 1e-6 and np.abs(y[mid] - ref[mid]).max() < 1e-6
        else:               # downsampling walks the table in steps of int(ratio * 512): the filter is stretched by up to 0.4 %, which
            #                 shows as a passband gain slightly above 1 (resampy's behaviour, restated as is)
            assert 1.0 < g < 512.0 / (so / 16000.0) / int(16000.0 / so * 512) and np.abs(y[mid] - g * ref[mid]).max() < 1e-4
    x = np.random.default_rng(0).standard_normal(500).astype(np.float32)
    assert R.librosa_resample(x, 16000, 16000) is not None and np.array_equal(R.librosa_resample(x, 16000, 16000), x)
    a, b = x[:400], x[100:]
    ya, yb, yab = (R.librosa_resample(v, 44100, 16000) for v in (a, b, (a + b).astype(np.float32)))
    assert np.abs(yab - (ya + yb)).max() < 5e-6                # linear up to float32 rounding
    assert R.librosa_resample(np.ones(1000, np.float32), 44100, 16000).shape[0] == 363   # int(362.8) = 362 samples, fixed to ceil = 363
    with np.testing.assert_raises(ValueError):
        R.resampy_resample(np.ones(2, np.float32), 48000, 16000)


def test_load_wav_decodes_like_soundfile_and_mixes_to_mono(tmp_path):
    """host half of audio.load_wav (no device needed at the native rate): sample-format scaling, channel mean"""
    from scipy.io import wavfile
    from oracle import resample_ref as R
    from wav2lip_amd import audio
    r = np.random.default_rng(1)
    cases = {"i16": r.integers(-32768, 32768, (800, 2), dtype=np.int16), "i32": r.integers(-2 ** 31, 2 ** 31, (800,), dtype=np.int32),
             "u8": r.integers(0, 256, (800, 2), dtype=np.uint8), "f32": r.uniform(-1, 1, (800, 3)).astype(np.float32)}
    for name, data in cases.items():
        path = str(tmp_path / (name + ".wav"))
        wavfile.write(path, 16000, data)
        got = audio.load_wav(path, 16000)
        assert got.dtype == np.float32 and got.shape == (800,)
        assert np.array_equal(got, R.load_wav(path, 16000))
    assert audio.load_wav(str(tmp_path / "i16.wav"), 16000)[0] == np.float32((cases["i16"][0, 0] / 32768.0 + cases["i16"][0, 1] / 32768.0) / 2)
    assert np.array_equal(audio.load_wav(str(tmp_path / "u8.wav"), 16000), ((cases["u8"].astype(np.float32) - 128) / 128).mean(axis=1))
