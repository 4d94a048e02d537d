"""Data-path kernels through the C ABI against the oracle: datagen pack (bit-exact), uint8 frames (bit-exact),
layout transposes (bit-exact), melspectrogram (1e-4 abs on the [-4,4] normalised scale), mel window gather
(bit-exact indices)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import audio_ref, datagen_ref
from wav2lip_amd import synthetic as synth
from wav2lip_amd import _lib, audio
from wav2lip_amd._lib import check, ptr

pytestmark = pytest.mark.gpu


def test_datagen_pack_bit_exact(cuda):
    lib = _lib.load()
    faces = synth.face_crops_u8(5, seed=3)
    faces[0, :, :, 0] = np.arange(96 * 96).reshape(96, 96) % 256       # every byte value on both halves
    img, _ = datagen_ref.datagen_batch(faces, synth.mel_windows(5, seed=3))
    ref = img.astype(np.float32)                                        # [5,96,96,6]
    f = torch.from_numpy(faces).to(cuda)
    y = torch.full((5, 96, 96, 8), 9.0, device=cuda)
    check(lib.w2l_datagen_pack(_lib.current_stream(), 5, 96, ptr(f), ptr(y), 8, 8))
    got = y.cpu().numpy()
    assert np.array_equal(got[..., :6], ref) and (got[..., 6:] == 0).all()
    # non-vector path (narrow destination)
    y6 = torch.full((5, 96, 96, 6), 9.0, device=cuda)
    check(lib.w2l_datagen_pack(_lib.current_stream(), 5, 96, ptr(f), ptr(y6), 6, 6))
    assert np.array_equal(y6.cpu().numpy(), ref)


def test_frames_to_u8_truncates_like_numpy(cuda):
    lib = _lib.load()
    rng = np.random.default_rng(0)
    pred = rng.uniform(0, 1, (3, 3, 20, 24)).astype(np.float32)
    pred[0, 0, 0, :8] = [0.0, 1.0, 254.99999 / 255, 255 / 255.0, 0.999999, 1 / 255.0, 0.5, 0.00392156]
    ref = datagen_ref.frames_to_u8(pred)
    x = torch.zeros(3, 20, 24, 4, device=cuda)
    x[..., :3] = torch.from_numpy(pred).permute(0, 2, 3, 1).to(cuda)
    out = torch.zeros(3, 20, 24, 3, dtype=torch.uint8, device=cuda)
    check(lib.w2l_frames_to_u8(_lib.current_stream(), 3, 20, 24, ptr(x), 4, ptr(out)))
    assert np.array_equal(out.cpu().numpy(), ref)


def test_layout_round_trip(cuda):
    lib = _lib.load()
    x = torch.randn(3, 15, 7, 9, device=cuda)
    y = torch.full((3, 7, 9, 16), 5.0, device=cuda)
    s = _lib.current_stream()
    check(lib.w2l_nchw_to_nhwc(s, 3, 15, 7, 9, ptr(x), ptr(y), 16, 16))
    assert torch.equal(y[..., :15], x.permute(0, 2, 3, 1)) and bool((y[..., 15] == 0).all())
    z = torch.empty(3, 15, 7, 9, device=cuda)
    check(lib.w2l_nhwc_to_nchw(s, 3, 15, 7, 9, ptr(y), 16, ptr(z)))
    assert torch.equal(z, x)


@pytest.mark.parametrize("name,wav", [("sine3s", synth.sine_wav()), ("noise1s", synth.noise_wav(16000, seed=7)),
                                      ("noise_ragged", synth.noise_wav(16000 * 2 + 123, seed=8)),
                                      ("short", synth.noise_wav(401, seed=9)),
                                      ("silence", np.zeros(4000, dtype=np.float32))])
def test_melspectrogram_matches_oracle(name, wav, golden, cuda):
    ref = audio_ref.melspectrogram(wav)
    got = audio.melspectrogram(wav)
    assert got.shape == ref.shape == (80, 1 + len(wav) // 200) and got.dtype == np.float32
    err = np.abs(got - ref).max()
    assert err <= 1e-4, err
    if "mel_" + name in golden.files:
        assert np.abs(got - golden["mel_" + name]).max() <= 1e-4
    assert not np.isnan(got).any() and got.min() >= -4 and got.max() <= 4


def test_mel_too_short_is_an_error(cuda):
    with pytest.raises(RuntimeError, match="reflect"):
        audio.melspectrogram(np.zeros(400, dtype=np.float32))


def test_mel_gather_windows_bit_exact(cuda):
    lib = _lib.load()
    wav = synth.sine_wav()
    mel = audio.melspectrogram_device(wav)
    T = mel.shape[1]
    starts = datagen_ref.mel_chunk_starts(T, 25.0)
    st = torch.tensor(starts, dtype=torch.int32, device=cuda)
    out = torch.full((len(starts), 80, 16, 4), 3.0, device=cuda)
    check(lib.w2l_mel_gather(_lib.current_stream(), ptr(mel), T, ptr(st), len(starts), ptr(out), 4, 4))
    ref = np.stack(datagen_ref.mel_chunks(mel.cpu().numpy(), 25.0))
    got = out.cpu().numpy()
    assert np.array_equal(got[..., 0], ref) and (got[..., 1:] == 0).all()


def test_l2norm_and_cosine(cuda):
    from wav2lip_amd.losses import cosine_loss, cosine_similarity
    a = torch.rand(7, 512)
    v = torch.rand(7, 512)
    y = torch.tensor([[1.], [0.], [1.], [1.], [0.], [0.], [1.]])
    cos = cosine_similarity(a.to(cuda), v.to(cuda)).cpu()
    assert (cos - torch.nn.functional.cosine_similarity(a, v)).abs().max() <= 1e-6
    loss = cosine_loss(a.to(cuda), v.to(cuda), y.to(cuda)).item()
    ref = torch.nn.functional.binary_cross_entropy(torch.nn.functional.cosine_similarity(a, v).unsqueeze(1), y).item()
    assert abs(loss - ref) <= 1e-5
    lib = _lib.load()
    x = torch.rand(5, 512, device=cuda)
    out = torch.empty(5, 512, device=cuda)
    check(lib.w2l_l2norm_rows(_lib.current_stream(), 5, 512, ptr(x), 512, ptr(out)))
    assert (out.cpu() - torch.nn.functional.normalize(x.cpu(), p=2, dim=1)).abs().max() <= 1e-6


# ---------------------------------------------------------------- crop/resize and resize/paste
def _boxes_for(H, W):
    return [(0, 96, 0, 96), (10, 106, 20, 116), (0, 192, 0, 192), (5, 197, 30, 222), (0, H, 0, W), (7, 50, 3, 61),
            (100, 229, 40, 187), (H - 97, H, W - 131, W), (0, 1, 0, 1), (3, 5, 200, 203), (50, 146, 60, 252)]


def test_crop_resize_matches_cv2_semantics_oracle(cuda):
    from oracle import resize_ref
    lib = _lib.load()
    rng = np.random.default_rng(5)
    H, W = 240, 320
    frames = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    boxes = _boxes_for(H, W)
    idx = [i % 3 for i in range(len(boxes))]
    f = torch.from_numpy(frames).to(cuda)
    bd = torch.tensor(boxes, dtype=torch.int32, device=cuda)
    idd = torch.tensor(idx, dtype=torch.int32, device=cuda)
    out = torch.zeros((len(boxes), 96, 96, 3), dtype=torch.uint8, device=cuda)
    check(lib.w2l_crop_resize_u8(_lib.current_stream(), len(boxes), ptr(f), H, W, ptr(idd), ptr(bd), 96, ptr(out)))
    got = out.cpu().numpy()
    for j, (b, i) in enumerate(zip(boxes, idx)):
        ref = resize_ref.crop_resize(frames[i], b)
        assert np.array_equal(got[j], ref), "box %s: %d bytes differ" % (b, int((got[j] != ref).sum()))


def test_resize_paste_matches_cv2_semantics_oracle(cuda):
    from oracle import resize_ref
    lib = _lib.load()
    rng = np.random.default_rng(6)
    H, W = 240, 320
    boxes = _boxes_for(H, W)
    n = len(boxes)
    frames = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    pred = rng.integers(0, 256, (n, 96, 96, 3), dtype=np.uint8)
    f = torch.from_numpy(frames).to(cuda)
    p = torch.from_numpy(pred).to(cuda)
    bd = torch.tensor(boxes, dtype=torch.int32, device=cuda)
    max_px = max((b[1] - b[0]) * (b[3] - b[2]) for b in boxes)
    check(lib.w2l_resize_paste_u8(_lib.current_stream(), n, ptr(p), 96, ptr(bd), None, ptr(f), H, W, max_px))
    got = f.cpu().numpy()
    for j, b in enumerate(boxes):
        ref = resize_ref.resize_paste(frames[j].copy(), pred[j], b)
        assert np.array_equal(got[j], ref), "box %s: %d bytes differ" % (b, int((got[j] != ref).sum()))


def test_run_frames_end_to_end_against_the_cpu_pipeline(cuda):
    """frames in, frames out (inference.py:121-126 + 259-271) against the oracle pipeline with the oracle generator"""
    from oracle import datagen_ref, models_ref, resize_ref
    from wav2lip_amd import models
    from wav2lip_amd.inference import Wav2LipRunner
    G = models.Wav2Lip()
    sd = synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0)
    G.load_state_dict(sd)
    G = G.to(cuda).eval()
    rng = np.random.default_rng(9)
    H, W = 160, 200
    frames = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    boxes = [(10, 150, 30, 170), (0, 96, 100, 196), (20, 84, 5, 69)]
    idx = [0, 1, 1]
    mels = synth.mel_windows(3, seed=2)
    runner = Wav2LipRunner(G, batch_size=3)
    out = runner.run_frames(torch.from_numpy(frames).to(cuda), idx, boxes, mel_windows=torch.from_numpy(mels).to(cuda))
    got = out.cpu().numpy()
    faces = np.stack([resize_ref.crop_resize(frames[i], b) for i, b in zip(idx, boxes)])
    img, mel = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(faces, mels))
    pred = datagen_ref.frames_to_u8(models_ref.wav2lip_forward(sd, torch.from_numpy(mel), torch.from_numpy(img)).numpy())
    for j, (i, b) in enumerate(zip(idx, boxes)):
        ref = resize_ref.resize_paste(frames[i].copy(), pred[j], b)
        diff = np.abs(got[j].astype(np.int32) - ref.astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).mean() <= 1e-3, (diff.max(), (diff != 0).mean())
        y1, y2, x1, x2 = b
        mask = np.ones((H, W), bool)
        mask[y1:y2, x1:x2] = False
        assert np.array_equal(got[j][mask], frames[i][mask])          # outside the box the frame is untouched
    with pytest.raises(ValueError):
        runner.run_frames(torch.from_numpy(frames).to(cuda), [0], [(0, 300, 0, 50)], mel_windows=torch.from_numpy(mels[:1]).to(cuda))


def test_mel_bank_windows_match_the_dataset_arithmetic(cuda):
    """per-clip mel computed once on the device, batches gathered by the reference's index expression"""
    from wav2lip_amd import train
    wavs = [synth.noise_wav(16000 * 3 + 77, seed=21), synth.sine_wav(2.0), synth.noise_wav(16000 * 4, seed=22)]
    bank = train.MelBank(cuda)
    ids = [bank.add(w) for w in wavs]
    assert ids == [0, 1, 2]
    refs = [audio_ref.melspectrogram(w).T for w in wavs]               # (T, 80) as the Dataset holds it
    clips, frames = [0, 2, 1, 0], [3, 40, 7, 1]
    mel = bank.windows(clips, frames).cpu().numpy()
    seg = bank.segmented(clips, frames).cpu().numpy()
    assert mel.shape == (4, 1, 80, 16) and seg.shape == (4, 5, 1, 80, 16)
    for j, (c, f) in enumerate(zip(clips, frames)):
        assert np.abs(mel[j, 0] - train.crop_audio_window(refs[c], f).T).max() <= 1e-4
        assert np.abs(seg[j, :, 0] - train.get_segmented_mels(refs[c], f)).max() <= 1e-4
    with pytest.raises(ValueError):
        bank.windows([1], [60])          # 2 s clip: frame 60 starts at column 192 > T - 16
    with pytest.raises(ValueError):
        bank.segmented([0], [0])


def test_lipsync_config1_end_to_end(cuda):
    """BASELINE configs[0]: one static 96x96 image + 3 s of 16 kHz sine -> 72 frames (inference.py:main for in-memory
    inputs), batch 16, pipelined lanes, against the CPU pipeline (oracle mel -> chunking -> datagen -> generator -> uint8)"""
    from oracle import datagen_ref, models_ref
    from wav2lip_amd import models
    from wav2lip_amd.inference import lipsync
    G = models.Wav2Lip()
    sd = synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0)
    G.load_state_dict(sd)
    G = G.to(cuda).eval()
    face = synth.face_crops_u8(1, seed=77)[0]
    wav = synth.sine_wav(3.0)
    out = lipsync(G, [face], wav, fps=25., batch_size=16, static=True)
    mel = audio_ref.melspectrogram(wav)
    starts = datagen_ref.mel_chunk_starts(mel.shape[1], 25.)
    assert len(out) == len(starts) == 72
    mw = np.stack([mel[:, s:s + 16] for s in starts])
    img, m = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(np.stack([face] * 72), mw))
    ref = datagen_ref.frames_to_u8(models_ref.wav2lip_forward(sd, torch.from_numpy(m), torch.from_numpy(img)).numpy())
    got = np.stack(out)
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert got.shape == ref.shape and diff.max() <= 1 and (diff != 0).mean() <= 1e-3, (diff.max(), (diff != 0).mean())


def test_pipelined_runner_equals_serial_runner(cuda):
    from wav2lip_amd import models
    from wav2lip_amd.inference import PipelinedRunner, Wav2LipRunner
    G = models.Wav2Lip()
    G.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0))
    G = G.to(cuda).eval()
    sizes = [8, 8, 8, 5, 8]
    faces = [torch.from_numpy(synth.face_crops_u8(n, seed=200 + i)).to(cuda) for i, n in enumerate(sizes)]
    mels = [torch.from_numpy(synth.mel_windows(n, seed=300 + i)).to(cuda) for i, n in enumerate(sizes)]
    serial = Wav2LipRunner(G, batch_size=8)
    want = [serial.run_batch(f, mel_windows=m).clone() for f, m in zip(faces, mels)]
    pipe = PipelinedRunner(G, batch_size=8, depth=2)
    got, pending = [], None
    for f, m in zip(faces, mels):
        t = pipe.submit(f, mel_windows=m)
        if pending is not None:
            got.append(pipe.result(pending).clone())
        pending = t
    got.append(pipe.result(pending).clone())
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert a.shape == b.shape
        d = (a.int() - b.int()).abs()
        assert int(d.max()) <= 1 and float((d != 0).float().mean()) <= 1e-3     # lanes autotune independently


# ---------------------------------------------------------------- audio.load_wav: sample-rate conversion (w2l_resample_sinc)
@pytest.mark.gpu
@pytest.mark.parametrize("orig_sr,n", [(48000, 4801), (44100, 4410), (8000, 799), (22050, 2300), (11025, 1200), (96000, 9000),
                                       (16001, 1700)])
def test_resample_is_bit_exact_with_the_resampy_restatement(orig_sr, n, cuda):
    """same term order, same per-term float32 rounding as the reference's numba loop: equal bit patterns, edges included"""
    from oracle import resample_ref
    x = synth.noise_wav(n, seed=orig_sr % 97)
    got = audio.resample(x, orig_sr, 16000)
    ref = resample_ref.librosa_resample(x, orig_sr, 16000)
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert audio.resample(x, 16000, 16000) is not None and np.array_equal(audio.resample(x, 16000, 16000), x)


@pytest.mark.gpu
def test_load_wav_resamples_a_44k1_stereo_file_and_feeds_the_mel_path(cuda, tmp_path):
    from scipy.io import wavfile
    from oracle import resample_ref
    r = np.random.default_rng(5)
    data = (r.standard_normal((6000, 2)) * 6000).astype(np.int16)
    path = str(tmp_path / "clip44k.wav")
    wavfile.write(path, 44100, data)
    got = audio.load_wav(path, 16000)
    ref = resample_ref.load_wav(path, 16000)
    assert got.shape == ref.shape == (int(np.ceil(6000 * 16000 / 44100)),)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    mel = audio.melspectrogram(got)
    assert mel.shape == (80, 1 + got.shape[0] // 200) and np.abs(mel - audio_ref.melspectrogram(ref)).max() <= 1e-3
    with pytest.raises(ValueError, match="too small"):
        audio.resample(np.ones(2, np.float32), 48000, 16000)


@pytest.mark.gpu
def test_resample_full_length_clip_against_the_analytic_signal(cuda):
    """size-independent property at a realistic size (60 s at 48 kHz, 2.9 M samples): a band-limited sine comes out as the same
    sine at the new rate, up to the documented passband gain of the stretched table walk"""
    so, secs = 48000, 60
    x = (0.5 * np.sin(2 * np.pi * 440 * np.arange(so * secs) / so)).astype(np.float32)
    y = audio.resample(x, so, 16000)
    assert y.shape == (16000 * secs,)
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(len(y)) / 16000)
    mid = slice(1000, len(y) - 1000)
    g = np.dot(y[mid].astype(np.float64), ref[mid]) / np.dot(ref[mid], ref[mid])
    assert 1.0 < g < 512.0 / 3.0 / 170 and np.abs(y[mid] - g * ref[mid]).max() < 1e-4


# ---------------------------------------------------------------- device-resident training set (wav2lip_amd/data.py)
def _toy_clips():
    """three clips as the reference's Dataset sees them: (frames BGR uint8, frame ids, wav)"""
    r = np.random.default_rng(42)
    ids0 = [i for i in range(24) if i != 10]                      # a missing 10.jpg: windows across the gap are rejected
    clip0 = ([r.integers(0, 256, (96, 96, 3), dtype=np.uint8) for _ in ids0], ids0, synth.noise_wav(16000, seed=1))
    ids1 = list(range(30))                                        # crops that need cv2.resize; audio ends before the video does
    clip1 = ([r.integers(0, 256, (120, 100, 3), dtype=np.uint8) for _ in ids1], ids1, synth.noise_wav(int(16000 * 0.9), seed=2))
    ids2 = list(range(12))                                        # <= 3 * syncnet_T frames: never sampled
    clip2 = ([r.integers(0, 256, (96, 96, 3), dtype=np.uint8) for _ in ids2], ids2, synth.noise_wav(16000, seed=3))
    return [clip0, clip1, clip2]


@pytest.mark.gpu
def test_clip_store_batches_equal_the_reference_dataset_arithmetic(cuda):
    import random
    from oracle import resize_ref
    from wav2lip_amd import data, train
    clips = _toy_clips()
    store = data.ClipStore(cuda)
    for frames, ids, wav in clips:
        store.add_clip(frames, ids, wav)
    assert len(store) == 3 and store.frames().shape == (23 + 30 + 12, 96, 96, 3)
    host = []                                                      # what cv2.imread + cv2.resize would hand the reference
    for frames, ids, wav in clips:
        host.append(({i: (f if f.shape[:2] == (96, 96) else resize_ref.resize_linear_u8(f, (96, 96))) for i, f in zip(ids, frames)},
                     audio_ref.melspectrogram(wav).T))
    B = 12
    x, indiv, mel, y, picks = store.sample_generator_batch(B, random.Random(7))
    assert x.shape == (B, 6, 5, 96, 96) and indiv.shape == (B, 5, 1, 80, 16) and mel.shape == (B, 1, 80, 16) and y.shape == (B, 3, 5, 96, 96)
    assert {p[0] for p in picks} <= {0, 1} and len({p[0] for p in picks}) == 2
    for b, (clip, img, wrong) in enumerate(picks):
        fr, mel_T = host[clip]
        assert all(img + t in fr and wrong + t in fr for t in range(5)) and img != wrong       # complete windows only
        ref = train.make_generator_sample([fr[img + t] for t in range(5)], [fr[wrong + t] for t in range(5)], mel_T, img, 25)
        assert ref is not None                                                                # audio windows inside the clip
        rx, rindiv, rmel, ry = ref
        assert torch.equal(x[b].cpu(), rx) and torch.equal(y[b].cpu(), ry)                     # bit-exact pixels (u8 / 255., mask)
        assert (mel[b].cpu() - rmel).abs().max() <= 1e-3 and (indiv[b].cpu() - rindiv).abs().max() <= 1e-3
    xs, mels, ys, spicks = store.sample_syncnet_batch(B, random.Random(11))
    assert xs.shape == (B, 15, 48, 96) and mels.shape == (B, 1, 80, 16) and ys.shape == (B, 1)
    assert 0 < float(ys.sum()) < B                                                            # both labels occur
    for b, (clip, img, wrong, in_sync) in enumerate(spicks):
        fr, mel_T = host[clip]
        chosen = img if in_sync else wrong
        rx, rmel = train.make_syncnet_sample([fr[chosen + t] for t in range(5)], mel_T, img, 25)
        assert torch.equal(xs[b].cpu(), rx) and float(ys[b]) == (1.0 if in_sync else 0.0)
        assert (mels[b].cpu() - rmel).abs().max() <= 1e-3                                    # the TRUE frame's audio either way


@pytest.mark.gpu
def test_clip_store_reads_the_preprocessed_directory_layout(cuda, tmp_path, monkeypatch):
    from PIL import Image
    from scipy.io import wavfile
    from wav2lip_amd import data
    r = np.random.default_rng(0)
    root = tmp_path / "lrs2_preprocessed"
    (tmp_path / "filelists").mkdir()
    (tmp_path / "filelists" / "train.txt").write_text("spk/00001 extra-column\nspk/00002\n")
    truth = {}
    for vid, n in (("spk/00001", 16), ("spk/00002", 17)):
        d = root / vid
        d.mkdir(parents=True)
        for i in range(n):
            img = r.integers(0, 256, (96, 96, 3), dtype=np.uint8)
            Image.fromarray(img).save(str(d / ("%d.jpg" % i)), quality=95)
            truth[(vid, i)] = np.asarray(Image.open(str(d / ("%d.jpg" % i))).convert("RGB"))[:, :, ::-1]
        wavfile.write(str(d / "audio.wav"), 16000, (synth.noise_wav(16000, seed=n) * 20000).astype(np.int16))
    monkeypatch.chdir(tmp_path)                                     # the reference opens 'filelists/<split>.txt' relative to the cwd
    store = data.ClipStore.from_directory(str(root), "train", cuda)
    assert store.names == ["spk/00001", "spk/00002"] and store.frame_ids[1] == list(range(17))   # numeric order, not '10' < '2'
    fr = store.frames().cpu().numpy()
    assert np.array_equal(fr[store.row_of[0][5]], truth[("spk/00001", 5)]) and np.array_equal(fr[store.row_of[1][16]], truth[("spk/00002", 16)])
    x, indiv, mel, y, picks = store.sample_generator_batch(4)
    assert x.shape == (4, 6, 5, 96, 96) and all(c in (0, 1) for c, _, _ in picks)
