"""Generates tests/golden/golden_datapath_v1.npz by EXECUTING the reference's own data-path source - `audio.py`,
`inference.py` (`datagen`, the mel chunking and the whole of `main()`), `wav2lip_train.py` and `color_syncnet_train.py`
(`Dataset.__getitem__` and its helper methods) - in the build container.  Runs only where /root/reference exists.

    python tests/golden/make_golden_datapath.py

The third-party modules those files import but the image lacks are replaced by stubs that hold NO arithmetic of the path:

  librosa   `core.load` -> the PCM16 reader, `stft` / `filters.mel` -> the restatements in oracle/audio_ref.py, with the
            arguments the reference passes asserted (audio.py:10,61,100).  What this pins: the reference's use of hparams,
            preemphasis (scipy), np.abs, the mel matmul, _amp_to_db, the ref_level_db subtraction, _normalize, the op order
            and numpy's casting - everything of audio.melspectrogram except librosa's own two functions.
  cv2       `imread` -> frames of an in-memory table, `resize` -> identity (asserted: every face here is already 96x96),
            `VideoWriter` -> a frame collector.  What this pins: datagen, the batch loop, the uint8 post-processing and the
            paste-back of inference.py, and the Dataset window / masking / layout arithmetic of the training scripts.
  face_detection -> empty (the run uses --box, inference.py:111-114); subprocess.call (ffmpeg mux) -> no-op.

One numpy-version shim: the reference pins numpy==1.17.1 (requirements.txt:2), whose value-based casting keeps
`np.maximum(min_level, x)` in audio.py:105 float32 (`min_level` is an np.float64 SCALAR, `x` a float32 array); numpy 2 (NEP 50)
promotes that expression to float64.  `audio.np.exp` is therefore wrapped to return a Python float for scalar input - a weak
scalar under NEP 50, i.e. exactly the pinned version's promotion - so that the module computes what it computes under its own
requirements.  The largest difference to the un-shimmed (float64) evaluation is stored as `mel_np2_maxdiff` (~1e-6).

Frozen: the mel spectrograms, what `datagen` yields, every frame `main()` writes for BASELINE configs[0] (one static 96x96
image + a 3 s 16 kHz sine, synthetic seeded weights in a reference-format checkpoint), and generator / SyncNet training
samples with the random picks that produced them.  Bit-exact tensors are stored as SHA-256 + a strided subsample, float
tensors that the HIP path reproduces only to a tolerance are stored whole.
"""
import hashlib
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import audio_ref  # noqa: E402
from wav2lip_amd import synthetic as synth  # noqa: E402

IMAGES = {}        # path -> uint8 BGR array served by the cv2.imread stub
WAVS = {}          # path -> float32 samples served by the librosa.core.load stub


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def clip_frame(clip_seed, frame_id):
    """deterministic 96x96 BGR frame of a synthetic training clip"""
    return synth.face_crops_u8(1, seed=1000 * clip_seed + frame_id)[0]


def parser_surface(parser):
    """[(flags, dest, type name, default, nargs, required, store-true?)] of an argparse parser, help action excluded"""
    rows = []
    for a in parser._actions:
        if a.dest == "help":
            continue
        rows.append([list(a.option_strings), a.dest, getattr(a.type, "__name__", None), a.default, a.nargs, bool(a.required),
                     type(a).__name__])
    return rows


def install_stubs(written):
    # ---- librosa
    librosa = types.ModuleType("librosa")
    core = types.ModuleType("librosa.core")
    filters = types.ModuleType("librosa.filters")

    def load(path, sr=22050):
        assert sr == 16000
        if path in WAVS:
            return WAVS[path], sr
        return audio_ref.load_wav_pcm16(path), sr

    def stft(y=None, n_fft=2048, hop_length=None, win_length=None):
        assert (n_fft, hop_length, win_length) == (800, 200, 800), (n_fft, hop_length, win_length)
        return audio_ref.stft(y)

    def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
        assert (sr, n_fft, n_mels, fmin, fmax) == (16000, 800, 80, 55, 7600), (sr, n_fft, n_mels, fmin, fmax)
        return audio_ref.mel_basis()

    core.load, librosa.stft, filters.mel = load, stft, mel
    librosa.core, librosa.filters = core, filters
    sys.modules.update({"librosa": librosa, "librosa.core": core, "librosa.filters": filters})
    # ---- cv2
    cv2 = types.ModuleType("cv2")

    def imread(path):
        img = IMAGES.get(path)
        return None if img is None else img.copy()

    def resize(img, dsize):
        assert tuple(dsize) == (img.shape[1], img.shape[0]), "this fixture only feeds faces that are already at the target size"
        return img

    class VideoWriter:
        def __init__(self, path, fourcc, fps, size):
            written["fps"], written["size"], written["frames"] = fps, size, []

        def write(self, f):
            written["frames"].append(f.copy())

        def release(self):
            pass

    cv2.imread, cv2.resize, cv2.VideoWriter = imread, resize, VideoWriter
    cv2.VideoWriter_fourcc = lambda *a: 0
    sys.modules["cv2"] = cv2
    sys.modules["face_detection"] = types.ModuleType("face_detection")


def main():
    assert os.path.isdir(REF), "the reference checkout is needed to generate this fixture"
    torch.set_num_threads(8)
    torch.manual_seed(0)
    written = {}
    install_stubs(written)
    sys.path.insert(0, REF)
    out = {}
    tmp = tempfile.mkdtemp(prefix="w2l_golden_")
    os.chdir(tmp)

    # ---------------------------------------------------------------- audio.py
    import audio as ref_audio
    assert ref_audio.__file__.startswith(REF)
    sine = synth.sine_wav(3.0)
    noise = synth.noise_wav(16000 * 2 + 123, seed=5)
    m_np2 = ref_audio.melspectrogram(noise)                 # un-shimmed: numpy 2 promotion (float64 from _amp_to_db on)
    assert m_np2.dtype == np.float64

    class _Numpy117(types.ModuleType):                      # see the module docstring: numpy==1.17.1 promotion at audio.py:104-105
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def exp(x):
            r = np.exp(x)
            return float(r) if np.ndim(r) == 0 else r
    ref_audio.np = _Numpy117("numpy")
    out["mel_np2_maxdiff"] = np.float64(np.abs(m_np2 - ref_audio.melspectrogram(noise)).max())
    for tag, wav in (("sine", sine), ("noise", noise)):
        m = ref_audio.melspectrogram(wav)
        assert m.dtype == np.float32 and m.shape == (80, 1 + len(wav) // 200), (m.dtype, m.shape)
        mo = audio_ref.melspectrogram(wav)
        assert np.array_equal(m, mo), "oracle/audio_ref.py does not reproduce audio.melspectrogram (stub librosa) bit for bit"
        out["mel_" + tag] = m
    out["wav_noise_len"] = np.int64(len(noise))

    # ---------------------------------------------------------------- inference.py: configs[0] end to end
    from scipy.io import wavfile
    face = synth.face_crops_u8(1, seed=3)[0]
    IMAGES["face.png"] = face
    open("face.png", "wb").close()
    wavfile.write("sine.wav", 16000, np.clip(np.round(sine * 32768.0), -32768, 32767).astype(np.int16))
    sys.path.insert(0, ROOT)
    G_keys = None
    import models as ref_models
    net = ref_models.Wav2Lip()
    G_keys = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synthetic_state_dict(G_keys, seed=0)
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}, "optimizer": None, "global_step": 7, "global_epoch": 1},
               "ckpt.pth")
    sys.argv = ["inference.py", "--checkpoint_path", "ckpt.pth", "--face", "face.png", "--audio", "sine.wav",
                "--box", "0", "96", "0", "96", "--wav2lip_batch_size", "32", "--outfile", "out.mp4"]
    import subprocess
    real_call = subprocess.call
    subprocess.call = lambda *a, **k: 0
    import inference as ref_inf
    assert ref_inf.__file__.startswith(REF) and ref_inf.args.static is True
    # everything main() hands to datagen / gets back from it is recorded by a pass-through wrapper around the REFERENCE's datagen
    seen = {}
    real_datagen = ref_inf.datagen

    def recording_datagen(frames, mels):
        seen["mels"] = [np.array(m) for m in mels]
        seen["batches"] = []
        for item in real_datagen(frames, mels):
            seen["batches"].append((item[0].copy(), item[1].copy(), [f.copy() for f in item[2]], list(item[3])))
            yield item
    ref_inf.datagen = recording_datagen
    ref_inf.main()
    import json
    out["cli_inference"] = np.array(json.dumps(parser_surface(ref_inf.parser)))
    frames = np.stack(written["frames"])
    assert frames.shape == (72, 96, 96, 3) and written["fps"] == 25.0 and written["size"] == (96, 96)
    out["inf_face"] = face
    out["inf_frames_sha"] = np.array(sha(frames))
    out["inf_frames_first8"] = frames[:8]
    out["inf_frames_last2"] = frames[-2:]
    out["inf_frames_mean"] = frames.reshape(72, -1).mean(axis=1)
    chunks = seen["mels"]
    assert len(chunks) == 72 and [len(b[0]) for b in seen["batches"]] == [32, 32, 8]
    mel = ref_audio.melspectrogram(ref_audio.load_wav("sine.wav", 16000))     # what main() computed (audio.py on the PCM16 file)
    out["inf_mel"] = mel
    out["inf_mel_chunks_sha"] = np.array(sha(np.stack(chunks)))
    assert np.array_equal(chunks[0], mel[:, 0:16]) and np.array_equal(chunks[-1], mel[:, -16:])
    img_batch, mel_batch, frame_batch, coords_batch = seen["batches"][0]
    assert img_batch.dtype == np.float64 and img_batch.shape == (32, 96, 96, 6) and mel_batch.shape == (32, 80, 16, 1)
    out["dg_img_batch0_sha"] = np.array(sha(img_batch))
    out["dg_img_batch0_item0"] = img_batch[0]
    out["dg_mel_batch0"] = mel_batch
    out["dg_tail_mel_batch"] = seen["batches"][2][1]
    out["dg_coords"] = np.array(coords_batch[0])
    assert all(np.array_equal(f, face) for f in frame_batch)
    # A second run of main() on NOISE audio at 30 fps (every spectrogram column distinct, a non-integer 80/fps): the start
    # column of every chunk main() cut is recovered by exact, unique matching - pins the index arithmetic of inference.py:231-240
    noise_a = synth.noise_wav(16000 * 2 + 777, seed=9)
    wavfile.write("noise.wav", 16000, np.clip(np.round(noise_a * 32768.0), -32768, 32767).astype(np.int16))
    ref_inf.args.audio, ref_inf.args.fps = "noise.wav", 30.0
    ref_inf.main()
    mel_n = ref_audio.melspectrogram(ref_audio.load_wav("noise.wav", 16000))
    starts = []
    for c in seen["mels"]:
        hits = [s_ for s_ in range(mel_n.shape[1] - 15) if np.array_equal(mel_n[:, s_:s_ + 16], c)]
        assert len(hits) == 1, hits
        starts.append(hits[0])
    out["chunk_starts_noise_fps30"] = np.array(starts, dtype=np.int64)
    out["chunk_noise_mel_frames"] = np.int64(mel_n.shape[1])
    out["inf_noise_frames_mean"] = np.stack(written["frames"]).reshape(len(starts), -1).mean(axis=1)
    subprocess.call = real_call
    ref_inf.datagen = real_datagen

    # ---------------------------------------------------------------- wav2lip_train.py / color_syncnet_train.py Datasets
    root = os.path.join(tmp, "data")
    clips = {"clipA": (1, 40), "clipB": (2, 30), "short": (3, 10)}      # name -> (seed, number of frames); `short` is rejected
    os.makedirs("filelists", exist_ok=True)
    with open("filelists/train.txt", "w") as fh:
        for name in clips:
            fh.write(name + " 0\n")
    for name, (seed, n) in clips.items():
        d = os.path.join(root, name)
        os.makedirs(d)
        for k in range(n):
            p = os.path.join(d, "%d.jpg" % k)
            open(p, "wb").close()
            IMAGES[p] = clip_frame(seed, k)
        WAVS[os.path.join(d, "audio.wav")] = synth.noise_wav(int(16000 * n / 25.0), seed=50 + seed)
    out["ds_clip_names"] = np.array(list(clips))
    out["ds_clip_seeds"] = np.array([v[0] for v in clips.values()])
    out["ds_clip_frames"] = np.array([v[1] for v in clips.values()])
    for name, (seed, n) in clips.items():
        out["ds_mel_" + name] = ref_audio.melspectrogram(WAVS[os.path.join(root, name, "audio.wav")])

    def run_dataset(modname, argv, nsamples, tag):
        sys.argv = argv
        mod = __import__(modname)
        assert mod.__file__.startswith(REF)
        out["cli_" + modname] = np.array(json.dumps(parser_surface(mod.parser)))
        ds = mod.Dataset("train")
        log = []
        real_choice = random.choice

        def logged_choice(seq):
            v = real_choice(seq)
            log.append(v)
            return v
        mod.random.choice = logged_choice
        random.seed(1234)
        samples = []
        for _ in range(nsamples):
            mark = len(log)
            item = ds[0]
            samples.append((item, log[mark:]))
        mod.random.choice = real_choice
        return samples

    gen_samples = run_dataset("wav2lip_train", ["wav2lip_train.py", "--data_root", root, "--checkpoint_dir", tmp,
                                                "--syncnet_checkpoint_path", "none"], 3, "gen")
    for j, ((x, indiv, melw, y), picks) in enumerate(gen_samples):
        names = [p for p in picks if isinstance(p, str)]
        img_name, wrong_name = names[-2], names[-1]              # the accepted draw is the last one of the call
        clip = os.path.basename(os.path.dirname(img_name))
        assert os.path.dirname(img_name) == os.path.dirname(wrong_name)
        out["gen%d_pick" % j] = np.array([list(clips).index(clip), int(os.path.basename(img_name).split(".")[0]),
                                          int(os.path.basename(wrong_name).split(".")[0])])
        assert x.shape == (6, 5, 96, 96) and indiv.shape == (5, 1, 80, 16) and melw.shape == (1, 80, 16) and y.shape == (3, 5, 96, 96)
        out["gen%d_x_sha" % j] = np.array(sha(x.numpy()))
        out["gen%d_y_sha" % j] = np.array(sha(y.numpy()))
        out["gen%d_x_sub" % j] = x.numpy()[:, :, ::12, ::12]
        out["gen%d_indiv" % j] = indiv.numpy()
        out["gen%d_mel" % j] = melw.numpy()
    sys.argv = ["hq_wav2lip_train.py", "--data_root", root, "--checkpoint_dir", tmp, "--syncnet_checkpoint_path", "none"]
    import hq_wav2lip_train as ref_hq
    out["cli_hq_wav2lip_train"] = np.array(json.dumps(parser_surface(ref_hq.parser)))
    sync_samples = run_dataset("color_syncnet_train", ["color_syncnet_train.py", "--data_root", root, "--checkpoint_dir", tmp],
                               4, "sync")
    for j, ((x, melw, y), picks) in enumerate(sync_samples):
        names = [p for p in picks if isinstance(p, str)]
        flags = [p for p in picks if isinstance(p, bool)]
        img_name, wrong_name, in_sync = names[-2], names[-1], flags[-1]
        clip = os.path.basename(os.path.dirname(img_name))
        out["sync%d_pick" % j] = np.array([list(clips).index(clip), int(os.path.basename(img_name).split(".")[0]),
                                           int(os.path.basename(wrong_name).split(".")[0]), int(in_sync)])
        assert x.shape == (15, 48, 96) and melw.shape == (1, 80, 16) and float(y) == float(in_sync)
        out["sync%d_x_sha" % j] = np.array(sha(x.numpy()))
        out["sync%d_x_sub" % j] = x.numpy()[:, ::8, ::8]
        out["sync%d_mel" % j] = melw.numpy()

    path = os.path.join(HERE, "golden_datapath_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %.1f kB" % (path, len(out), os.path.getsize(path) / 1e3))
    for k in sorted(out):
        if k.endswith("_pick") or k.endswith("starts"):
            print(k, out[k].tolist() if out[k].size < 12 else out[k][:12].tolist())


if __name__ == "__main__":
    main()
