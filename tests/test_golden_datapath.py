"""CPU side of the reference-executed data-path fixture (tests/golden/golden_datapath_v1.npz, written by
tests/golden/make_golden_datapath.py from the reference's OWN audio.py / inference.py / wav2lip_train.py /
color_syncnet_train.py running with stub librosa / cv2): the oracle restatements and the host logic of the package against it.
Bit-exact where the arithmetic is integer / float64-then-rounded; the GPU side is tests/test_golden_datapath_gpu.py."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import audio_ref, datagen_ref, models_ref
from wav2lip_amd import synthetic as synth

G = np.load(os.path.join(ROOT, "tests", "golden", "golden_datapath_v1.npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def surface(parser):
    rows = []
    for a in parser._actions:
        if a.dest == "help":
            continue
        rows.append([list(a.option_strings), a.dest, getattr(a.type, "__name__", None), a.default, a.nargs, bool(a.required),
                     type(a).__name__])
    return rows


def test_audio_oracle_reproduces_the_reference_module_bit_for_bit():
    """oracle/audio_ref.py vs audio.melspectrogram of the reference (stub librosa = these same stft / mel-basis functions):
    pins hparams use, preemphasis, abs, mel matmul, _amp_to_db, ref-level subtraction, _normalize, op order, dtype"""
    assert np.array_equal(audio_ref.melspectrogram(synth.sine_wav(3.0)), G["mel_sine"])
    noise = synth.noise_wav(int(G["wav_noise_len"]), seed=5)
    assert np.array_equal(audio_ref.melspectrogram(noise), G["mel_noise"])
    assert G["mel_sine"].dtype == np.float32 and float(G["mel_np2_maxdiff"]) < 1e-5


def test_mel_chunking_and_datagen_oracle_match_the_reference_run():
    from wav2lip_amd.inference import mel_chunk_starts
    # index arithmetic: the start columns main() used on noise audio at 30 fps (recovered by unique matching in the generator)
    T_n = int(G["chunk_noise_mel_frames"])
    assert datagen_ref.mel_chunk_starts(T_n, 30.0) == G["chunk_starts_noise_fps30"].tolist()
    assert mel_chunk_starts(T_n, 30.0) == G["chunk_starts_noise_fps30"].tolist()     # the package's host logic
    mel = G["inf_mel"]
    starts = datagen_ref.mel_chunk_starts(mel.shape[1], 25.0)
    assert len(starts) == 72 and mel_chunk_starts(mel.shape[1], 25.0) == starts
    face = G["inf_face"]
    chunks = datagen_ref.mel_chunks(mel, 25.0)
    assert sha(np.stack(chunks)) == str(G["inf_mel_chunks_sha"])
    img, melb = datagen_ref.datagen_batch(np.stack([face] * 32), np.stack(chunks[:32]))
    assert img.dtype == np.float64 and sha(img) == str(G["dg_img_batch0_sha"])
    assert np.array_equal(img[0], G["dg_img_batch0_item0"])
    assert np.array_equal(melb, G["dg_mel_batch0"])
    _, tail = datagen_ref.datagen_batch(np.stack([face] * 8), np.stack(chunks[64:]))
    assert np.array_equal(tail, G["dg_tail_mel_batch"])
    assert G["dg_coords"].tolist() == [0, 96, 0, 96]


def test_oracle_pipeline_reproduces_the_frames_the_reference_main_wrote():
    """datagen -> oracle generator -> uint8 frames for BASELINE configs[0] vs every frame inference.py:main() wrote (same
    batching 32/32/8; the oracle network is pinned bit-for-bit to the reference's by tests/golden/make_golden.py)"""
    torch.set_num_threads(8)
    G_keys = None
    from wav2lip_amd import models
    G_keys = {k: tuple(v.shape) for k, v in models.Wav2Lip().state_dict().items()}
    sd = synth.synthetic_state_dict(G_keys, seed=0)
    mel, face = G["inf_mel"], G["inf_face"]
    chunks = datagen_ref.mel_chunks(mel, 25.0)
    out = []
    for lo in (0, 32, 64):
        n = min(32, 72 - lo)
        img, melb = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(np.stack([face] * n), np.stack(chunks[lo:lo + n])))
        pred = models_ref.wav2lip_forward(sd, torch.from_numpy(melb), torch.from_numpy(img)).numpy()
        out.append(datagen_ref.frames_to_u8(pred))
    frames = np.concatenate(out)
    ref8, ref2 = G["inf_frames_first8"], G["inf_frames_last2"]
    d = np.abs(frames[:8].astype(np.int32) - ref8.astype(np.int32))
    assert int(d.max()) <= 1 and float((d != 0).mean()) <= 1e-3            # same network, same batching: at most rounding-edge bytes
    assert int(np.abs(frames[-2:].astype(np.int32) - ref2.astype(np.int32)).max()) <= 1
    assert np.abs(frames.reshape(72, -1).mean(axis=1) - G["inf_frames_mean"]).max() <= 1e-2


def _clip_frames(seed, n):
    return [synth.face_crops_u8(1, seed=1000 * seed + k)[0] for k in range(n)]


def test_host_sample_arithmetic_matches_the_reference_datasets():
    """wav2lip_amd.train.make_generator_sample / make_syncnet_sample (host index + layout arithmetic) against what the
    reference's Dataset.__getitem__ returned for the same picks: pixels bit-exact, mel windows bit-exact (same oracle mel)"""
    from wav2lip_amd import train
    seeds, nframes = G["ds_clip_seeds"].tolist(), G["ds_clip_frames"].tolist()
    names = [str(n) for n in G["ds_clip_names"]]
    for j in range(3):
        clip, img, wrong = G["gen%d_pick" % j].tolist()
        fr = _clip_frames(seeds[clip], nframes[clip])
        mel_T = G["ds_mel_" + names[clip]].T
        x, indiv, melw, y = train.make_generator_sample(fr[img:img + 5], fr[wrong:wrong + 5], mel_T, img)
        assert sha(x.numpy()) == str(G["gen%d_x_sha" % j]) and sha(y.numpy()) == str(G["gen%d_y_sha" % j])
        assert np.array_equal(x.numpy()[:, :, ::12, ::12], G["gen%d_x_sub" % j])
        assert np.array_equal(indiv.numpy(), G["gen%d_indiv" % j]) and np.array_equal(melw.numpy(), G["gen%d_mel" % j])
    for j in range(4):
        clip, img, wrong, in_sync = G["sync%d_pick" % j].tolist()
        fr = _clip_frames(seeds[clip], nframes[clip])
        mel_T = G["ds_mel_" + names[clip]].T
        chosen = img if in_sync else wrong
        x, melw = train.make_syncnet_sample(fr[chosen:chosen + 5], mel_T, img)
        assert sha(x.numpy()) == str(G["sync%d_x_sha" % j])
        assert np.array_equal(melw.numpy(), G["sync%d_mel" % j])


def test_command_line_surfaces_equal_the_reference_scripts():
    """flags, dests, types, defaults, nargs, required-ness of the four argparse parsers vs the reference's (frozen from its
    own `parser` objects): inference.py:13-51, wav2lip_train.py:19-29, hq_wav2lip_train.py:19-30, color_syncnet_train.py:19-27"""
    from wav2lip_amd import inference, trainer
    assert surface(inference.parser) == json.loads(str(G["cli_inference"]))
    assert surface(trainer.wav2lip_train_parser()) == json.loads(str(G["cli_wav2lip_train"]))
    assert surface(trainer.hq_wav2lip_train_parser()) == json.loads(str(G["cli_hq_wav2lip_train"]))
    assert surface(trainer.color_syncnet_train_parser()) == json.loads(str(G["cli_color_syncnet_train"]))
    a = inference.parser.parse_args(["--checkpoint_path", "c", "--face", "f.mp4", "--audio", "a.wav"])
    assert a.wav2lip_batch_size == 128 and a.pads == [0, 10, 0, 0] and a.box == [-1, -1, -1, -1] and a.static is False


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_fixture_generator_names_the_reference_files_it_executes():
    src = open(os.path.join(ROOT, "tests", "golden", "make_golden_datapath.py")).read()
    for mod in ("import audio as ref_audio", "import inference as ref_inf", '"wav2lip_train"', '"color_syncnet_train"'):
        assert mod in src
