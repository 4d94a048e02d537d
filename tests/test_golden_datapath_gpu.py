"""GPU side of the reference-executed data-path fixture (tests/golden/golden_datapath_v1.npz: the reference's own audio.py /
inference.py / Dataset code run with stub librosa / cv2, tests/golden/make_golden_datapath.py): the HIP path - mel kernel,
`inference.main()` end to end, the reference-signature `datagen`, the device-resident training store - against what the
reference produced.  Tolerances: mel 1e-4 on the [-4, 4] scale (f64 DFT on the device vs numpy's FFT); uint8 frames at most
1 LSB off in at most 0.1 % of the bytes (truncation of v*255 at rounding edges); pixels of training samples bit-exact."""
import hashlib
import os
import random

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import datagen_ref, models_ref, resize_ref
from wav2lip_amd import synthetic as synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(ROOT, "tests", "golden", "golden_datapath_v1.npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _gen_state_dict():
    from wav2lip_amd import models
    return synth.synthetic_state_dict({k: tuple(v.shape) for k, v in models.Wav2Lip().state_dict().items()}, seed=0)


def _write_inputs(tmp, face_bgr, wav):
    from PIL import Image
    from scipy.io import wavfile
    Image.fromarray(np.ascontiguousarray(face_bgr[:, :, ::-1])).save(os.path.join(tmp, "face.png"))      # PNG: lossless
    wavfile.write(os.path.join(tmp, "audio.wav"), 16000, np.clip(np.round(wav * 32768.0), -32768, 32767).astype(np.int16))
    sd = _gen_state_dict()
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}, "optimizer": None, "global_step": 7, "global_epoch": 1},
               os.path.join(tmp, "ckpt.pth"))
    return sd


def test_hip_mel_matches_the_reference_audio_module(cuda):
    from wav2lip_amd import audio
    for tag, wav in (("sine", synth.sine_wav(3.0)), ("noise", synth.noise_wav(int(G["wav_noise_len"]), seed=5))):
        got = audio.melspectrogram(wav)
        assert got.shape == G["mel_" + tag].shape and got.dtype == np.float32
        assert float(np.abs(got - G["mel_" + tag]).max()) <= 1e-4, tag


def test_inference_main_reproduces_the_frames_the_reference_main_wrote(cuda, tmp_path):
    """BASELINE configs[0] through the command-line entry point (inference.py:181-277): one static 96x96 image + a 3 s 16 kHz
    sine WAV, --box, batch 32, a `module.`-prefixed reference-format checkpoint"""
    from wav2lip_amd import container, inference
    tmp = str(tmp_path)
    _write_inputs(tmp, G["inf_face"], synth.sine_wav(3.0))
    out = os.path.join(tmp, "results", "out.avi")
    frames = inference.main(["--checkpoint_path", os.path.join(tmp, "ckpt.pth"), "--face", os.path.join(tmp, "face.png"),
                             "--audio", os.path.join(tmp, "audio.wav"), "--box", "0", "96", "0", "96",
                             "--wav2lip_batch_size", "32", "--outfile", out])
    frames = np.stack(frames)
    assert frames.shape == (72, 96, 96, 3) and inference.args.static is True
    for got, ref in ((frames[:8], G["inf_frames_first8"]), (frames[-2:], G["inf_frames_last2"])):
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        assert int(d.max()) <= 1 and float((d != 0).mean()) <= 1e-3, (int(d.max()), float((d != 0).mean()))
    assert float(np.abs(frames.reshape(72, -1).mean(axis=1) - G["inf_frames_mean"]).max()) <= 1e-2
    clip = container.read_avi(out)                                  # the written container holds those frames + the audio
    assert np.array_equal(clip["frames"], frames) and clip["fps"] == 25.0 and clip["audio_sr"] == 16000
    assert clip["audio"].shape[0] == 48000


def test_reference_signature_datagen_yields_what_the_reference_yields(cuda):
    """`datagen(frames, mels)` driven by the module-level args (inference.py:108-154): float64 [B,96,96,6] batches bit-equal
    to the reference's, mel batches equal, frame copies and coords as the reference returns them"""
    from wav2lip_amd import inference
    mel, face = G["inf_mel"], G["inf_face"]
    chunks = datagen_ref.mel_chunks(mel, 25.0)
    saved = inference.args
    try:
        inference.args = inference.parser.parse_args(["--checkpoint_path", "", "--face", "", "--audio", "", "--box", "0", "96", "0",
                                                      "96", "--wav2lip_batch_size", "32"])
        inference.args.img_size, inference.args.static = 96, True
        gen = list(inference.datagen([face], chunks))
    finally:
        inference.args = saved
    assert [len(b[0]) for b in gen] == [32, 32, 8]
    img, melb, fb, cb = gen[0]
    assert img.dtype == np.float64 and sha(img) == str(G["dg_img_batch0_sha"]) and np.array_equal(melb, G["dg_mel_batch0"])
    assert np.array_equal(gen[2][1], G["dg_tail_mel_batch"]) and tuple(cb[0]) == (0, 96, 0, 96)
    assert all(np.array_equal(f, face) for f in fb)


def test_main_with_a_box_that_needs_resizing_matches_the_restated_pipeline(cuda, tmp_path):
    """a 120x150 frame with a 90x110 face box: crop -> resize to 96 (device) -> generator -> resize to the box -> paste (device),
    against the oracle chain (resize restated from OpenCV, oracle network, same uint8 truncation)"""
    from wav2lip_amd import inference
    tmp = str(tmp_path)
    r = np.random.default_rng(8)
    frame = r.integers(0, 256, (120, 150, 3), dtype=np.uint8)
    wav = synth.noise_wav(16000, seed=3)
    sd = _write_inputs(tmp, frame, wav)
    box = (10, 100, 20, 130)
    frames = inference.main(["--checkpoint_path", os.path.join(tmp, "ckpt.pth"), "--face", os.path.join(tmp, "face.png"),
                             "--audio", os.path.join(tmp, "audio.wav"), "--box"] + [str(v) for v in box] +
                            ["--wav2lip_batch_size", "16", "--outfile", os.path.join(tmp, "o.avi")])
    from oracle import audio_ref
    from scipy.io import wavfile
    mel = audio_ref.melspectrogram(audio_ref.load_wav_pcm16(os.path.join(tmp, "audio.wav")))
    chunks = datagen_ref.mel_chunks(mel, 25.0)
    assert len(frames) == len(chunks)
    face = resize_ref.crop_resize(frame, box)
    n = 4                                                            # the first frames are enough for the oracle chain
    img, melb = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(np.stack([face] * n), np.stack(chunks[:n])))
    pred = datagen_ref.frames_to_u8(models_ref.wav2lip_forward(sd, torch.from_numpy(melb), torch.from_numpy(img)).numpy())
    for k in range(n):
        ref = resize_ref.resize_paste(frame.copy(), pred[k], box)
        d = np.abs(frames[k].astype(np.int32) - ref.astype(np.int32))
        assert int(d.max()) <= 2 and float((d != 0).mean()) <= 2e-3, (k, int(d.max()), float((d != 0).mean()))
        assert np.array_equal(frames[k][:10], frame[:10])            # outside the box: the input frame, untouched


def _store(cuda):
    from wav2lip_amd.data import ClipStore
    store = ClipStore(cuda)
    seeds, nframes = G["ds_clip_seeds"].tolist(), G["ds_clip_frames"].tolist()
    for seed, n in zip(seeds, nframes):
        frames = [synth.face_crops_u8(1, seed=1000 * seed + k)[0] for k in range(n)]
        store.add_clip(frames, list(range(n)), synth.noise_wav(int(16000 * n / 25.0), seed=50 + seed))
    return store


def test_clip_store_reproduces_the_reference_dataset_samples(cuda):
    """the device-resident training store on the picks the reference's `Dataset.__getitem__` drew (wav2lip_train.py:108-164,
    color_syncnet_train.py:68-131): pixel tensors bit-exact, mel windows within the mel tolerance"""
    store = _store(cuda)
    picks = [tuple(G["gen%d_pick" % j].tolist()) for j in range(3)]
    x, indiv, mel, y = store.generator_batch(picks)
    for j in range(3):
        assert sha(x[j].cpu().numpy()) == str(G["gen%d_x_sha" % j]) and sha(y[j].cpu().numpy()) == str(G["gen%d_y_sha" % j])
        assert np.array_equal(x[j].cpu().numpy()[:, :, ::12, ::12], G["gen%d_x_sub" % j])
        assert float(np.abs(indiv[j].cpu().numpy() - G["gen%d_indiv" % j]).max()) <= 1e-4
        assert float(np.abs(mel[j].cpu().numpy() - G["gen%d_mel" % j]).max()) <= 1e-4
    spicks = [tuple(int(v) if i < 3 else bool(v) for i, v in enumerate(G["sync%d_pick" % j].tolist())) for j in range(4)]
    xs, mels, ys = store.syncnet_batch(spicks)
    for j in range(4):
        assert sha(xs[j].cpu().numpy()) == str(G["sync%d_x_sha" % j])
        assert float(np.abs(mels[j].cpu().numpy() - G["sync%d_mel" % j]).max()) <= 1e-4
        assert float(ys[j]) == float(spicks[j][3])
    # the `short` clip (10 frames <= 3*T) is never drawn and the sampler only returns picks the reference accepts
    _, _, _, _, drawn = store.sample_generator_batch(16, random.Random(3))
    assert all(c in (0, 1) for c, _, _ in drawn)
    with pytest.raises(ValueError):
        store.generator_batch([(0, 0, 5)])                           # frame 0 has no preceding frame for the segmented mels


def test_training_loops_follow_the_reference_scripts(cuda, tmp_path):
    """train() / eval_model() of the three scripts for a few steps on a tiny device-resident set: event order (sample images on the
    pre-increment step, checkpoints at step 1 and every interval, named after the global step), the syncnet_wt switch after a
    low evaluation, resume through load_checkpoint / overwrite_global_states"""
    from wav2lip_amd import models, optim, trainer
    from wav2lip_amd.hparams import hparams
    store = _store(cuda)
    ck = str(tmp_path)
    logs = []
    saved = {k: getattr(hparams, k) for k in ("syncnet_wt", "eval_interval", "syncnet_eval_interval", "disc_wt")}
    try:
        # ---- color_syncnet_train
        S = models.SyncNet_color().to(cuda)
        optS = optim.Adam([p for p in S.parameters() if p.requires_grad], lr=hparams.syncnet_lr)
        run = trainer.Run(ck)
        run.log = lambda *a: logs.append(" ".join(str(v) for v in a))
        hparams.set_hparam("syncnet_eval_interval", 2)
        loader = trainer.ClipLoader(store, 2, "syncnet", random.Random(1))
        assert len(loader) == 2                                         # ceil(3 clips / 2)
        trainer.train_syncnet(run, cuda, S, loader, loader, optS, checkpoint_interval=2, nepochs=2, eval_steps=0)
        assert run.global_step == 4 and run.global_epoch == 2
        assert sorted(f for f in os.listdir(ck) if f.startswith("checkpoint")) == [
            "checkpoint_step000000001.pth", "checkpoint_step000000002.pth", "checkpoint_step000000004.pth"]
        assert sum("Evaluating for 0 steps" in l for l in logs) == 2
        sync_ckpt = os.path.join(ck, "checkpoint_step000000004.pth")
        # ---- wav2lip_train with the frozen expert from that checkpoint
        G_ = models.Wav2Lip().to(cuda)
        E = models.SyncNet_color().to(cuda)
        for p in E.parameters():
            p.requires_grad = False
        run = trainer.Run(ck, E)
        run.log = lambda *a: logs.append(" ".join(str(v) for v in a))
        optG = optim.Adam([p for p in G_.parameters() if p.requires_grad], lr=hparams.initial_learning_rate)
        trainer.load_checkpoint(run, sync_ckpt, E, None, reset_optimizer=True, overwrite_global_states=False)
        assert run.global_step == 0                                      # the expert's counters do not overwrite the run's
        hparams.set_hparam("syncnet_wt", 0.03)
        hparams.set_hparam("eval_interval", 2)
        # batch 2: the frozen expert runs in train mode (wav2lip_train.py never calls .eval() on it), so a batch of ONE raises
        # torch's "Expected more than 1 value per channel when training" - here as in the reference
        gl = trainer.ClipLoader(store, 2, "generator", random.Random(2))
        ck2 = os.path.join(ck, "gen")
        os.mkdir(ck2)
        trainer.train_wav2lip(run, cuda, G_, gl, gl, optG, checkpoint_dir=ck2, checkpoint_interval=2, nepochs=2, eval_steps=0)
        assert run.global_step == 4 and run.global_epoch == 2
        names = sorted(os.listdir(ck2))
        assert names == ["checkpoint_step000000001.pth", "checkpoint_step000000002.pth", "checkpoint_step000000004.pth",
                         "samples_step000000000", "samples_step000000002"], names
        assert len(os.listdir(os.path.join(ck2, "samples_step000000000"))) == 10      # batch 2 x T 5 collages
        with pytest.raises(ValueError, match="more than 1 value per channel"):
            one = trainer.ClipLoader(store, 1, "generator", random.Random(4))
            trainer.train_wav2lip(trainer.Run(ck2, E), cuda, G_, one, one, optG, checkpoint_dir=ck2, checkpoint_interval=100,
                                  nepochs=1, eval_steps=0, max_steps=1)
        assert hparams.syncnet_wt in (0.03, 0.01)                         # 0.01 iff an evaluation averaged a sync loss < 0.75
        payload = torch.load(os.path.join(ck2, "checkpoint_step000000002.pth"), weights_only=False)
        assert payload["global_step"] == 2 and payload["global_epoch"] == 0 and payload["optimizer"] is not None
        # resume: counters and optimiser state come back (wav2lip_train.py:327-349)
        G2 = models.Wav2Lip().to(cuda)
        opt2 = optim.Adam([p for p in G2.parameters() if p.requires_grad], lr=hparams.initial_learning_rate)
        run2 = trainer.Run(ck2, E)
        run2.log = lambda *a: None
        trainer.load_checkpoint(run2, os.path.join(ck2, "checkpoint_step000000002.pth"), G2, opt2)
        assert (run2.global_step, run2.global_epoch) == (2, 0) and int(opt2.state_dict()["state"][0]["step"]) == 2
        # ---- hq_wav2lip_train: two steps
        D = models.Wav2Lip_disc_qual().to(cuda)
        optD = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=hparams.disc_initial_learning_rate, betas=(0.5, 0.999))
        ck3 = os.path.join(ck, "hq")
        os.mkdir(ck3)
        run3 = trainer.Run(ck3, E)
        run3.log = lambda *a: logs.append(" ".join(str(v) for v in a))
        trainer.train_hq(run3, cuda, G2, D, gl, gl, opt2, optD, checkpoint_interval=2, nepochs=1, eval_steps=0, max_steps=2)
        assert run3.global_step == 2
        assert {"checkpoint_step000000001.pth", "disc_checkpoint_step000000001.pth", "checkpoint_step000000002.pth",
                "disc_checkpoint_step000000002.pth", "samples_step000000000"} <= set(os.listdir(ck3))
        assert any(l.startswith("L1: ") and "Percep" in l for l in logs)
    finally:
        for k, v in saved.items():
            hparams.set_hparam(k, v)


def test_standalone_block_in_train_mode_runs_on_the_train_graph(cuda):
    """models/conv.py:14-19 for ONE block called on its own in train mode: batch statistics, residual, ReLU, running-stat
    update, and gradients through it - vs torch"""
    import torch.nn.functional as F
    from wav2lip_amd.models.conv import Conv2d
    torch.manual_seed(3)
    blk = Conv2d(8, 8, 3, 1, 1, residual=True).to(cuda).train()
    x = torch.randn(4, 8, 10, 12)
    conv, bn = blk.conv_block[0], blk.conv_block[1]
    w, b, g, be = (t.detach().cpu().clone().requires_grad_(True) for t in (conv.weight, conv.bias, bn.weight, bn.bias))
    xr = x.clone().requires_grad_(True)
    rm, rv = bn.running_mean.cpu().clone(), bn.running_var.cpu().clone()
    ref = F.relu(F.batch_norm(F.conv2d(xr, w, b, padding=1), rm, rv, g, be, training=True, momentum=0.1, eps=1e-5) + xr)
    ref.square().sum().backward()
    xg = x.to(cuda).requires_grad_(True)
    out = blk(xg)
    out.square().sum().backward()
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 1e-4
    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= 2e-4 * float(xr.grad.abs().max())
    assert float((conv.weight.grad.cpu() - w.grad).abs().max()) <= 2e-4 * float(w.grad.abs().max())
    assert float((bn.running_mean.cpu() - rm).abs().max()) <= 1e-6 and float((bn.running_var.cpu() - rv).abs().max()) <= 1e-6
