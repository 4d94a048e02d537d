"""Seeded synthetic weights and inputs for benchmarks, smoke() and the tests (numpy PCG64: stable across torch versions, so
the committed golden outputs stay valid).  There is no network here for datasets or checkpoints: bench.py and
tools/*_bench.py build their workloads from this module.  Pure data generation, no reference arithmetic.

Weights: He-scaled conv kernels, small biases and *randomised* BatchNorm statistics (fresh-init BN is nearly the
identity and would hide BN-fusion bugs).  Inputs follow SURVEY.md section 8(d).
"""
import re
import zlib

import numpy as np
import torch


_CONVT = re.compile(r"face_decoder_blocks\.[1-6]\.0\.conv_block\.0\.weight$")


def _rng(seed, key):
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def synthetic_state_dict(shapes, seed=0):
    """shapes: {key: shape} in reference state-dict naming -> {key: torch tensor}"""
    sd = {}
    for key, shape in shapes.items():
        r = _rng(seed, key)
        shape = tuple(shape)
        if key.endswith("num_batches_tracked"):
            v = np.array(100, dtype=np.int64)
        elif key.endswith("running_mean"):
            v = r.normal(0.0, 0.1, shape)
        elif key.endswith("running_var"):
            v = r.uniform(0.6, 1.4, shape)
        elif ".conv_block.1." in key and key.endswith("weight"):   # BN gamma
            v = r.uniform(0.7, 1.1, shape)
        elif ".conv_block.1." in key and key.endswith("bias"):     # BN beta
            v = r.normal(0.0, 0.1, shape)
        elif key.endswith("weight"):                                # conv / convT kernels
            if _CONVT.match(key):                                   # [cin, cout, k, k]; ~k*k/s^2 taps per output
                fan_in = shape[0] * (1.0 if shape[0] == 1024 and "blocks.1." in key else 2.25)
            else:
                fan_in = int(np.prod(shape[1:]))
            v = r.normal(0.0, np.sqrt(1.0 / max(fan_in, 1)), shape)
        else:                                                       # conv bias
            v = r.normal(0.0, 0.05, shape)
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.int64 if v.dtype == np.int64 else np.float32).copy())
    return sd


def face_crops_u8(n, seed=0, size=96):
    """uint8 BGR crops [n, size, size, 3]"""
    return _rng(seed, "faces").integers(0, 256, (n, size, size, 3), dtype=np.uint8)


def mel_windows(n, seed=0):
    """[n, 80, 16] float32 in the normalised mel range U(-4, 4)"""
    return _rng(seed, "mel").uniform(-4.0, 4.0, (n, 80, 16)).astype(np.float32)


def sine_wav(seconds=3.0, freq=440.0, sr=16000, amp=0.5):
    """config-1 audio: sine written as PCM16 and read back the way librosa/soundfile does (int16 / 32768)"""
    t = np.arange(int(seconds * sr)) / sr
    pcm = np.round(amp * np.sin(2 * np.pi * freq * t) * 32767.0).astype(np.int16)
    return (pcm.astype(np.float32) / 32768.0).astype(np.float32)


def noise_wav(nsamples, seed=0):
    return _rng(seed, "noise").uniform(-1.0, 1.0, nsamples).astype(np.float32)


def sync_faces(n, seed=0):
    """SyncNet face input [n, 15, 48, 96] U(0,1)"""
    return _rng(seed, "syncfaces").uniform(0.0, 1.0, (n, 15, 48, 96)).astype(np.float32)


def disc_frames(n, t, seed=0):
    """[n, 3, t, 96, 96] U(0,1)"""
    return _rng(seed, "discframes").uniform(0.0, 1.0, (n, 3, t, 96, 96)).astype(np.float32)


# ---------------------------------------------------------------- S3FD (no s3fd.pth offline)
_S3FD_CONVS = [("conv1_1", 3, 64, 3), ("conv1_2", 64, 64, 3), ("conv2_1", 64, 128, 3), ("conv2_2", 128, 128, 3),
               ("conv3_1", 128, 256, 3), ("conv3_2", 256, 256, 3), ("conv3_3", 256, 256, 3), ("conv4_1", 256, 512, 3),
               ("conv4_2", 512, 512, 3), ("conv4_3", 512, 512, 3), ("conv5_1", 512, 512, 3), ("conv5_2", 512, 512, 3),
               ("conv5_3", 512, 512, 3), ("fc6", 512, 1024, 3), ("fc7", 1024, 1024, 1), ("conv6_1", 1024, 256, 1),
               ("conv6_2", 256, 512, 3), ("conv7_1", 512, 128, 1), ("conv7_2", 128, 256, 3)]
_S3FD_NORMS = [("conv3_3_norm", 256, 10.), ("conv4_3_norm", 512, 8.), ("conv5_3_norm", 512, 5.)]
_S3FD_HEADS = [("conv3_3_norm", 256, 4), ("conv4_3_norm", 512, 2), ("conv5_3_norm", 512, 2), ("fc7", 1024, 2), ("conv6_2", 512, 2),
               ("conv7_2", 256, 2)]


def s3fd_state_dict(seed=0):
    """He-scaled VGG trunk (first layer divided by 128: pixel values are O(128)), small heads with a background-leaning conf
    bias: a detector that fires on a few positions only, in the reference's state-dict naming"""
    r = np.random.default_rng(seed)
    sd = {}
    for name, cin, cout, k in _S3FD_CONVS:
        sd[name + ".weight"] = torch.from_numpy(r.normal(0, np.sqrt(2.0 / (cin * k * k)), (cout, cin, k, k)).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy(r.normal(0, 0.05, cout).astype(np.float32))
    sd["conv1_1.weight"] = sd["conv1_1.weight"] / 128.0
    for name, c, scale in _S3FD_NORMS:
        sd[name + ".weight"] = torch.from_numpy((scale * r.uniform(0.8, 1.2, c)).astype(np.float32))
    for src, cin, ncls in _S3FD_HEADS:
        sd[src + "_mbox_conf.weight"] = torch.from_numpy(r.normal(0, 0.02, (ncls, cin, 3, 3)).astype(np.float32))
        b = r.normal(0, 0.05, ncls).astype(np.float32)
        b[-1] -= 1.0
        sd[src + "_mbox_conf.bias"] = torch.from_numpy(b)
        sd[src + "_mbox_loc.weight"] = torch.from_numpy(r.normal(0, 0.02, (4, cin, 3, 3)).astype(np.float32))
        sd[src + "_mbox_loc.bias"] = torch.from_numpy(r.normal(0, 0.05, 4).astype(np.float32))
    return sd


def s3fd_frames(seed=1, B=2, H=96, W=128):
    """uint8 BGR frames with one saturated block each (drives a few detector positions above the 0.5 threshold)"""
    r = np.random.default_rng(seed)
    img = r.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    img[0, 20:60, 30:90] = 255
    if B > 1:
        img[1, 40:80, 10:70] = 0
    return img


# ---------------------------------------------------------------- training batches at the BASELINE launch shapes
def train_batch(cfg, B, seed=0, T=5):
    """Seeded inputs of one training step (BASELINE configs[2..4]; the committed goldens of tests/golden/
    make_golden_train_baseline.py, the GPU tests at those shapes and tools/train_bench.py's in-run parity check all call this).
      cfg 3: {"x": [B,15,48,96] U(0,1), "mel": [B,1,80,16] U(-4,4), "y": [B,1] alternating 1, 0}     color_syncnet_train.py:155-165
      cfg 4 / 5: {"x": [B,6,T,96,96] (masked ground truth | wrong window), "indiv_mels": [B,T,1,80,16], "mel": [B,1,80,16],
                  "gt": [B,3,T,96,96]}                                      wav2lip_train.py:220-231, hq_wav2lip_train.py:221-256"""
    if cfg == 3:
        y = np.zeros((B, 1), np.float32)
        y[0::2] = 1.0
        return {"x": sync_faces(B, seed), "mel": mel_windows(B, seed)[:, None], "y": y}
    if cfg not in (4, 5):
        raise ValueError("train_batch: cfg must be 3, 4 or 5")
    r = _rng(seed, "trainbatch")
    gt = r.uniform(0.0, 1.0, (B, 3, T, 96, 96)).astype(np.float32)
    wrong = r.uniform(0.0, 1.0, (B, 3, T, 96, 96)).astype(np.float32)
    masked = gt.copy()
    masked[:, :, :, 48:] = 0.0
    return {"x": np.concatenate([masked, wrong], axis=1),
            "indiv_mels": r.uniform(-4.0, 4.0, (B, T, 1, 80, 16)).astype(np.float32),
            "mel": r.uniform(-4.0, 4.0, (B, 1, 80, 16)).astype(np.float32), "gt": gt}


def sketch_vectors(name, n, k=4):
    """k fixed +-1 vectors of length n for parameter `name`: the inner products of a gradient with them ("sketches") pin its
    DIRECTION in a golden file that cannot hold the tensor (E <d, r>^2 = |d|^2 for a difference d)"""
    r = np.random.default_rng([77, zlib.crc32(name.encode())])
    return r.integers(0, 2, (k, n), dtype=np.int8) * 2 - 1
