"""Seeded synthetic weights and inputs shared by the tests, smoke() and bench.py (numpy PCG64: stable across
torch versions, so the committed golden outputs stay valid).  TEST INFRASTRUCTURE.

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
