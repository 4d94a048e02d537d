#!/usr/bin/env python
"""Throughput of the device-resident training set (wav2lip_amd/data.py) on synthetic clips (GPU box): build time per clip and
generator / SyncNet batches per second, next to the training step they feed (cfg4: 64 samples per 73 ms step)."""
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from wav2lip_amd import data, synthetic


def main():
    dev = torch.device("cuda")
    r = np.random.default_rng(0)
    n_clips, n_frames = 100, 75                       # 3 s clips at 25 fps
    store = data.ClipStore(dev)
    t0 = time.perf_counter()
    for c in range(n_clips):
        size = (96, 96) if c % 2 == 0 else (128, 112)  # half of the crops need the resize kernel
        frames = [r.integers(0, 256, size + (3,), dtype=np.uint8) for _ in range(n_frames)]
        store.add_clip(frames, list(range(n_frames)), synthetic.noise_wav(16000 * 3 + 800, seed=c))
    store.frames()
    torch.cuda.synchronize()
    build = time.perf_counter() - t0
    rng = random.Random(0)
    out = {"what": "ClipStore (device-resident frames + mel bank), synthetic clips", "clips": n_clips, "frames": n_clips * n_frames,
           "build_ms_per_clip": round(1e3 * build / n_clips, 2), "hbm_mb": round(store.frames().numel() / 1e6, 1)}
    for name, fn in (("generator", store.sample_generator_batch), ("syncnet", store.sample_syncnet_batch)):
        B = 64 if name == "generator" else 512
        fn(B, rng)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            fn(B, rng)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name + "_batch"] = B
        out[name + "_ms_per_batch"] = round(1e3 * dt, 3)
        out[name + "_samples_per_s"] = round(B / dt, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
