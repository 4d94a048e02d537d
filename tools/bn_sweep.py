#!/usr/bin/env python
"""Micro-benchmark of the HBM-bound BatchNorm passes (GPU box): statistics, apply, backward on [rows, C] activations of the
generator's big layers.  Prints ms and effective TB/s (bytes the pass must move / time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wav2lip_amd import _lib

SHAPES = [(320 * 96 * 96, 64), (320 * 48 * 48, 128), (320 * 24 * 24, 256), (320 * 12 * 12, 384), (320 * 6 * 6, 512)]


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    tag = os.environ.get("W2L_HIP_LIB", "default").split("libw2l_hip")[-1]
    s = _lib.current_stream
    p = _lib.ptr
    for rows, C in SHAPES:
        z, y, dy, dz, g = (torch.randn(rows, C, device=dev) for _ in range(5))
        v = [torch.ones(C, device=dev) for _ in range(10)]
        nbytes = rows * C * 4

        def stats():
            _lib.check(lib.w2l_bn_train_stats(s(), rows, C, p(z), C, p(v[0]), p(v[1]), 1e-5, 0.1, p(v[2]), p(v[3]), p(v[4]), p(v[5]),
                                              p(v[6]), p(v[7])), "stats")

        def apply():
            _lib.check(lib.w2l_affine_act(s(), rows, C, p(z), C, p(v[4]), p(v[5]), p(dy), C, 1, p(y), C), "affine")

        def bwd():
            _lib.check(lib.w2l_bn_train_bwd(s(), rows, C, p(dy), C, p(y), C, p(z), C, 1, p(v[2]), p(v[3]), p(v[4]), p(v[8]), p(v[9]),
                                            p(dz), C, p(g), C), "bwd")
        t_s, t_a, t_b = timeit(stats), timeit(apply), timeit(bwd)
        print("%s rows=%8d C=%4d (%5.0f MB)  stats %.3f ms %.2f TB/s | apply(+res) %.3f ms %.2f TB/s | bwd %.3f ms %.2f TB/s" %
              (tag, rows, C, nbytes / 1e6, t_s, nbytes / t_s / 1e9, t_a, 3 * nbytes / t_a / 1e9, t_b, 8 * nbytes / t_b / 1e9), flush=True)


if __name__ == "__main__":
    main()
