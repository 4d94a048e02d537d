#!/usr/bin/env python
"""Micro-benchmark of w2l_conv_wgrad on the generator's 3x3 / s1 / p1 layers (GPU box).  The Winograd F(3x3,2x2) kernel is
switched by the environment variable W2L_WINO_WGRAD, read once per process:
    W2L_WINO_WGRAD=0 python tools/wgrad_sweep.py ; W2L_WINO_WGRAD=1 python tools/wgrad_sweep.py
Prints one line per layer: ms per launch and direct-convolution-equivalent TFLOP/s."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wav2lip_amd import _lib

# (cin, cout, H, W) of the face encoder / decoder residual blocks and the output block (models/wav2lip.py)
LAYERS = [(32, 32, 48, 48), (64, 64, 24, 24), (128, 128, 12, 12), (256, 256, 6, 6), (512, 512, 6, 6), (384, 384, 12, 12),
          (256, 256, 24, 24), (128, 128, 48, 48), (64, 64, 96, 96), (80, 32, 96, 96)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=80)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--layers", type=int, nargs="*", default=None, help="indices into LAYERS (default: all)")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda")
    tag = "wino=%s%s" % (os.environ.get("W2L_WINO_WGRAD", "1"), os.environ.get("W2L_HIP_LIB", "").split("libw2l_hip")[-1])
    for cin, cout, H, W in (LAYERS if args.layers is None else [LAYERS[i] for i in args.layers]):
        x = torch.randn(args.N, H, W, cin, device=dev)
        dz = torch.randn(args.N, H, W, cout, device=dev)
        dw = torch.empty(cout, cin, 3, 3, device=dev)
        g = _lib.ConvGeom(0, cin, cout, 3, 3, 1, 1, 1, 1, 0, 0, 0)

        def run():
            _lib.check(lib.w2l_conv_wgrad(C.byref(g), _lib.current_stream(), args.N, H, W, _lib.ptr(x), cin, _lib.ptr(dz), cout,
                                          _lib.ptr(dw)), "wgrad")
        run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.reps)
        fl = 2.0 * args.N * H * W * cin * cout * 9
        print("%s wgrad %4d->%4d @%3dx%-3d N=%d  %8.3f ms %7.2f TFLOP/s" % (tag, cin, cout, H, W, args.N, best, fl / best / 1e9),
              flush=True)


if __name__ == "__main__":
    main()
