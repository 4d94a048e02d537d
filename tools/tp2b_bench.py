#!/usr/bin/env python
"""Isolated launch times of the stride-2 transposed 3x3 layers of the bf16 training path (forward of the decoder's upsampling
layers, data gradients of the stride-2 convs; 320 frames) - run once per setting of W2L_CONVB_TP2B (0: four-phase implicit GEMM,
1: fused-phase kernel for the 32-cout tile, 2: for every tile):
    for v in 0 1 2; do W2L_CONVB_TP2B=$v python tools/tp2b_bench.py; done"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wav2lip_amd import bf16
from wav2lip_amd._lib import ACT_NONE, ConvGeom

# (name, cin, cout, N, H, W, with_res): transposed 3x3 / stride 2 / pad 1 / output pad 1, input H x W
SHAPES = [("dec6.0 fwd 160->64 @48", 160, 64, 320, 48, 48, False), ("dec5.0 fwd 320->128 @24", 320, 128, 320, 24, 24, False),
          ("dec4.0 fwd 512->256 @12", 512, 256, 320, 12, 12, False), ("dec3.0 fwd 768->384 @6", 768, 384, 320, 6, 6, False),
          ("enc1.0 dgrad 32->16 @48", 32, 16, 320, 48, 48, True), ("enc2.0 dgrad 64->32 @24", 64, 32, 320, 24, 24, True),
          ("enc3.0 dgrad 128->64 @12", 128, 64, 320, 12, 12, True), ("enc4.0 dgrad 256->128 @6", 256, 128, 320, 6, 6, True),
          ("S.face4 dgrad 128->64 @12x24 b512", 128, 64, 512, 12, 24, False), ("S.face8 dgrad 256->128 @6x12 b512", 256, 128, 512, 6, 12, False)]


def timed(fn, reps=10):
    for _ in range(3):
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
    dev = torch.device("cuda")
    print("W2L_CONVB_TP2B=%s" % os.environ.get("W2L_CONVB_TP2B", "1"))
    for name, cin, cout, N, H, W, with_res in SHAPES:
        w = torch.randn(cin, cout, 3, 3, device=dev) / (cin * 9) ** 0.5
        layer = bf16.ConvB(ConvGeom(1, cin, cout, 3, 3, 2, 2, 1, 1, 1, 1, ACT_NONE), w)
        x = torch.randn(N, H, W, bf16.round8(cin), device=dev).to(torch.bfloat16)
        y = torch.zeros(N, 2 * H, 2 * W, bf16.round8(cout), device=dev, dtype=torch.bfloat16)
        A = bf16.ActB
        res = A(y, 0, cout) if with_res else None
        ms = timed(lambda: layer.run(A(x, 0, cin), A(y, 0, cout), res, None, None, 0))
        gf = 2.0 * N * H * W * cin * cout * 9 / 1e9
        mb = (x.numel() + y.numel() * (2 if with_res else 1)) * 2 / 1e6
        print("  %-36s %7.3f ms  %7.1f TFLOP/s  %6.0f GB/s of %5.0f MB in + out" % (name, ms, gf / ms, mb / ms, mb))


if __name__ == "__main__":
    main()
