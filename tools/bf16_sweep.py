#!/usr/bin/env python
"""Per-layer timings of the bf16-storage training kernels on the shapes of BASELINE configs[3] (generator, 320 frames): the
forward conv / data-gradient launches (w2l_convb_*), every tile, next to the round-2 kernel they replace (fp32 tensors, operands
rounded to bf16 inside conv_igemm_bf16_kernel), and the weight gradients (w2l_conv_wgrad_bf16 vs w2l_conv_wgrad_prec).

    python tools/bf16_sweep.py [--fwd] [--wgrad] [--N 320] [--tiles]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from wav2lip_amd import _lib, bf16
from wav2lip_amd._lib import ACT_NONE, ACT_RELU, ConvGeom, check, ptr

# (name, transposed, cin, cout, k, stride, pad, outpad, H, W, residual)
LAYERS = [
    ("enc0 7x7 6->16 @96", 0, 6, 16, 7, 1, 3, 0, 96, 96, 0),
    ("enc1 s2 16->32 @96", 0, 16, 32, 3, 2, 1, 0, 96, 96, 0),
    ("res32 @48", 0, 32, 32, 3, 1, 1, 0, 48, 48, 1),
    ("enc s2 32->64 @48", 0, 32, 64, 3, 2, 1, 0, 48, 48, 0),
    ("res64 @24", 0, 64, 64, 3, 1, 1, 0, 24, 24, 1),
    ("res128 @12", 0, 128, 128, 3, 1, 1, 0, 12, 12, 1),
    ("res256 @6", 0, 256, 256, 3, 1, 1, 0, 6, 6, 1),
    ("res512 @3", 0, 512, 512, 3, 1, 1, 0, 3, 3, 1),
    ("convT 1024->512 @3", 1, 1024, 512, 3, 2, 1, 1, 3, 3, 0),
    ("res512 @6", 0, 512, 512, 3, 1, 1, 0, 6, 6, 1),
    ("convT 768->384 @6", 1, 768, 384, 3, 2, 1, 1, 6, 6, 0),
    ("res384 @12", 0, 384, 384, 3, 1, 1, 0, 12, 12, 1),
    ("convT 512->256 @12", 1, 512, 256, 3, 2, 1, 1, 12, 12, 0),
    ("res256 @24", 0, 256, 256, 3, 1, 1, 0, 24, 24, 1),
    ("convT 320->128 @24", 1, 320, 128, 3, 2, 1, 1, 24, 24, 0),
    ("res128 @48", 0, 128, 128, 3, 1, 1, 0, 48, 48, 1),
    ("convT 160->64 @48", 1, 160, 64, 3, 2, 1, 1, 48, 48, 0),
    ("res64 @96", 0, 64, 64, 3, 1, 1, 0, 96, 96, 1),
    ("out 80->32 @96", 0, 80, 32, 3, 1, 1, 0, 96, 96, 0),
]


def timed(fn, reps=5):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--fwd", action="store_true")
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--tiles", action="store_true", help="--fwd: time every tile of the bf16-storage kernel, not only the automatic one")
    ap.add_argument("--N", type=int, default=320)
    ap.add_argument("--only", type=str, default=None, help="substring filter on the layer name")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    from wav2lip_amd.autograd import RawConv
    from wav2lip_amd.engine import Act
    N = args.N
    tot = {}
    for name, tr, cin, cout, k, s, p, op, H, W, res in LAYERS:
        if args.only and args.only not in name:
            continue
        g = ConvGeom(tr, cin, cout, k, k, s, s, p, p, op, op, ACT_RELU)
        w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device=dev) * 0.05
        macs = int(lib.w2l_conv_macs(C.byref(g), N, H, W))
        if args.fwd:
            layer = bf16.ConvB(g, w)
            Ho, Wo = layer.out_hw(H, W)
            xb = torch.randn(N, H, W, bf16.round8(cin), device=dev).to(torch.bfloat16)
            yb = bf16.new_buf(N, Ho, Wo, cout, dev)
            xa, ya = bf16.ActB(xb, 0, cin), bf16.ActB(yb, 0, cout)
            sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
            row = "%-22s" % name
            best = None
            for tile in ([None] + list(range(lib.w2l_convb_num_tiles())) if args.tiles else [None]):
                layer.set_tile(-1 if tile is None else tile)
                ms = timed(lambda: layer.run(xa, ya, xa if res else None, sc, sh))
                row += "  %s %.3f ms %6.1f TF" % ("auto" if tile is None else "t%d" % tile, ms, 2e-9 * macs / ms)
                if tile is None:
                    best = ms
            # round 2's path: fp32 tensors, bf16 contraction inside the implicit GEMM
            ones, zeros = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
            old = RawConv(g, w, ones, zeros, "bf16c")
            xf = torch.randn(N, H, W, (cin + 3) // 4 * 4, device=dev)
            yf = torch.empty(N, Ho, Wo, (cout + 3) // 4 * 4, device=dev)
            xo, yo = Act(xf, 0, cin), Act(yf, 0, cout)
            ms_old = timed(lambda: old.run(xo, yo, xo if res else None))
            row += "  | r02 %.3f ms %6.1f TF  (x%.2f)" % (ms_old, 2e-9 * macs / ms_old, ms_old / best)
            tot["fwd_new"] = tot.get("fwd_new", 0) + best
            tot["fwd_old"] = tot.get("fwd_old", 0) + ms_old
            print(row, flush=True)
            del layer, old, xb, yb, xf, yf
        if args.wgrad:
            Ho = (H - 1) * s - 2 * p + k + op if tr else (H + 2 * p - k) // s + 1
            Wo = (W - 1) * s - 2 * p + k + op if tr else (W + 2 * p - k) // s + 1
            xb = torch.randn(N, H, W, bf16.round8(cin), device=dev).to(torch.bfloat16)
            dzb = torch.randn(N, Ho, Wo, bf16.round8(cout), device=dev).to(torch.bfloat16)
            dw = torch.empty_like(w)
            s_ = _lib.current_stream
            ms = timed(lambda: check(lib.w2l_conv_wgrad_bf16(C.byref(g), s_(), N, H, W, ptr(xb), xb.shape[-1], ptr(dzb), dzb.shape[-1],
                                                             ptr(dw)), "conv_wgrad_bf16"))
            xf = torch.randn(N, H, W, (cin + 3) // 4 * 4, device=dev)
            dzf = torch.randn(N, Ho, Wo, (cout + 3) // 4 * 4, device=dev)
            ms_old = timed(lambda: check(lib.w2l_conv_wgrad_prec(C.byref(g), s_(), N, H, W, ptr(xf), xf.shape[-1], ptr(dzf),
                                                                 dzf.shape[-1], ptr(dw), _lib.PREC_BF16), "conv_wgrad"))
            print("%-22s wgrad %.3f ms %6.1f TF  | r02 %.3f ms %6.1f TF  (x%.2f)" % (name, ms, 2e-9 * macs / ms, ms_old,
                                                                                    2e-9 * macs / ms_old, ms_old / ms), flush=True)
            tot["wgrad_new"] = tot.get("wgrad_new", 0) + ms
            tot["wgrad_old"] = tot.get("wgrad_old", 0) + ms_old
            del xb, dzb, xf, dzf
        torch.cuda.empty_cache()
    print("totals (one launch per listed layer):", {k: round(v, 3) for k, v in tot.items()}, flush=True)


if __name__ == "__main__":
    main()
