#!/usr/bin/env python
"""Micro-benchmark of the fused conv kernel on synthetic layers (GPU box): time vs K, tile config, shape.
usage: python tools/conv_sweep.py [--tiles] [--ksweep]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wav2lip_amd import engine
from wav2lip_amd.models.conv import Conv2d, Conv2dTranspose

TILES = ["128x128", "128x64", "64x128", "64x64", "128x32", "32x128", "wino64x64k8", "wino32x128k16", "wino2_32x64", "wino2_64x32", "tp2", "wino4_32x64", "wino2q_32x32", "split128x128", "split128x64", "split64x128", "split64x64", "split128x32", "split32x128", "wino2s_64x64", "tp2s", "stem7s", "k3s"]


def bench(cin, cout, H, W, N, k=3, s=1, p=1, res=True, tile=None, reps=5, transposed=False, ks=None):
    dev = torch.device("cuda")
    if transposed:
        m = Conv2dTranspose(cin, cout, k, s, p, 1).to(dev).eval()
    else:
        m = Conv2d(cin, cout, k, s, p, residual=res and cin == cout and s == 1).to(dev).eval()
    layer = m.fused()
    if tile is not None:
        layer.set_tile(tile)
    ho, wo = layer.out_hw(H, W)
    cin_p = layer.cin_p                      # activations carry channel counts padded to 4 (zero pad channels)
    xt = torch.zeros(N, H, W, cin_p, device=dev)
    xt[..., :cin] = torch.randn(N, H, W, cin, device=dev)
    x = engine.Act(xt, 0, cin_p)
    y = engine.Act(torch.empty(N, ho, wo, cout, device=dev), 0, cout)
    plan = engine.Plan()
    plan.add("l", layer, x, y, x if m.residual else None)
    if ks is not None:
        plan.tuned = True
        plan.set_config(0, tile, ks)
    plan.run()
    torch.cuda.synchronize()
    ms = min(plan.profile(reps=reps)[0][1] for _ in range(3))
    macs = layer.macs(N, H, W)
    return ms, 2 * macs / ms / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ksweep", action="store_true")
    ap.add_argument("--tiles", action="store_true")
    ap.add_argument("--wino", action="store_true", help="Winograd configurations on the 3x3 s1 decoder shapes")
    ap.add_argument("--convt", action="store_true", help="the decoder's stride-2 transposed layers: implicit-GEMM tiles vs conv_tp2")
    ap.add_argument("--convt-table", action="store_true", help="the stride-2 transposed layers at every tuned batch size: table entry vs conv_tp2s")
    ap.add_argument("--k3s", action="store_true", help="the 32-cout 3x3 layers: table entry vs conv_k3s")
    ap.add_argument("--stem", action="store_true", help="the 7x7 first layer at every tuned batch size: table entry vs conv_stem7s")
    ap.add_argument("--one", type=int, nargs=4, metavar=("CIN", "COUT", "H", "W"), help="time one 3x3 s1 p1 layer")
    ap.add_argument("--cinsweep", action="store_true", help="Winograd 64-cout layer at 96x96: time vs cin (fixed-cost fit)")
    ap.add_argument("--tile", type=int, default=None)
    ap.add_argument("--only-tile", type=int, default=None, help="--wino: time only this configuration id")
    ap.add_argument("--N", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    tag = os.environ.get("W2L_HIP_LIB", "default").split("libw2l_hip")[-1]
    if args.one:
        cin, cout, H, W = args.one
        ms, tf = bench(cin, cout, H, W, args.N, tile=args.tile, reps=args.reps)
        print("%s one %d->%d @%dx%d N=%d tile=%s  %8.3f ms %7.2f TFLOP/s" %
              (tag, cin, cout, H, W, args.N, "auto" if args.tile is None else TILES[args.tile], ms, tf), flush=True)
    if args.cinsweep:
        # fixed M (96x96, batch N), cout 64, Winograd tile 6: time vs number of K-steps -> per-workgroup fixed cost = intercept
        for cin in (8, 16, 32, 64, 128, 256):
            ms, tf = bench(cin, 64, 96, 96, args.N, res=False, tile=6 if args.tile is None else args.tile, reps=args.reps)
            print("%s cinsweep cin=%4d steps=%3d  %8.3f ms %7.2f TFLOP/s" % (tag, cin, cin // 8, ms, tf), flush=True)
    if args.wino:
        import ctypes
        ntiles = engine._lib.load().w2l_conv_num_tiles()
        shapes = [("dec6 64@96", 64, 64, 96, 96), ("dec5 128@48", 128, 128, 48, 48), ("dec4 256@24", 256, 256, 24, 24),
                  ("dec3 384@12", 384, 384, 12, 12), ("dec2 512@6", 512, 512, 6, 6), ("enc2 64@24", 64, 64, 24, 24),
                  ("enc3 128@12", 128, 128, 12, 12), ("enc1 32@48", 32, 32, 48, 48), ("aud 32@80x16", 32, 32, 80, 16),
                  ("out 80->32@96", 80, 32, 96, 96)]
        for name, cin, cout, H, W in shapes:
            for tile in ([3, 4] if cout == 32 else []) + list(range(6, ntiles)):
                if args.only_tile is not None and tile != args.only_tile:
                    continue
                if tile == 10 or (tile >= 6 and (cout % 64 if tile not in (9, 12) else cout % 32)):
                    continue
                ms, tf = bench(cin, cout, H, W, args.N, tile=tile)
                print("%s wino %-14s tile=%-14s %8.3f ms %7.2f TFLOP/s" % (tag, name, TILES[tile], ms, tf), flush=True)
    if args.convt:
        shapes = [("dec2.0 1024->512@3", 1024, 512, 3, 3), ("dec3.0 768->384@6", 768, 384, 6, 6), ("dec4.0 512->256@12", 512, 256, 12, 12),
                  ("dec5.0 320->128@24", 320, 128, 24, 24), ("dec6.0 160->64@48", 160, 64, 48, 48)]
        for name, cin, cout, H, W in shapes:
            for tile, ks in ((0, None), (1, None), (2, None), (3, None), (5, None), (10, None), (13, 1), (13, 2), (15, 1), (15, 2), (20, None), (20, 2), (20, 4), (20, 8)):
                if args.only_tile is not None and tile != args.only_tile:
                    continue
                if ks is not None and tile == 20 and ks > 1 and H > 12:
                    continue
                ms, tf = bench(cin, cout, H, W, args.N, k=3, s=2, p=1, res=False, tile=tile, transposed=True, ks=ks)
                print("%s convt %-20s tile=%-12s ks=%-4s %8.3f ms %7.2f TFLOP/s" % (tag, name, TILES[tile], ks, ms, tf), flush=True)
    if args.convt_table:
        # the five upsampling layers at every tuned batch size: what the committed table / heuristic runs against conv_tp2s (id 20)
        shapes = [("dec2.0", 1024, 512, 3, 3), ("dec3.0", 768, 384, 6, 6), ("dec4.0", 512, 256, 12, 12), ("dec5.0", 320, 128, 24, 24),
                  ("dec6.0", 160, 64, 48, 48)]
        for N in (8, 16, 32, 64, 128, 256):
            for name, cin, cout, H, W in shapes:
                ms0, _ = bench(cin, cout, H, W, N, k=3, s=2, p=1, res=False, tile=None, transposed=True)
                ms1, _ = bench(cin, cout, H, W, N, k=3, s=2, p=1, res=False, tile=20, transposed=True)
                print("%s convt-table N=%-4d %-7s %4d->%-4d @%-3d table %8.4f ms   tp2s %8.4f ms   ratio %.3f" %
                      (tag, N, name, cin, cout, H, ms0, ms1, ms1 / ms0), flush=True)
    if args.k3s:
        # the 32-cout 3x3 layers: table entry vs conv_k3s (id 22); the output block carries its fused head in the plan, here without
        for name, cin, H, W, res in (("out 80->32@96", 80, 96, 96, False), ("enc1 32@48", 32, 48, 48, True), ("aud 32@80x16", 32, 80, 16, True)):
            for N in (16, 64, 128, 256):
                ms0, _ = bench(cin, 32, H, W, N, res=res, tile=None)
                ms1, _ = bench(cin, 32, H, W, N, res=res, tile=22)
                print("%s k3s %-14s N=%-4d table %8.4f ms   k3s %8.4f ms   ratio %.3f" % (tag, name, N, ms0, ms1, ms1 / ms0), flush=True)
    if args.stem:
        # the generator's first layer at every tuned batch size: table entry vs conv_stem7s (id 21)
        for N in (1, 2, 4, 8, 16, 32, 64, 128, 256):
            ms0, _ = bench(6, 16, 96, 96, N, k=7, s=1, p=3, res=False, tile=None)
            ms1, _ = bench(6, 16, 96, 96, N, k=7, s=1, p=3, res=False, tile=21)
            print("%s stem N=%-4d table %8.4f ms   stem7s %8.4f ms   ratio %.3f" % (tag, N, ms0, ms1, ms1 / ms0), flush=True)
    if args.ksweep:
        # fixed M = 128*48*48 = 294912 (2304 row tiles of 128), cout 128, K = 9*cin
        for tile in (0, 1, 3):
            for cin in (32, 64, 128, 256, 512):
                ms, tf = bench(cin, 128, 48, 48, 128, res=False, tile=tile)
                print("%s ksweep tile=%s cin=%4d K=%5d steps=%3d  %8.3f ms %7.2f TFLOP/s" %
                      (tag, TILES[tile], cin, 9 * cin, 9 * cin // 32, ms, tf), flush=True)
    if args.tiles:
        shapes = [("dec6 64@96", 64, 64, 96, 96), ("dec5 128@48", 128, 128, 48, 48), ("dec4 256@24", 256, 256, 24, 24),
                  ("dec3 384@12", 384, 384, 12, 12), ("dec2 512@6", 512, 512, 6, 6), ("enc5 512@3", 512, 512, 3, 3),
                  ("out 80->32@96", 80, 32, 96, 96)]
        for name, cin, cout, H, W in shapes:
            for tile in range(6):
                ms, tf = bench(cin, cout, H, W, 128, tile=tile)
                print("%s tiles %-14s tile=%-8s %8.3f ms %7.2f TFLOP/s" % (tag, name, TILES[tile], ms, tf), flush=True)


if __name__ == "__main__":
    main()
