#!/usr/bin/env python
"""Phase timeline of conv_wino4_f32_kernel from in-kernel shader-clock stamps (a variant library built with -DW4_TRACE:
bash tools/build_variant.sh trace conv_wino4.hip -DW4_TRACE; W2L_HIP_LIB=wav2lip_amd/lib/libw2l_hip_trace.so).
Stamps per work item: 0 start, 1 raw block 0 in LDS, 2 prologue done, 3 K loop done, 4-7 epilogue rounds done.

    python tools/wino4_trace.py CIN COUT H W [--N 128] [--nores]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from wav2lip_amd import _lib, engine
from wav2lip_amd.models.conv import Conv2d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", type=int, nargs=4)
    ap.add_argument("--N", type=int, default=128)
    ap.add_argument("--nores", action="store_true")
    ap.add_argument("--mhz", type=float, default=0.0, help="counter frequency (s_memtime runs at the 100 MHz reference clock)")
    args = ap.parse_args()
    cin, cout, H, W = args.shape
    dev = torch.device("cuda")
    m = Conv2d(cin, cout, 3, 1, 1, residual=(not args.nores) and cin == cout).to(dev).eval()
    layer = m.fused()
    layer.set_tile(11)
    x = engine.Act(torch.randn(args.N, H, W, cin, device=dev), 0, cin)
    y = engine.Act(torch.empty(args.N, H, W, cout, device=dev), 0, cout)
    plan = engine.Plan()
    plan.add("l", layer, x, y, x if m.residual else None)
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = np.zeros(256 * 16 * 10, dtype=np.uint64)
    rc = lib.w2l_dbg_w4_trace(buf.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    tr = buf[:256 * 16 * 8].reshape(256, 16, 8).astype(np.int64)
    rt = buf[256 * 16 * 8:].reshape(256, 16, 2).astype(np.int64)
    ok0 = tr[:, 0, 0] > 0
    mhz = float(np.median((tr[ok0, 0, 7] - tr[ok0, 0, 0]) / np.maximum(rt[ok0, 0, 1] - rt[ok0, 0, 0], 1)) * 100.0)
    print("s_memtime runs at %.0f MHz against the 100 MHz s_memrealtime over item 0" % mhz)
    per_item = []
    for it in range(16):
        ok = (tr[:, it, 0] > 0) & (rt[:, it, 1] > rt[:, it, 0])
        if ok.any():
            per_item.append("%.0f" % float(np.median((tr[ok, it, 7] - tr[ok, it, 0]) / (rt[ok, it, 1] - rt[ok, it, 0])) * 100.0))
    print("  per item:", " ".join(per_item))
    if args.mhz <= 0:
        args.mhz = mhz
    us = 1.0 / args.mhz
    names = ["raw0 landed", "prologue", "K loop", "epi r0", "epi r1", "epi r2", "epi r3"]
    print("layer %d->%d @%dx%d N=%d res=%s   (us, median over workgroups; min..max)" % (cin, cout, H, W, args.N, m.residual))
    nitems = int((tr[:, :, 0] > 0).sum(axis=1).max())
    for it in range(min(nitems, 12)):
        ok = tr[:, it, 0] > 0
        d = np.diff(tr[ok, it, :], axis=1) * us
        row = "item %2d: " % it + "  ".join("%s %.2f" % (n, np.median(d[:, k])) for k, n in enumerate(names))
        tot = (tr[ok, it, 7] - tr[ok, it, 0]) * us
        gap = ""
        if it + 1 < 16 and (tr[ok, it + 1, 0] > 0).all():
            gap = "  next-start gap %.2f" % np.median((tr[ok, it + 1, 0] - tr[ok, it, 7]) * us)
        print(row + "  | total %.2f (%.2f..%.2f)%s" % (np.median(tot), tot.min(), tot.max(), gap))
    t0 = tr[:, 0, 0][tr[:, 0, 0] > 0]
    print("first-item start spread over workgroups: %.2f us; kernel span %.2f us" %
          ((t0.max() - t0.min()) * us, (tr.max() - t0.min()) * us))


if __name__ == "__main__":
    main()
