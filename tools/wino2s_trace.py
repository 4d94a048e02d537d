#!/usr/bin/env python
"""Phase timeline of conv_wino2s_kernel from in-kernel shader-clock stamps (a variant library built with -DW2S_TRACE:
bash tools/build_variant.sh trace conv_wino2s.hip -DW2S_TRACE; W2L_HIP_LIB=wav2lip_amd/lib/libw2l_hip_trace.so).
Stamps of the second work item of every workgroup, lane 0 of wave 0 (row half g = 0) and wave 4 (g = 1): 0 item start, 1 prologue
done, then per chunk: requests issued, transform done, barrier, MFMAs issued, wait, barrier; 46 K loop left, 47 epilogue done.

    python tools/wino2s_trace.py CIN COUT H W [--N 128] [--nores]
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
    args = ap.parse_args()
    cin, cout, H, W = args.shape
    dev = torch.device("cuda")
    m = Conv2d(cin, cout, 3, 1, 1, residual=(not args.nores) and cin == cout).to(dev).eval()
    layer = m.fused()
    lib = _lib.load()
    layer.set_tile(lib.w2l_conv_num_tiles() - 1)
    x = engine.Act(torch.randn(args.N, H, W, cin, device=dev), 0, cin)
    y = engine.Act(torch.empty(args.N, H, W, cout, device=dev), 0, cout)
    plan = engine.Plan()
    plan.add("l", layer, x, y, x if m.residual else None)
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    dbuf = torch.zeros(256 * 2 * 50, dtype=torch.int64, device=dev)
    rc = lib.w2l_dbg_w2s_trace(C.c_void_p(dbuf.data_ptr()))
    assert rc == 0, rc
    buf = dbuf.cpu().numpy().view(np.uint64)
    tr = buf[:256 * 2 * 48].reshape(256, 2, 48).astype(np.int64)
    rt = buf[256 * 2 * 48:].reshape(256, 2, 2).astype(np.int64)
    ok = (tr[:, 0, 0] > 0) & (tr[:, 0, 47] > tr[:, 0, 0]) & (rt[:, 0, 1] > rt[:, 0, 0])
    mhz = float(np.median((tr[ok, 0, 47] - tr[ok, 0, 0]) / (rt[ok, 0, 1] - rt[ok, 0, 0])) * 100.0)
    us = 1.0 / mhz
    nchunk = cin // 16 + (cin // 16 & 1)
    print("layer %d->%d @%dx%d N=%d res=%s: %d workgroups traced, counter %.0f MHz; us, median over workgroups" %
          (cin, cout, H, W, args.N, m.residual, int(ok.sum()), mhz))
    for g in (0, 1):
        t = tr[ok, g, :]
        d = lambda a, b: float(np.median((t[:, b] - t[:, a]) * us))   # noqa: E731
        print("g%d  prologue %.2f" % (g, d(0, 1)))
        names = ["requests", "transform", "barrier", "MFMAs", "wait", "barrier"]
        for c in range(min(nchunk, 7)):
            b = 2 + 6 * c
            print("g%d  chunk %d: " % (g, c) + "  ".join("%s %.2f" % (n, d(b + k - 1 if (c or k) else 1, b + k)) for k, n in enumerate(names))
                  + "  | chunk total %.2f" % d(b - 1 if c else 1, b + 5))
        print("g%d  K loop end -> epilogue start %.2f, epilogue %.2f, item total %.2f" %
              (g, d(2 + 6 * min(nchunk, 7) - 1, 46) if nchunk <= 7 else float("nan"), d(46, 47), d(0, 47)))


if __name__ == "__main__":
    main()
