#!/usr/bin/env python3
"""Accuracy of an fp32 contraction computed as bf16 pieces on the bf16 MFMA path ("next").

An fp32 value is the exact sum of three bf16 values (3 x 8 significant bits = 24): a = a0 + a1 + a2.  A product a*b then is the sum
of nine piece products; keeping the six with i + j <= 2 drops terms below 2^-24 relative.  Each piece product is exact in fp32, the
accumulation is fp32 as in the fp32 MFMA path.  This script measures, on CPU, the error of the 3-, 6- and 9-product variants against
an fp64 contraction, beside the error of a plain fp32 contraction, for K = 9 * 256 (a 3x3 layer of 256 channels) with activation- and
weight-like operands.  CPU only (numpy); no GPU code is involved.
"""
import numpy as np


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16, returned as fp32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x0 = bf16_round(x)
    r = (x - x0).astype(np.float32)
    x1 = bf16_round(r)
    x2 = bf16_round((r - x1).astype(np.float32))
    return x0, x1, x2


def mm32(a, b, chunk=16):
    """fp32 accumulation over K in MFMA-sized chunks (each chunk's dot in fp64, then rounded: the matrix core's inner sum is wider
    than fp32), chunk sums added in fp32 in K order."""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(0, a.shape[1], chunk):
        acc = (acc + (a[:, k:k + chunk].astype(np.float64) @ b[k:k + chunk].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc


def main():
    rng = np.random.default_rng(0)
    M, K, N = 256, 9 * 256, 64
    a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)        # post-ReLU activations
    b = (rng.standard_normal((K, N)) * 0.03).astype(np.float32)              # weights
    ref = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(ref).max()
    pa, pb = split3(a), split3(b)
    assert np.array_equal((pa[0].astype(np.float64) + pa[1] + pa[2]).astype(np.float32), a)
    rows = [("fp32 (chunks of 2, as v_mfma_f32_32x32x2_f32)", mm32(a, b, 2))]
    for name, pairs in (("bf16 x3 (i+j<=1)", [(0, 0), (0, 1), (1, 0)]),
                        ("bf16 x6 (i+j<=2)", [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
                        ("bf16 x9 (all)", [(i, j) for i in range(3) for j in range(3)])):
        acc = np.zeros((M, N), np.float32)
        # small terms first, so that they are not absorbed one by one into a large accumulator
        for i, j in sorted(pairs, key=lambda p: -(p[0] + p[1])):
            acc = (acc + mm32(pa[i], pb[j], 16)).astype(np.float32)
        rows.append((name, acc))
    print(f"M={M} K={K} N={N}  max |ref| = {scale:.3f}")
    for name, out in rows:
        err = np.abs(out - ref)
        print(f"  {name:48s} max abs err {err.max():.3e}   max err / max|ref| {err.max() / scale:.3e}   rms {np.sqrt((err**2).mean()):.3e}")


if __name__ == "__main__":
    main()
