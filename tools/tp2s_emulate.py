#!/usr/bin/env python
"""The index arithmetic of conv_tp2s.hip (run), conv_k3s.hip (run_k3s) and conv_stem7s.hip (run_stem7s) restated on numpy arrays (no GPU).
conv_tp2s: the raw-slot -> LDS-plane map, the shifted A-fragment
addresses, the weight-fragment order of tp2_pack -> tp2s_pack, the tap sequence with its phases and shifts, and the accumulator ->
output-pixel map, run for one launch and compared with a direct transposed convolution.  The piece split itself is the identity here
(plane 0 holds the value): what is checked is WHERE every value goes.  Used by tests/test_wino2s_layout.py.

    python tools/tp2s_emulate.py            # a few geometries, prints max |err|
"""
import numpy as np

KYT = [1, 1, 1, 0, 2, 0, 0, 2, 2]      # tap -> (ky, kx), conv_tp2.hip's numbering
KXT = [1, 0, 2, 1, 1, 0, 2, 0, 2]
SEQ = [0, 2, 4, 8, 1, 7, 3, 6, 5]      # the order conv_tp2s walks the taps in (grouped by shift)


def ts_phase(t):
    return 0 if t == 0 else (1 if t < 3 else (2 if t < 5 else 3))


def ts_shift(t):
    return 1 if t in (1, 7) else (2 if t in (3, 6) else (3 if t == 5 else 0))


def tp2_pack(w):                      # w[cin][cout][3][3] -> conv_tp2's fp32 fragment order
    cin, cout = w.shape[:2]
    nks = cin // 8
    u = np.zeros(cout * cin * 9)
    for i in range(u.size):
        e, n, h = i & 3, (i >> 2) & 31, (i >> 7) & 1
        rest = i >> 8
        tap, r2 = rest % 9, rest // 9
        kc, nbk = r2 % nks, r2 // nks
        u[i] = w[kc * 8 + 4 * h + e, nbk * 32 + n, KYT[tap], KXT[tap]]
    return u


def tp2s_pack(u32, cin, cout):        # -> [(nb*nkc + kc)*9 + tap][plane][lane*8 + e]
    nkc, nks = cin // 16, cin // 8
    us = np.zeros((cout // 32 * nkc * 9, 3, 512))
    for i in range(cout * cin * 9):
        e, ln = i & 7, (i >> 3) & 63
        rest = i >> 9
        tap, r2 = rest % 9, rest // 9
        kc, nbk = r2 % nkc, r2 // nkc
        kc8, h, e4, n = kc * 2 + (ln >> 5), e >> 2, e & 3, ln & 31
        us[(nbk * nkc + kc) * 9 + tap, 0, ln * 8 + e] = u32[((nbk * nks + kc8) * 9 + tap) * 256 + (h * 32 + n) * 4 + e4]
    return us


def run(N, H, W, cin, cout, bh, bw, ni, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, H, W, cin))
    w = rng.standard_normal((cin, cout, 3, 3))
    us = tp2s_pack(tp2_pack(w), cin, cout)
    nkc = cin // 16
    RH, RW = bh + 1, bw + 1
    RP = ni * RH * RW
    assert RP <= 256 and bh * bw * ni <= 128
    bhw = bh * bw
    y = np.zeros((N, 2 * H, 2 * W, cout))
    nby, nbx, ngi, tiles_n = -(-H // bh), -(-W // bw), -(-N // ni), cout // 64
    for bid in range(ngi * nby * nbx * tiles_n):
        tile_n, mb = bid % tiles_n, bid // tiles_n
        bx_i, mb = mb % nbx, mb // nbx
        by_i, gi = mb % nby, mb // nby
        n0 = tile_n * 64
        acc = np.zeros((2, 2, 2, 4, 32, 32))                 # [wm][wn][b][phase][row][col]
        for step in range(nkc):
            planes = np.zeros((2, 256, 8))                   # plane 0 only: [kh][p][8]
            for t in range(256):
                for k in range(2):
                    e = t + 256 * k
                    kh, p = e & 1, e >> 1
                    if p < RP:
                        rxx, p2 = p % RW, p // RW
                        ry, il = p2 % RH, p2 // RH
                        n = gi * ni + il
                        iy, ix = by_i * bh + ry, bx_i * bw + rxx
                        if n < N and iy < H and ix < W:
                            planes[kh, p] = x[n, iy, ix, step * 16 + kh * 8: step * 16 + kh * 8 + 8]
            for wm in range(2):
                for wn in range(2):
                    nb = (n0 >> 5) + wn
                    for tap in SEQ:
                        ph, sd = ts_phase(tap), ts_shift(tap)
                        shpx = [0, 1, RW, RW + 1][sd]
                        B = np.zeros((16, 32))
                        for lane in range(64):
                            B[8 * (lane >> 5): 8 * (lane >> 5) + 8, lane & 31] = us[(nb * nkc + step) * 9 + tap, 0, lane * 8: lane * 8 + 8]
                        for b in range(2):
                            A = np.zeros((32, 16))
                            for lane in range(64):
                                m = wm * 64 + b * 32 + (lane & 31)
                                il, r = m // bhw, m % bhw
                                qyl, qxl = r // bw, r % bw
                                p = (il * RH + qyl) * RW + qxl if il < ni else 0
                                A[lane & 31, 8 * (lane >> 5): 8 * (lane >> 5) + 8] = planes[lane >> 5, p + shpx]
                            acc[wm, wn, b, ph] += A @ B
        for wm in range(2):
            for wn in range(2):
                for b in range(2):
                    for row in range(32):
                        m = wm * 64 + b * 32 + row
                        il, r = m // bhw, m % bhw
                        qyl, qxl = r // bw, r % bw
                        n, qy, qx = gi * ni + il, by_i * bh + qyl, bx_i * bw + qxl
                        if il < ni and n < N and qy < H and qx < W:
                            for ph in range(4):
                                y[n, 2 * qy + (ph >> 1), 2 * qx + (ph & 1), n0 + wn * 32: n0 + wn * 32 + 32] = acc[wm, wn, b, ph, row]
    # direct ConvTranspose2d(k3, s2, p1, output_padding 1): y[2i - 1 + ky][2j - 1 + kx] += x[i][j] w[ky][kx]
    ref = np.zeros((N, 2 * H + 2, 2 * W + 2, cout))
    for ky in range(3):
        for kx in range(3):
            ref[:, ky:ky + 2 * H:2, kx:kx + 2 * W:2] += np.einsum("nhwc,co->nhwo", x, w[:, :, ky, kx])
    ref = ref[:, 1:2 * H + 1, 1:2 * W + 1]
    return float(np.abs(y - ref).max()), float(np.abs(ref).max())


def run_k3s(N, H, W, cin, bh, bw, ni, seed=0):
    """conv_k3s.hip: 3x3 stride-1 pad-1, 32 couts; raw slots (3 per thread) -> planes [kh][p][8], nine taps = nine pixel shifts, wave w =
    rows 64w .. 64w+63, weight fragments [(kc * 9 + tap)][lane * 8 + e] = w[lane & 31][kc*16 + 8*(lane>>5) + e][tap]"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, H, W, cin))
    w = rng.standard_normal((32, cin, 3, 3))
    nkc = cin // 16
    us = np.zeros((nkc * 9, 512))
    for i in range(32 * cin * 9):
        e, ln = i & 7, (i >> 3) & 63
        rest = i >> 9
        tap, kc = rest % 9, rest // 9
        us[kc * 9 + tap, ln * 8 + e] = w[ln & 31, kc * 16 + 8 * (ln >> 5) + e, tap // 3, tap % 3]
    RH, RW = bh + 2, bw + 2
    RP = ni * RH * RW
    assert RP <= 384 and bh * bw * ni <= 256
    bhw = bh * bw
    y = np.zeros((N, H, W, 32))
    nby, nbx, ngi = -(-H // bh), -(-W // bw), -(-N // ni)
    for bid in range(ngi * nby * nbx):
        bx_i, mb = bid % nbx, bid // nbx
        by_i, gi = mb % nby, mb // nby
        acc = np.zeros((4, 2, 32, 32))
        for step in range(nkc):
            planes = np.zeros((2, 384, 8))
            for t in range(256):
                for k in range(3):
                    e = t + 256 * k
                    kh, p = e & 1, e >> 1
                    if p < RP:
                        rxx, p2 = p % RW, p // RW
                        ry, il = p2 % RH, p2 // RH
                        n = gi * ni + il
                        iy, ix = by_i * bh + ry - 1, bx_i * bw + rxx - 1
                        if n < N and 0 <= iy < H and 0 <= ix < W:
                            planes[kh, p] = x[n, iy, ix, step * 16 + kh * 8: step * 16 + kh * 8 + 8]
            for wave in range(4):
                for tap in range(9):
                    shpx = (tap // 3) * RW + tap % 3
                    B = np.zeros((16, 32))
                    for lane in range(64):
                        B[8 * (lane >> 5): 8 * (lane >> 5) + 8, lane & 31] = us[step * 9 + tap, lane * 8: lane * 8 + 8]
                    for b in range(2):
                        A = np.zeros((32, 16))
                        for lane in range(64):
                            m = wave * 64 + b * 32 + (lane & 31)
                            il, r = m // bhw, m % bhw
                            p = (il * RH + r // bw) * RW + r % bw if il < ni else 0
                            A[lane & 31, 8 * (lane >> 5): 8 * (lane >> 5) + 8] = planes[lane >> 5, p + shpx]
                        acc[wave, b] += A @ B
        for m in range(256):
            il, r = m // bhw, m % bhw
            n, qy, qx = gi * ni + il, by_i * bh + r // bw, bx_i * bw + r % bw
            if il < ni and n < N and qy < H and qx < W:
                y[n, qy, qx] = acc[m // 64, (m % 64) // 32, m % 32]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = sum(np.einsum("nhwc,oc->nhwo", xp[:, ky:ky + H, kx:kx + W], w[:, :, ky, kx]) for ky in range(3) for kx in range(3))
    return float(np.abs(y - ref).max()), float(np.abs(ref).max())


def run_stem7s(N, H, W, cin, seed=0):
    """conv_stem7s.hip: 7x7 stride-1 pad-3, cin <= 8, 16 couts on the 16x16x32 MFMA (A: row = lane & 15, k-group = lane >> 4; C: row =
    4 (lane >> 4) + reg, col = lane & 15): region slots -> [p][8], chunk c = taps 4c .. 4c+3 (one per k-group), wave w = block rows
    4w .. 4w+3, weights [c][lane * 8 + e] = w[lane & 15][e][4c + (lane >> 4)] (zero past tap 48 / channel cin - 1)"""
    rng = np.random.default_rng(seed)
    x = np.zeros((N, H, W, 8))
    x[..., :cin] = rng.standard_normal((N, H, W, cin))
    w = rng.standard_normal((16, cin, 7, 7))
    us = np.zeros((13, 512))
    for i in range(13 * 512):
        e, ln, c = i & 7, (i >> 3) & 63, i >> 9
        tap = 4 * c + (ln >> 4)
        if e < cin and tap < 49:
            us[c, ln * 8 + e] = w[ln & 15, e, tap // 7, tap % 7]
    y = np.zeros((N, H, W, 16))
    nby, nbx = -(-H // 16), -(-W // 16)
    for bid in range(N * nby * nbx):
        bx_i, mb = bid % nbx, bid // nbx
        by_i, n = mb % nby, mb // nby
        y0, x0 = by_i * 16, bx_i * 16
        A_lds = np.zeros((512, 8))
        for t in range(256):
            for k in range(2):
                e = t + 256 * k
                if e < 484:
                    ry, rx = e // 22, e % 22
                    iy, ix = y0 + ry - 3, x0 + rx - 3
                    if 0 <= iy < H and 0 <= ix < W:
                        A_lds[e] = x[n, iy, ix]
        for wave in range(4):
            acc = np.zeros((4, 16, 16))
            for c in range(13):
                B = np.zeros((32, 16))
                for lane in range(64):
                    B[8 * (lane >> 4): 8 * (lane >> 4) + 8, lane & 15] = us[c, lane * 8: lane * 8 + 8]
                for j in range(4):
                    A = np.zeros((16, 32))
                    for lane in range(64):
                        g = lane >> 4
                        tap = 4 * c + g
                        tap = tap if tap < 49 else 0
                        dy = (tap * 37) >> 8
                        apix = (wave * 4) * 22 + (lane & 15)
                        A[lane & 15, 8 * g: 8 * g + 8] = A_lds[apix + j * 22 + dy * 22 + (tap - 7 * dy)]
                    acc[j] += A @ B
            for j in range(4):
                oy = y0 + wave * 4 + j
                for ox_l in range(16):
                    if oy < H and x0 + ox_l < W:
                        y[n, oy, x0 + ox_l] = acc[j, ox_l]
    xp = np.pad(x[..., :cin], ((0, 0), (3, 3), (3, 3), (0, 0)))
    ref = sum(np.einsum("nhwc,oc->nhwo", xp[:, ky:ky + H, kx:kx + W], w[:, :, ky, kx]) for ky in range(7) for kx in range(7))
    return float(np.abs(y - ref).max()), float(np.abs(ref).max())


if __name__ == "__main__":
    for args in ((2, 5, 6, 32, 64, 4, 6, 5), (3, 3, 3, 16, 128, 3, 3, 14), (1, 9, 7, 48, 64, 8, 8, 2), (5, 1, 1, 16, 64, 1, 1, 64)):
        print(args, "max err %.2e of %.1f" % run(*args))
    print("k3s", run_k3s(2, 9, 7, 32, 4, 8, 6), run_k3s(1, 17, 16, 16, 16, 16, 1))
    print("stem7s", run_stem7s(2, 20, 35, 6), run_stem7s(1, 5, 3, 8))
