#!/usr/bin/env python
"""CPU emulation of csrc/conv_wino2s.hip at the level of its ADDRESS ARITHMETIC (no GPU): every index expression of the kernel -
DMA slot -> (image, row, column, channel quad), raw planes, transform task -> V bytes, A / B fragment addresses, accumulator
lane layout, staging tile, output pass - is restated here on numpy arrays that stand for the LDS buffers, the registers of the
64 lanes of a wave and the packed weight buffer, and the result is compared with a float64 convolution.  What it pins: the data
layout contract between wino2s_pack_kernel, the DMA, the transform, the MFMA operands and the epilogue.  What it does not model:
time (slots, barriers, waits).  Run by tests/test_wino2s_layout.py on ragged shapes and every tile-block shape.

    python tools/wino2s_emulate.py
"""
import numpy as np

BT, BC, KS = 64, 64, 16
VPOS = 2 * BT * 16
VPLANE = 16 * VPOS
CELLS = 240
RAWSLOTS = 4 * CELLS * 2
LDY = BC + 4
BLOCKS = [(8, 8, 1), (4, 8, 2), (8, 4, 2), (4, 4, 4), (2, 8, 4), (8, 2, 4), (4, 8, 1), (8, 4, 1), (2, 4, 8), (4, 2, 8), (4, 4, 2),
          (3, 3, 7), (2, 2, 12), (2, 2, 8), (3, 3, 3), (1, 4, 12), (4, 1, 8), (1, 2, 20), (2, 1, 13), (1, 1, 30), (6, 6, 1),
          (3, 6, 3), (6, 3, 3)]
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def ceil_div(a, b):
    return (a + b - 1) // b


def block_fits(b):
    bh, bw, ni = b
    return bh * bw * ni <= BT and ni * (2 * bh + 2) * (bw + 1) <= CELLS


def conflict_cost(bh, bw, ni, p, is_):
    """the kernel's cost of a (pitch, image stride): padding + 64 per colliding pair in a ds_read_b128 lane group"""
    bhw = bh * bw
    cost = 0
    for half in range(2):
        for gq in range(4):
            cnt = [0] * 16
            for k in range(16):
                lane = GROUPS[gq & 1][k] + 32 * (gq >> 1)
                tl, q = half * 32 + (lane >> 1), lane & 1
                il, r = tl // bhw, tl % bhw
                ilc = il if il < ni else 0
                cnt[((ilc * is_ + 2 * (r // bw) * p + r % bw) * 2 + q) & 15] += 1
            cost += sum((c - 1) * 64 * c for c in cnt if c > 1)
    return cost


def plane_geom(bh, bw, ni):
    """wino2s_plane_geom (conv_wino2s.hip)"""
    RH = 2 * bh + 2
    best, pitch, istride = 1 << 30, bw + 1, RH * (bw + 1)
    for p in range(bw + 1, bw + 9):
        for is_ in range(RH * p, RH * p + 16):
            if ni * is_ > CELLS:
                break
            cost = (p - bw - 1) + (is_ - RH * p) + conflict_cost(bh, bw, ni, p, is_)
            if cost < best:
                best, pitch, istride = cost, p, is_
    return pitch, istride


def pick_block(N, TH, TW):
    best, best_cost = (1, 1, 1), 1e300
    for b in BLOCKS:
        if not block_fits(b):
            continue
        bh, bw, ni = b
        items = ceil_div(TH, bh) * ceil_div(TW, bw) * ceil_div(N, ni)
        halo = (2 * bh + 2) * (2 * bw + 2) / (4.0 * bh * bw)
        cost = items * (1.0 + 0.05 * halo)
        if cost < best_cost:
            best_cost, best = cost, b
    return best


def bf16_round(x):
    """float32 -> nearest-even bfloat16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_round(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_round(r1)
    l = bf16_round((r1 - m).astype(np.float32))
    return h, m, l


def pack_weights(w):
    """wino_pack (fp64 transform, one rounding to fp32) + wino2s_pack_kernel: u[nb][kc][pos][plane][lane][e] (float32 holding bf16)"""
    cout, cin = w.shape[:2]
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    U = np.einsum("ia,ocab,jb->ijoc", G, w.astype(np.float64), G).astype(np.float32)      # [4][4][cout][cin]
    u = np.zeros((cout // 32, cin // 16, 16, 3, 64, 8), np.float32)
    for ln in range(64):
        for e in range(8):
            co = np.arange(cout // 32) * 32 + (ln & 31)
            ci = np.arange(cin // 16) * 16 + 8 * (ln >> 5) + e
            v = U[:, :, co][:, :, :, ci].reshape(16, len(co), len(ci)).transpose(1, 2, 0)     # [nb][kc][pos]
            h, m, l = split3(v)
            u[:, :, :, 0, ln, e], u[:, :, :, 1, ln, e], u[:, :, :, 2, ln, e] = h, m, l
    return u


def emulate(x, w, scale, shift, res=None, act="relu", block=None):
    """x [N, H, W, cin] NHWC float32, w [cout, cin, 3, 3] -> y [N, H, W, cout] exactly as the kernel's data flow computes it
    (piece products accumulated in float64: the emulation checks indices, not rounding)"""
    N, H, W, cin = x.shape
    cout = w.shape[0]
    assert cin % KS == 0 and cout % BC == 0
    TH, TW = (H + 1) // 2, (W + 1) // 2
    bh, bw, ni = block or pick_block(N, TH, TW)
    assert block_fits((bh, bw, ni))
    nby, nbx, ngi = ceil_div(TH, bh), ceil_div(TW, bw), ceil_div(N, ni)
    RH, RW = 2 * bh + 2, 2 * bw + 2
    pitch, istride = plane_geom(bh, bw, ni)
    assert ni * istride <= CELLS and pitch >= bw + 1 and istride >= RH * pitch
    nkc, tiles_n = cin // KS, cout // BC
    nsteps2 = nkc + (nkc & 1)
    u = pack_weights(w)
    y = np.full((N, H, W, cout), np.nan, np.float32)
    bhw = bh * bw
    lane = np.arange(64)
    for bid in range(ngi * nby * nbx * tiles_n):
        tile_n = bid % tiles_n
        mbk = bid // tiles_n
        bx_i, mbk = mbk % nbx, mbk // nbx
        by_i, gi = mbk % nby, mbk // nby
        n0 = tile_n * BC
        # tile table
        opix, oflag = np.full(BT, -1), np.zeros(BT, int)
        for t in range(BT):
            il, r = t // bhw, t % bhw
            tyl, txl = r // bw, r % bw
            n, ty, tx = gi * ni + il, by_i * bh + tyl, bx_i * bw + txl
            if il < ni and n < N and ty < TH and tx < TW:
                opix[t] = (n * H + 2 * ty) * W + 2 * tx
                oflag[t] = (1 if 2 * tx + 1 < W else 0) | (2 if 2 * ty + 1 < H else 0)
        acc = np.zeros((8, 4, 2, 64, 16), np.float64)      # [wave][j][mb][lane][r]
        for step in range(nsteps2):
            # ---- DMA: slot e of the raw buffer (16-byte slots = 4 floats)
            raw = np.zeros((RAWSLOTS, 4), np.float32)
            for e in range(RAWSLOTS):
                P, rem = e // (2 * CELLS), e % (2 * CELLS)
                cell, qq = rem >> 1, rem & 1
                il, r2 = cell // istride, cell % istride
                ry, cx = r2 // pitch, r2 % pitch
                rxx = 2 * cx + (P & 1)
                n = gi * ni + il
                iy, ix = 2 * by_i * bh - 1 + ry, 2 * bx_i * bw - 1 + rxx
                ok = step < nkc and il < ni and ry < RH and rxx < RW and n < N and 0 <= iy < H and 0 <= ix < W
                if ok:
                    quad = (P >> 1) * 2 + qq
                    raw[e] = x[n, iy, ix, step * KS + quad * 4: step * KS + quad * 4 + 4]
            rawb = raw.reshape(-1)                          # float index = byte / 4
            # ---- transform: V bytes as bf16 values in a float32 array indexed by (byte offset / 2)
            V = np.zeros(3 * VPLANE // 2, np.float32)
            for wave in range(8):
                g, wn, row = wave >> 2, (wave >> 1) & 1, 2 * (wave >> 2) + (wave & 1)
                ra = 0 if row == 0 else (2 if row == 2 else 1)
                rb = 2 if row == 0 else (2 if row == 1 else (1 if row == 2 else 3))
                sg = 1.0 if row == 1 else -1.0
                q = lane & 1
                tl = wn * 32 + (lane >> 1)
                il, r = tl // bhw, tl % bhw
                tyl, txl = r // bw, r % bw
                ilc = np.where(il < ni, il, 0)
                cell0 = ilc * istride + 2 * tyl * pitch + txl
                tf_a = (cell0 + ra * pitch) * 32 + q * 16
                tf_b = (cell0 + rb * pitch) * 32 + q * 16
                vwr = (4 * row) * VPOS + (wn * 32 + (lane >> 1)) * 16 + q * 8
                for kh in range(2):
                    da = []
                    for c in range(4):
                        po = ((2 * kh + (c & 1)) * CELLS + (c >> 1)) * 32
                        va = np.stack([rawb[(tf_a + po) // 4 + e] for e in range(4)], 1)
                        vb = np.stack([rawb[(tf_b + po) // 4 + e] for e in range(4)], 1)
                        da.append((sg * vb + va).astype(np.float32))
                    for j in range(4):
                        v = [da[0] - da[2], da[1] + da[2], da[2] - da[1], da[1] - da[3]][j].astype(np.float32)
                        pieces = split3(v)
                        dst = vwr + j * VPOS + kh * 1024
                        for p in range(3):
                            for e in range(4):
                                V[(dst + p * VPLANE) // 2 + e] = pieces[p][:, e]
            # ---- MFMAs
            for wave in range(8):
                wn, row = (wave >> 1) & 1, 2 * (wave >> 2) + (wave & 1)
                nb = (n0 >> 5) + wn
                ard = (4 * row) * VPOS + (lane >> 5) * 1024 + (lane & 31) * 16
                for j in range(4):
                    pos = 4 * row + j
                    for mb in range(2):
                        A = [np.stack([V[(ard + p * VPLANE + j * VPOS + mb * 512) // 2 + e] for e in range(8)], 1) for p in range(3)]
                        B = [u[nb, step, pos, p] if step < nkc else np.zeros((64, 8), np.float32) for p in range(3)]
                        D = np.zeros((32, 32))
                        for pa, pb in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
                            Am = np.zeros((32, 16))
                            Bm = np.zeros((16, 32))
                            for l in range(64):
                                Am[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = A[pa][l]
                                Bm[8 * (l >> 5): 8 * (l >> 5) + 8, l & 31] = B[pb][l]
                            D += Am @ Bm
                        for r in range(16):
                            acc[wave, j, mb, :, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
        # ---- epilogue
        KROW = 32 * 2 * LDY
        for rnd in range(2):
            S = np.zeros(4 * KROW)
            for wave in range(8):
                wn, row = (wave >> 1) & 1, 2 * (wave >> 2) + (wave & 1)
                base = row * KROW + wn * 32 + (lane & 31)
                for r in range(16):
                    m0, m1, m2, m3 = (acc[wave, j, rnd, :, r] for j in range(4))
                    tl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                    S[base + (tl * 2 + 0) * LDY] = (m0 + m1) + m2
                    S[base + (tl * 2 + 1) * LDY] = (m1 - m2) - m3
            CG = BC // 4
            for t in range(512):
                c4 = t % CG
                ch = n0 + c4 * 4
                for i in range(4):
                    id_ = i * 512 + t
                    pxl = id_ // CG
                    tile, oa, ob = 32 * rnd + (pxl >> 2), (pxl >> 1) & 1, pxl & 1
                    ok = opix[tile] >= 0 and (oa == 0 or oflag[tile] & 2) and (ob == 0 or oflag[tile] & 1)
                    if not ok:
                        continue
                    pix = opix[tile] + oa * W + ob
                    src = ((pxl >> 2) * 2 + (pxl & 1)) * LDY + c4 * 4 + oa * KROW
                    p0, p1, p2 = S[src: src + 4], S[src + KROW: src + KROW + 4], S[src + 2 * KROW: src + 2 * KROW + 4]
                    m = (p0 + p1) + p2 if oa == 0 else (p0 - p1) - p2
                    v = m * scale[ch: ch + 4] + shift[ch: ch + 4]
                    if res is not None:
                        v = v + res.reshape(-1, cout)[pix, ch: ch + 4]
                    if act == "relu":
                        v = np.maximum(v, 0)
                    elif act == "leaky":
                        v = np.where(v > 0, v, 0.01 * v)
                    y.reshape(-1, cout)[pix, ch: ch + 4] = v
    return y


def reference(x, w, scale, shift, res=None, act="relu"):
    import torch
    import torch.nn.functional as F
    z = F.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(w).double(), padding=1).permute(0, 2, 3, 1).numpy()
    z = z * scale + shift
    if res is not None:
        z = z + res
    if act == "relu":
        z = np.maximum(z, 0)
    elif act == "leaky":
        z = np.where(z > 0, z, 0.01 * z)
    return z


def check(N, H, W, cin, cout, seed=0, with_res=False, act="relu", block=None):
    r = np.random.default_rng(seed)
    x = r.standard_normal((N, H, W, cin)).astype(np.float32)
    w = (r.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    scale, shift = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32) * 0.1
    res = r.standard_normal((N, H, W, cout)).astype(np.float32) if with_res else None
    got = emulate(x, w, scale, shift, res, act, block)
    ref = reference(x, w, scale, shift, res, act)
    assert not np.isnan(got).any(), "an output pixel was never written"
    err = float(np.abs(got - ref).max())
    return err


if __name__ == "__main__":
    for args in ((2, 6, 6, 16, 64), (3, 5, 7, 32, 64), (1, 16, 16, 16, 128)):
        print(args, "max |err| vs float64 conv: %.3e" % check(*args))
