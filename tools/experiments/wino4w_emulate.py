#!/usr/bin/env python
"""Thread-level numpy emulation of tools/experiments/conv_wino4w.hip for ONE persistent workgroup walking several work items
of a small convolution: the host's block choice and raw-plane geometry, the raw slot -> LDS plane mapping, per-item global
offsets with their padding, the six transform waves (row / column passes with the kernel's coefficients and LDS offsets), the V
layout, the 16x16x4 MFMA lane maps, the cross-item software pipeline with its buffer parities and tile-table parity, and the
per-lane epilogue with its border flags - statement by statement as the kernel has them, on float64 arrays, compared with a
direct 3x3 convolution (+ scale / shift / residual / ReLU).  It checks the kernel's LOGIC; what only a GPU can check (the
compiler's waits, LDS timing, speed) stays open.

    python tools/experiments/wino4w_emulate.py
"""
import numpy as np

KS, BC, NRAW = 8, 64, 3
# set by configure(): 512-thread workgroup over 32 tiles (W4W_HALF=0) or 256-thread workgroup over 16 tiles (W4W_HALF=1)
HALF, THREADS, BT, VPP, VBUF, RAW4, PS, QS = 0, 512, 32, 512, 9216, 1536, 186, 744


def configure(half):
    global HALF, THREADS, BT, VPP, VBUF, RAW4, PS, QS
    HALF = half
    THREADS, BT = (256, 16) if half else (512, 32)
    VPP = (BT // 16) * 64 * 4
    VBUF = 18 * VPP
    RAW4 = THREADS * NRAW
    PS = 90 if half else 186
    QS = 4 * PS
    assert 2 * QS <= RAW4 and PS % 8 == 2 and QS % 16 == 8


BLOCKS = [(4, 8, 1), (8, 4, 1), (4, 4, 2), (2, 8, 2), (8, 2, 1), (2, 4, 3), (4, 2, 3), (3, 3, 3),
          (2, 2, 6), (2, 3, 4), (3, 2, 4), (1, 4, 6), (4, 1, 5), (1, 2, 10), (2, 1, 9), (1, 1, 15),
          (4, 4, 1), (2, 8, 1), (2, 4, 2), (4, 2, 2), (2, 2, 4), (2, 3, 2), (3, 2, 2), (1, 4, 4), (4, 1, 3),
          (1, 2, 8), (2, 1, 8), (1, 1, 16), (3, 3, 1), (2, 2, 3)]
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
TF = {0: (0, 0., 0, 4., 2, -5., 4), 1: (1, -4., 2, -4., 3, 1., 4), 2: (1, 4., 2, -4., 3, -1., 4),
      3: (1, -2., 2, -1., 3, 2., 4), 4: (1, 2., 2, -1., 3, -2., 4), 5: (1, 0., 1, 4., 3, -5., 5)}
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
              [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])


def ceil_div(a, b):
    return (a + b - 1) // b


def block_fits(b):
    bh, bw, ni = b
    return bh * bw * ni <= BT and ni * (4 * bh + 2) * (4 * bw + 2) * 2 <= RAW4 and ni * (4 * bh + 2) * (bw + 1) <= PS


def pick_block(N, TH, TW):
    best, best_cost = (1, 1, 1), 1e300
    for b in BLOCKS:
        if not block_fits(b):
            continue
        items = ceil_div(TH, b[0]) * ceil_div(TW, b[1]) * ceil_div(N, b[2])
        cost = items * (1.0 + 0.05 * (4 * b[0] + 2) * (4 * b[1] + 2) / (16.0 * b[0] * b[1]))
        if cost < best_cost:
            best_cost, best = cost, b
    return best


def plane_geom(b):
    bh, bw, ni = b
    RH, bhw = 4 * bh + 2, bh * bw
    best, pitch, istride = 1 << 30, bw + 1, RH * (bw + 1)
    for p in range(bw + 1, bw + 5):
        for is_ in range(RH * p, RH * p + 16):
            if ni * is_ > PS:
                break
            cost = (p - bw - 1) + (is_ - RH * p)
            for g in range(4):
                cnt = [0] * 16
                for k in range(16):
                    lane = GROUPS[g & 1][k] + 32 * (g >> 1)
                    tl, q = ((lane & 31) >> 1 if HALF else lane >> 1), lane & 1
                    il, r = tl // bhw, tl % bhw
                    ilc = il if il < ni else 0
                    cnt[(q * QS + ilc * is_ + 4 * (r // bw) * p + r % bw) & 15] += 1
                cost += sum((c - 1) * 64 * c for c in cnt if c > 1)
            if cost < best:
                best, pitch, istride = cost, p, is_
    return pitch, istride


def at_row(row, m):
    m0, m1, m2, m3, m4, m5 = m
    if row == 0:
        return (m0 + (m1 + m2)) + (m3 + m4)
    if row == 1:
        d = m3 - m4
        return (m1 - m2) + (d + d)
    if row == 2:
        return (m1 + m2) + 4 * (m3 + m4)
    d = m3 - m4
    return ((m1 - m2) + 4 * (d + d)) + m5


def run(N, H, W, cin, cout, ring, workgroups, seed, half=0):
    configure(half)
    NW = THREADS // 64
    r = np.random.default_rng(seed)
    x_cs, y_cs = cin + 4, cout + 8
    x = r.normal(size=(N * H * W, x_cs))
    w = r.normal(size=(cout, cin, 3, 3))
    res = r.normal(size=(N * H * W, y_cs))
    scale, shift = r.normal(size=cout), r.normal(size=cout)
    y = np.full((N * H * W, y_cs), np.nan)
    # ---- host: wino4_launch
    TH, TW = (H + 3) // 4, (W + 3) // 4
    bh, bw, ni = pick_block(N, TH, TW)
    nby, nbx, ngi = ceil_div(TH, bh), ceil_div(TW, bw), ceil_div(N, ni)
    RH, RW = 4 * bh + 2, 4 * bw + 2
    R4 = ni * RH * RW
    pitch, istride = plane_geom((bh, bw, ni))
    inv_rw, inv_rh = np.float32(1.0) / np.float32(RW), np.float32(1.0) / np.float32(RH)
    nks, tiles_n = cin // 8, cout // BC
    total = ngi * nby * nbx * tiles_n
    grid = min(ceil_div(total, 8) * 8, workgroups)
    bhw = bh * bw
    # ---- wino4w_pack_kernel
    U = np.einsum("ik,ockl,jl->ijoc", G, w, G).reshape(36, cout, cin)
    u = np.full(cout * cin * 36, np.nan)
    for j in range(cout * cin):
        g1, ln = j & 1, (j >> 1) & 63
        blk = j >> 7
        kc, cbk = blk % nks, blk // nks
        dst = blk * 18 * 256 + ln * 4 + g1
        for pos in range(36):
            u[dst + (pos >> 1) * 256 + (pos & 1) * 2] = U[pos, cbk * 16 + (ln & 15), kc * 8 + 2 * (ln >> 4) + g1]

    def div_recip(num, inv):      # (int)(((float)num + 0.5f) * inv), float32 arithmetic as on the device
        return int(np.float32(np.float32(num) + np.float32(0.5)) * inv)

    T = np.arange(THREADS)
    per, gw = ceil_div(total, 8), grid >> 3
    for block in range(grid):
        xcd = block & 7
        # item-invariant: rst
        rst = np.full((NRAW, THREADS), -1)
        geo = {}
        for k in range(NRAW):
            for t in T:
                e = t + THREADS * k
                q, pix = (e >> 3) & 1, (e >> 4) * 8 + (e & 7)
                if pix < R4:
                    p2 = div_recip(pix, inv_rw)
                    rxx = pix - p2 * RW
                    il = div_recip(p2, inv_rh)
                    ry = p2 - il * RH
                    assert 0 <= rxx < RW and 0 <= ry < RH and il < ni, "reciprocal division off"
                    rst[k, t] = q * QS + (rxx & 3) * PS + il * istride + ry * pitch + (rxx >> 2)      # in 16-byte entries
                    geo[(k, t)] = (q, il, ry, rxx)

        def item_goff(gi, by_i, bx_i, valid):
            g = np.full((NRAW, THREADS, 2), -1)      # (pixel row of x, channel offset) or -1 = out of range
            if valid:
                for (k, t), (q, il, ry, rxx) in geo.items():
                    n, iy, ix = gi * ni + il, 4 * by_i * bh - 1 + ry, 4 * bx_i * bw - 1 + rxx
                    if n < N and 0 <= iy < H and 0 <= ix < W:
                        g[k, t] = ((n * H + iy) * W + ix, q * 4)
            return g

        def coords(bid):
            tile_n = bid % tiles_n
            mb = bid // tiles_n
            bx_i = mb % nbx
            mb //= nbx
            return tile_n, bx_i, mb % nby, mb // nby

        def raw_gload(goff, step):
            out = np.zeros((NRAW, THREADS, 4))
            for k in range(NRAW):
                for t in T:
                    p, c = goff[k, t]
                    if p >= 0 and c + step * KS + 4 <= x_cs:
                        out[k, t] = x[p, c + step * KS:c + step * KS + 4]
            return out

        Rs = np.full((2, RAW4, 4), np.nan)
        Vs = np.full((2, VBUF), np.nan)
        tab = np.zeros((2, 2, BT), dtype=np.int64)

        def raw_store(buf, rawreg):
            for k in range(NRAW):
                for t in T:
                    if rst[k, t] >= 0:
                        Rs[buf, rst[k, t]] = rawreg[k, t]

        def transform(rbuf, vbuf):       # the transform waves: tf_rows for the six columns, then both halves of tf_cols_store
            for wave in range(3 if HALF else 6):
                for lane in range(64):
                    trow = 2 * wave + (lane >> 5) if HALF else wave
                    ra, ca, rb, cb, rc, cc, rd = TF[trow]
                    q, tl = lane & 1, ((lane & 31) >> 1 if HALF else lane >> 1)
                    il, rr_ = tl // bhw, tl % bhw
                    tyl, txl = rr_ // bw, rr_ % bw
                    ilc = il if il < ni else 0
                    tf_base = q * QS + ilc * istride + 4 * tyl * pitch + txl
                    rows = []
                    for c in range(6):
                        co = (c & 3) * PS + (c >> 2)
                        va, vb = Rs[rbuf, tf_base + ra * pitch + co], Rs[rbuf, tf_base + rb * pitch + co]
                        vc, vd = Rs[rbuf, tf_base + rc * pitch + co], Rs[rbuf, tf_base + rd * pitch + co]
                        rows.append(ca * va + (cb * vb + (cc * vc + vd)))
                    rr = rows
                    p = -4. * rr[2] + rr[4]
                    qq = 4. * rr[1] - rr[3]
                    p2 = rr[4] - rr[2]
                    q2 = 2. * (rr[3] - rr[1])
                    v = [4. * rr[0] + (-5. * rr[2] + rr[4]), p - qq, p + qq, p2 + q2, p2 - q2, 4. * rr[1] + (-5. * rr[3] + rr[5])]
                    vwr = (trow * 3) * VPP + (tl >> 4) * 256 + ((2 * q) * 16 + (((tl & 15) + 8 * q) & 15)) * 4
                    for h in range(2):
                        for jp in range(3):
                            dst = vwr + 64 * h + jp * VPP
                            Vs[vbuf, dst:dst + 4] = [v[2 * jp][2 * h], v[2 * jp][2 * h + 1], v[2 * jp + 1][2 * h], v[2 * jp + 1][2 * h + 1]]

        item_parity, gpar, first = 0, 0, True
        bq = [None] * ring                      # per-wave rings of (ublk, pp) requests; data fetched when consumed
        bq = [[None] * ring for _ in range(NW)]
        jw = block >> 3
        while jw < per:
            bid = xcd * per + jw
            if bid >= total:
                break
            tile_n, bx_i, by_i, gi = coords(bid)
            n0 = tile_n * BC
            bid_n = xcd * per + jw + gw
            has_next = (jw + gw < per) and (bid_n < total)
            tile_nn, bx_n, by_n, gi_n = coords(bid_n if has_next else bid)
            opix, oflag = tab[item_parity, 0], tab[item_parity, 1]
            item_parity ^= 1
            for t in range(BT):
                il, r_ = t // bhw, t % bhw
                tyl, txl = r_ // bw, r_ % bw
                n, ty, tx = gi * ni + il, by_i * bh + tyl, bx_i * bw + txl
                o, f = -1, 0
                if il < ni and n < N and ty < TH and tx < TW:
                    o = (n * H + 4 * ty) * W + 4 * tx
                    for k in range(4):
                        f |= ((1 << k) if 4 * ty + k < H else 0) | ((16 << k) if 4 * tx + k < W else 0)
                opix[t], oflag[t] = o, f
            goff = item_goff(gi, by_i, bx_i, True)
            cqs = [wv if HALF else wv >> 1 for wv in range(NW)]
            ths = [0 if HALF else wv & 1 for wv in range(NW)]
            ublk0 = [((n0 >> 4) + cqs[wv]) * nks for wv in range(NW)]
            ublk_n = [(((tile_nn * BC) >> 4) + cqs[wv]) * nks for wv in range(NW)]
            acc = np.zeros((NW, 36, 64, 4))
            if first:
                first = False
                rawreg = raw_gload(goff, 0)
                for wv in range(NW):
                    for i in range(ring):
                        bq[wv][i] = (ublk0[wv], i)
                raw_store(gpar, rawreg)
                rawreg = raw_gload(goff, 1 if nks > 1 else 0)
                transform(gpar, gpar)
                raw_store(gpar ^ 1, rawreg)
            for step in range(nks):
                buf = (gpar + step) & 1
                tail = step + 2 >= nks
                for s in range(18):
                    for wv in range(NW):
                        th = ths[wv]
                        ub_next = ublk0[wv] + step + 1 if step + 1 < nks else ublk_n[wv]
                        uc_req = bq[wv][s % ring]
                        bq[wv][s % ring] = (ublk0[wv] + step, s + ring) if s < 18 - ring else (ub_next, s + ring - 18)
                        assert uc_req == (ublk0[wv] + step, s), "weight ring out of step"
                        a_op = np.zeros((64, 4))
                        b_op = np.zeros((64, 4))
                        for lane in range(64):
                            ab = th * 256 + ((lane & 48) | (((lane & 15) + ((lane >> 5) << 3)) & 15)) * 4 + s * VPP
                            b_op[lane] = Vs[buf, ab:ab + 4]
                            uo = uc_req[0] * 18 * 256 + uc_req[1] * 256 + lane * 4
                            a_op[lane] = u[uo:uo + 4]
                        for f_, pos in ((0, 2 * s), (2, 2 * s + 1), (1, 2 * s), (3, 2 * s + 1)):
                            A = np.zeros((16, 4))
                            B = np.zeros((4, 16))
                            A[np.arange(64) & 15, np.arange(64) >> 4] = a_op[:, f_]
                            B[np.arange(64) >> 4, np.arange(64) & 15] = b_op[:, f_]
                            D = A @ B
                            for reg in range(4):
                                acc[wv, pos, :, reg] += D[4 * (np.arange(64) >> 4) + reg, np.arange(64) & 15]
                    if s == 0:
                        rawreg = raw_gload(goff, step + 2) if not tail else raw_gload(item_goff(gi_n, by_n, bx_n, has_next), step + 2 - nks)
                    elif s == 15:
                        transform(buf ^ 1, buf ^ 1)
                    elif s == 16:
                        raw_store(buf, rawreg)
            gpar = (gpar + nks) & 1
            # ---- epilogue
            for wv in range(NW):
                th, cq = ths[wv], cqs[wv]
                for lane in range(64):
                    tile = th * 16 + (lane & 15)
                    op, fl = opix[tile], oflag[tile]
                    ch = n0 + cq * 16 + 4 * (lane >> 4)
                    for oa in range(4):
                        Tm = [at_row(oa, [acc[wv, 6 * i + j, lane] for i in range(6)]) for j in range(6)]
                        for ob in range(4):
                            if op >= 0 and (fl >> oa) & 1 and (fl >> (4 + ob)) & 1:
                                pix = op + oa * W + ob
                                xv = at_row(ob, Tm) * scale[ch:ch + 4] + shift[ch:ch + 4] + res[pix, ch:ch + 4]
                                assert np.isnan(y[pix, ch:ch + 4]).all(), "an output element is written twice"
                                y[pix, ch:ch + 4] = np.maximum(xv, 0.)
            jw += gw
    # ---- reference
    xi = x[:, :cin].reshape(N, H, W, cin)
    xp = np.zeros((N, H + 2, W + 2, cin))
    xp[:, 1:-1, 1:-1] = xi
    ref = np.zeros((N, H, W, cout))
    for a in range(3):
        for b in range(3):
            ref += np.einsum("nhwc,oc->nhwo", xp[:, a:a + H, b:b + W], w[:, :, a, b])
    ref = np.maximum(ref * scale + shift + res[:, :cout].reshape(N, H, W, cout), 0.)
    got = y[:, :cout].reshape(N, H, W, cout)
    assert not np.isnan(got).any(), "some output element was never written"
    assert np.isnan(y[:, cout:]).all(), "the kernel wrote into the channel padding of y"
    err = np.abs(got - ref).max()
    print("%d threads: N=%d %dx%d cin=%d cout=%d block %s items %d on %d workgroups, ring %d: max |emulated kernel - direct conv| = %.2e"
          % (THREADS, N, H, W, cin, cout, (bh, bw, ni), total, grid, ring, err))
    assert err < 1e-9


def main():
    # eight workgroups (one per XCD slot, gw = 1): every workgroup walks total / 8 items through the cross-item pipeline
    run(N=24, H=16, W=16, cin=16, cout=128, ring=3, workgroups=8, seed=0)   # 3 items per workgroup, even K-step count
    run(N=60, H=12, W=12, cin=24, cout=64, ring=3, workgroups=8, seed=1)    # odd K-step count: the buffer parity flips per item
    run(N=40, H=6, W=10, cin=16, cout=64, ring=6, workgroups=8, seed=2)     # ragged tiles (H, W not multiples of 4), ring of 6
    run(N=7, H=8, W=8, cin=16, cout=128, ring=3, workgroups=8, seed=3)      # workgroups with 1 and 0 items, a ragged image group
    # W4W_HALF=1: 256-thread workgroups over 16 tiles, three transform waves with two rows each
    run(N=12, H=16, W=16, cin=16, cout=128, ring=3, workgroups=8, seed=4, half=1)
    run(N=20, H=12, W=12, cin=24, cout=64, ring=3, workgroups=8, seed=5, half=1)
    run(N=21, H=6, W=10, cin=16, cout=64, ring=3, workgroups=8, seed=6, half=1)


if __name__ == "__main__":
    main()
