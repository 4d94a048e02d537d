#!/usr/bin/env python
"""CPU check of the index algebra of tools/experiments/conv_wino4w.hip (the kernel itself was written without a GPU at hand):
the weight packing order, the LDS layout the transform waves write and the MFMA waves read (with its slot rotation), the
operand / result lane maps of v_mfma_f32_16x16x4_f32, the A^T M A rows of the per-lane epilogue and the (tile, cout) a lane
ends up holding are replayed in numpy, address formula by address formula as the kernel states them, for one workgroup item
(32 tiles x 64 couts) and compared with a direct 3x3 convolution in float64.  Also counts LDS bank conflicts of the 16-lane
groups the hardware forms for 128-bit accesses, and replays the cross-item software pipeline symbolically (pipeline_check).

    python tools/experiments/wino4w_layout_check.py
"""
import numpy as np

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
               [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
              [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
VPP, VBUF = 2 * 64 * 4, 18 * 2 * 64 * 4


def at_row(row, m):     # w4w_at<ROW>
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


def main():
    r = np.random.default_rng(0)
    cin, cout, ntile = 16, 128, 32
    nks = cin // 8
    w = r.normal(size=(cout, cin, 3, 3))
    d = r.normal(size=(ntile, cin, 6, 6))                       # raw 6x6 input patch of every tile
    # reference: 4x4 outputs of every tile by direct correlation
    ref = np.zeros((ntile, cout, 4, 4))
    for a in range(4):
        for b in range(4):
            ref[:, :, a, b] = np.einsum("tcij,ocij->to", d[:, :, a:a + 3, b:b + 3], w)
    V = np.einsum("ik,tckl,jl->ijtc", BT, d, BT).reshape(36, ntile, cin)      # [pos][tile][channel]
    U = np.einsum("ik,ockl,jl->ijoc", G, w, G).reshape(36, cout, cin)         # [pos][cout][channel]

    # ---- wino4w_pack_kernel: thread j -> (cout, cin), 36 stores
    u = np.full(cout * cin * 36, np.nan)
    for j in range(cout * cin):
        g1, ln = j & 1, (j >> 1) & 63
        blk = j >> 7
        kc, cbk = blk % nks, blk // nks
        co, ci = cbk * 16 + (ln & 15), kc * 8 + 2 * (ln >> 4) + g1
        dst = blk * 18 * 256 + ln * 4 + g1
        for pos in range(36):
            u[dst + (pos >> 1) * 256 + (pos & 1) * 2] = U[pos, co, ci]
    assert not np.isnan(u).any(), "the pack kernel leaves holes"

    for tile_n in range(cout // 64):
        n0 = tile_n * 64
        acc = np.zeros((8, 36, 64, 4))                          # [wave][pos][lane][reg]
        for kc in range(nks):
            # ---- transform waves: tf_cols_store
            Vs = np.full(VBUF, np.nan)
            wr_groups = []
            for wave in range(6):
                for jp in range(3):
                    for s in range(2):
                        addrs = []
                        for lane in range(64):
                            tl, q = lane >> 1, lane & 1
                            vwr = (wave * 3) * VPP + (tl >> 4) * 256 + ((2 * q) * 16 + (((tl & 15) + 8 * q) & 15)) * 4
                            dst = vwr + jp * VPP + 64 * s
                            va = V[6 * wave + 2 * jp, tl, kc * 8 + 4 * q:kc * 8 + 4 * q + 4]
                            vb = V[6 * wave + 2 * jp + 1, tl, kc * 8 + 4 * q:kc * 8 + 4 * q + 4]
                            Vs[dst:dst + 4] = [va[2 * s], va[2 * s + 1], vb[2 * s], vb[2 * s + 1]]
                            addrs.append(dst)
                        wr_groups.append(addrs)
            assert not np.isnan(Vs).any(), "the transform waves leave holes in V"
            # ---- MFMA waves
            rd_groups = []
            for wave in range(8):
                th, cq = wave & 1, wave >> 1
                cb16 = (n0 >> 4) + cq
                for pp in range(18):
                    a_op = np.zeros((64, 4))
                    b_op = np.zeros((64, 4))
                    addrs = []
                    for lane in range(64):
                        ab = th * 256 + ((lane & 48) | (((lane & 15) + ((lane >> 5) << 3)) & 15)) * 4 + pp * VPP
                        b_op[lane] = Vs[ab:ab + 4]
                        uo = ((cb16 * nks + kc) * 18 + pp) * 256 + lane * 4
                        a_op[lane] = u[uo:uo + 4]
                        addrs.append(ab)
                    rd_groups.append(addrs)
                    for f, pos in ((0, 2 * pp), (2, 2 * pp + 1), (1, 2 * pp), (3, 2 * pp + 1)):
                        A = np.zeros((16, 4))
                        B = np.zeros((4, 16))
                        for lane in range(64):
                            A[lane & 15, lane >> 4] = a_op[lane, f]
                            B[lane >> 4, lane & 15] = b_op[lane, f]
                        D = A @ B
                        for lane in range(64):
                            for reg in range(4):
                                acc[wave, pos, lane, reg] += D[4 * (lane >> 4) + reg, lane & 15]
        # ---- epilogue
        worst = 0.0
        for wave in range(8):
            th, cq = wave & 1, wave >> 1
            for lane in range(64):
                tile = th * 16 + (lane & 15)
                ch = n0 + cq * 16 + 4 * (lane >> 4)
                for oa in range(4):
                    T = [at_row(oa, [acc[wave, 6 * i + j, lane] for i in range(6)]) for j in range(6)]
                    for ob in range(4):
                        y = at_row(ob, T)
                        worst = max(worst, np.abs(y - ref[tile, ch:ch + 4, oa, ob]).max())
        print("cout block %d: max |kernel algebra - direct conv| = %.2e" % (tile_n, worst))
        assert worst < 1e-9

    # ---- LDS bank slots (16 bytes) of the 16-lane groups of a 128-bit access; the groups measured on gfx950 for ds_read_b128
    # (conv_wino4.hip, wino4_plane_geom) are {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in g] for g in groups[:2]]
    for name, accs in (("reads", rd_groups), ("writes", wr_groups)):
        conf = 0
        for addrs in accs:
            for g in groups:
                slots = [(addrs[lane] // 4) % 16 for lane in g]
                conf += len(slots) - len(set(slots))
        print("V %s: %d accesses, %d bank-slot collisions inside a 16-lane group" % (name, len(accs), conf))
        assert conf == 0


def pipeline_check():
    """symbolic replay of the kernel's control flow over the work items of one workgroup: raw blocks, V blocks and weight
    fragments carry (item, K-step[, position pair]) tags instead of data, moved exactly as the kernel moves them (prologue of the
    first item only, requests two steps ahead rolling over into the next item, buffer parity carried across items); every MFMA
    group must find its own item's block of its own step in the V buffer it reads and its own fragment in the ring"""
    for ring in (3, 6):
        for nsteps in (2, 3, 4, 5, 8):
            for nitems in (1, 2, 3, 4):
                Rs, V, bq = [None, None], [None, None], [None] * ring
                gpar, first = 0, True
                for it in range(nitems):
                    nxt = it + 1 if it + 1 < nitems else None          # no next item: out-of-range loads (tag None)
                    if first:
                        first = False
                        rawreg = (it, 0)
                        for i in range(ring):
                            bq[i] = (it, 0, i)
                        Rs[gpar] = rawreg
                        rawreg = (it, 1)
                        V[gpar] = Rs[gpar]                               # transform of block 0
                        Rs[gpar ^ 1] = rawreg
                    for step in range(nsteps):
                        buf = (gpar + step) & 1
                        tail = step + 2 >= nsteps
                        ub_next = (it, step + 1) if step + 1 < nsteps else (nxt, 0)
                        for s in range(18):
                            uc = bq[s % ring]
                            bq[s % ring] = (it, step, s + ring) if s < 18 - ring else (ub_next[0], ub_next[1], s + ring - 18)
                            if s == 0:
                                rawreg = (it, step + 2) if not tail else (nxt, step + 2 - nsteps)
                            elif s == 14:
                                V[buf ^ 1] = Rs[buf ^ 1]                 # the next step's block, transformed during this step
                            elif s == 16:
                                Rs[buf] = rawreg
                            assert V[buf] == (it, step), (ring, nsteps, nitems, it, step, V)
                            assert uc == (it, step, s), (ring, nsteps, nitems, it, step, s, uc)
                        want = (it, step + 1) if step + 1 < nsteps else (nxt, 0)
                        assert V[buf ^ 1] == want, (ring, nsteps, nitems, it, step, V, want)
                    gpar = (gpar + nsteps) & 1
    print("pipeline replay: every MFMA group meets its own item's V block and weight fragment (rings 3 / 6, 2-8 K-steps, 1-4 items)")


if __name__ == "__main__":
    main()
    pipeline_check()
