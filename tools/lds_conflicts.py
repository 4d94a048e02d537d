#!/usr/bin/env python3
"""LDS bank-conflict calculator for gfx950 layouts (CPU only): cycles of the LDS array per wave-instruction under the lane groups
and bank maps of MI355X_MICROARCH.md ("LDS" table), for an address function lane -> byte address.

A wave64 access is served in fixed lane groups, one LDS cycle per group when every bank is touched by at most one distinct dword
address; each further distinct address on a busy bank adds a cycle.  This is the arithmetic behind the row strides and swizzles
of the kernels in wav2lip_amd/csrc (and the tool that found the 2-way store conflict of the padded rows in the three-plane
kernel); `python tools/lds_conflicts.py` prints the in-tree layouts, `--stride/--swizzle` explores others.

    python tools/lds_conflicts.py [--stride BYTES] [--swizzle]
"""
import argparse

_R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = {
    # instruction: (lane groups, bytes per lane, number of banks in the bank map)
    "ds_read_b32": ([list(range(0, 32)), list(range(32, 64))], 4, 32),
    "ds_read_b64": ([list(range(0, 32)), list(range(32, 64))], 8, 64),
    "ds_read_b128": (_R128 + [[l + 32 for l in g] for g in _R128], 16, 64),
    "ds_write_b32": ([list(range(0, 32)), list(range(32, 64))], 4, 32),
    "ds_write_b64": ([list(range(i, i + 16)) for i in range(0, 64, 16)], 8, 32),
    "ds_write_b128": ([list(range(i, i + 8)) for i in range(0, 64, 8)], 16, 32),
}
IDEAL = {k: len(v[0]) for k, v in GROUPS.items()}


def cycles(instr, addr):
    """LDS-array cycles of one wave-instruction; addr(lane) -> byte address (aligned to the access size)"""
    groups, width, nbanks = GROUPS[instr]
    total = 0
    for g in groups:
        per_bank = {}
        for lane in g:
            a = addr(lane)
            assert a % min(width, 16) == 0 or width == 8 and a % 8 == 0, (instr, lane, a)
            for d in range(width // 4):
                dword = a // 4 + d
                per_bank.setdefault(dword % nbanks, set()).add(dword)
        total += max(len(v) for v in per_bank.values())
    return total


def rows_layout(stride, swizzle):
    """K-major operand tile of conv_igemm_bf16_kernel: 32 bf16 (64 B) of K per row; fragment read = lane l takes the 16-byte chunk
    (l >> 5) + 2 kq of row l & 31; staging store = lane t writes 8 B (kg = t & 7) or 16 B (chunk t & 3) of row t >> 3 / t >> 2"""
    def slot(row, c):
        return (c ^ ((row >> 2) & 3)) if swizzle else c
    out = {}
    out["fragment ds_read_b128"] = max(cycles("ds_read_b128", lambda l, kq=kq, b=b: ((l & 31) + b) * stride + slot((l & 31) + b, (l >> 5) + 2 * kq) * 16)
                                       for kq in (0, 1) for b in (0, 32, 64, 96))
    out["staging ds_write_b64"] = max(cycles("ds_write_b64", lambda l, r0=r0: ((l >> 3) + r0) * stride + slot((l >> 3) + r0, (l & 7) >> 1) * 16 + (l & 1) * 8)
                                      for r0 in (0, 8, 16, 24))
    out["staging ds_write_b128"] = max(cycles("ds_write_b128", lambda l, r0=r0: ((l >> 2) + r0) * stride + slot((l >> 2) + r0, l & 3) * 16)
                                       for r0 in (0, 16))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stride", type=int, default=0, help="row stride in bytes of a layout to evaluate (with or without --swizzle)")
    ap.add_argument("--swizzle", action="store_true", help="16-byte chunk c of row r stored at slot c ^ ((r >> 2) & 3)")
    args = ap.parse_args()
    cases = [("one-plane kernel (round 2): 80-byte rows (64 + 16 pad)", 80, False),
             ("three-plane (split-operand) kernel: 64-byte rows, chunk swizzle", 64, True),
             ("64-byte rows without the swizzle", 64, False)]
    if args.stride:
        cases = [("stride %d%s" % (args.stride, ", swizzled" if args.swizzle else ""), args.stride, args.swizzle)]
    for name, stride, sw in cases:
        r = rows_layout(stride, sw)
        print(name)
        for k, v in r.items():
            ideal = IDEAL[k.split()[1]]
            print("    %-24s %2d LDS cycles per wave-instruction (conflict-free: %d)" % (k, v, ideal))


if __name__ == "__main__":
    main()
