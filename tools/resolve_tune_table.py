#!/usr/bin/env python
"""Rewrite every entry of wav2lip_amd/tune_table.json as the configuration a launch of that shape RESOLVES to (GPU box).

Round 2's autotune timed every configuration id for every layer, including ids the layer cannot run; those silently ran the
heuristic, and where one of them "won" the table recorded an id that never executes.  Behaviour was deterministic (the
fall-through is a function of the shape) but the table misdescribed it.  This tool builds a layer handle per entry, resolves
the launch exactly as w2l_conv_forward would (w2l_plan_executed_flops: a dry run, nothing is launched) and stores the resolved
(configuration id, split-K) - so the rewritten table runs the SAME kernels as before, and every entry names what runs.

    python tools/resolve_tune_table.py [--table wav2lip_amd/tune_table.json] [--check]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", default=os.path.join(ROOT, "wav2lip_amd", "tune_table.json"))
    ap.add_argument("--check", action="store_true", help="do not write; exit 1 if any entry does not resolve to itself")
    args = ap.parse_args()
    from wav2lip_amd import _lib
    from wav2lip_amd._lib import ConvGeom, check, ptr
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    doc = json.load(open(args.table))
    nk = doc["key_ints"]
    assert nk == lib.w2l_tune_key_ints()
    lib.w2l_tune_clear()
    assert _lib.load_tune_table(lib, args.table) == len(doc["entries"])
    dummy = torch.zeros(4096, device=dev)
    stream = _lib.current_stream()
    changed, out = 0, []
    handles = {}
    for e in doc["entries"]:
        tr, cin, cout, kh, kw, sh, sw, ph, pw, oph, opw, prec, has_res, head_c, N, H, W = e[:nk]
        gkey = tuple(e[:14])
        h = handles.get(gkey)
        if h is None:
            g = ConvGeom(tr, cin, cout, kh, kw, sh, sw, ph, pw, oph, opw, _lib.ACT_RELU)
            wshape = (cin, cout, kh, kw) if tr else (cout, cin, kh, kw)
            w = torch.randn(wshape, device=dev) * 0.05
            sc, sf = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
            h = C.c_void_p()
            check(lib.w2l_conv_create(C.byref(g), ptr(w), ptr(sc), ptr(sf), stream, C.byref(h)), "conv_create")
            if prec != _lib.PREC_F32:
                check(lib.w2l_conv_set_precision(h, prec), "conv_set_precision")
            if head_c:
                hw, hb = torch.randn(head_c, cout, device=dev), torch.zeros(head_c, device=dev)
                check(lib.w2l_conv_attach_head(h, ptr(hw), ptr(hb), head_c, _lib.ACT_SIGMOID, stream), "conv_attach_head")
            torch.cuda.synchronize()
            handles[gkey] = h
        p = C.c_void_p()
        check(lib.w2l_plan_create(C.byref(p)), "plan_create")
        cs_in = (cin + 3) // 4 * 4
        cs_out = max(4, ((head_c or cout) + 3) // 4 * 4)
        check(lib.w2l_plan_add_conv(p, h, N, H, W, ptr(dummy), cs_in, ptr(dummy), cs_out, ptr(dummy) if has_res else None,
                                    cs_out if has_res else 0), "plan_add_conv")
        fl = (C.c_longlong * 1)()
        cfg = (C.c_int * 2)()
        check(lib.w2l_plan_executed_flops(p, fl, cfg), "plan_executed_flops")
        lib.w2l_plan_destroy(p)
        r = [int(cfg[0]), int(cfg[1])]
        if r != e[nk:nk + 2]:
            changed += 1
            print("entry %s: recorded (%d, %d) resolves to (%d, %d)" % (e[:nk], e[nk], e[nk + 1], r[0], r[1]))
        if not lib.w2l_tune_entry_applicable((C.c_int * nk)(*e[:nk]), r[0]):
            raise SystemExit("w2l_tune_entry_applicable disagrees with the launcher on %s -> %s" % (e[:nk], r))
        out.append(e[:nk] + r)
    for h in handles.values():
        lib.w2l_conv_destroy(h)
    print("%d of %d entries rewritten" % (changed, len(out)))
    if args.check:
        sys.exit(1 if changed else 0)
    doc["entries"] = sorted(out)
    doc["note"] = doc.get("note", "") + "; entries rewritten as the configuration they resolve to (tools/resolve_tune_table.py)"
    with open(args.table, "w") as fh:
        fh.write(json.dumps(doc, indent=None, separators=(",", ":")).replace("],[", "],\n[") + "\n")


if __name__ == "__main__":
    main()
