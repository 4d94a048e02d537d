#!/usr/bin/env python
"""Generator inference plan at one batch size (GPU box): every launch timed as configured (per-plan list / tune table) and under each
split-operand implicit-GEMM configuration (conv_igemm_bf16_kernel<.., 3>: fp32 operands as three bf16 pieces on the bf16 matrix
cores), with the L-inf distance of the whole forward against the configured plan.  Prints one line per layer with the fastest
split candidate; --emit writes the per-plan list with the winners substituted where they beat the configured launch by --margin.

    python tools/split_sweep.py [--batch 128] [--emit gpurun_out/x/plan_128.json] [--emit-table gpurun_out/x/table_128.json]

--emit-table writes the winners as tune-table entries (the keys the library files these launches under, exported after one run of
the plan with stopwatch tuning off): the file can be loaded on top of the committed table (W2L_TUNE_TABLE / _lib.load_tune_table) or
merged into it by hand, as the batch-128 entries of round 4 were.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--margin", type=float, default=0.97, help="a split candidate replaces the configured launch below this time ratio")
    ap.add_argument("--emit", default="")
    ap.add_argument("--emit-table", default="", help="write the winners as tune-table entries (JSON, the committed table's format)")
    args = ap.parse_args()
    from wav2lip_amd import _lib, models
    from wav2lip_amd import synthetic as synth
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    G = models.Wav2Lip()
    G.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0))
    G = G.to(dev).eval()
    B = args.batch
    r = np.random.default_rng(1)
    mel = torch.from_numpy(r.uniform(-4, 4, (B, 1, 80, 16)).astype(np.float32)).to(dev)
    face = torch.from_numpy(r.uniform(0, 1, (B, 6, 96, 96)).astype(np.float32)).to(dev)
    g = G.graph(B, 96, 96, dev)
    plan = g.plan
    g.load_nchw(mel, face)
    g.run()
    torch.cuda.synchronize()
    ref = g.output_nchw().clone()
    base_cfg = [tuple(c) for c in plan.configs()]
    base_res = plan.resolved()
    n = len(base_res)

    def times():
        best = None
        for _ in range(3):
            t = [ms for _, ms, _ in plan.profile(reps=args.reps)]
            best = t if best is None else [min(a, b) for a, b in zip(best, t)]
        return best

    t_base = times()
    ntiles = lib.w2l_conv_num_tiles()
    nig = lib.w2l_conv_num_igemm_tiles()
    split_ids = [i for i in range(ntiles) if lib.w2l_conv_config_family(i) in (5, 6, 7, 8, 9)]     # split implicit GEMM tiles + split F(2x2) Winograd + split fused-phase transposed
    cand = {}     # layer index -> [(ms, id, ks)]
    for sid in split_ids:
        for ks in ((1, 2, 4, 8) if lib.w2l_conv_config_family(sid) == 5 else (1,)):
            for i in range(n):
                plan.set_config(i, sid, ks)
            res = plan.resolved()
            live = [i for i in range(n) if res[i][3][0] == sid and res[i][3][1] == ks]
            if not live:
                continue
            t = times()
            for i in live:
                cand.setdefault(i, []).append((t[i], sid, ks))
    # all split (fastest split candidate per layer, Winograd layers included): distance of the whole forward
    out_cfg = []
    print("%-28s %-6s %-8s %9s   %-10s %9s  ratio" % ("layer", "family", "config", "ms", "best split", "ms"))
    tot_b = tot_n = 0.0
    for i in range(n):
        name, fl, fam, cfg = base_res[i]
        c = sorted(cand.get(i, []))
        pick = (cfg[0], cfg[1])
        tb = t_base[i]
        if c:
            ms, sid, ks = c[0]
            rel = ms / tb
            if rel < args.margin:
                pick = (sid, ks)
            print("%-28s %-6s (%2d,%2d)  %9.4f   (%2d,%2d)    %9.4f  %5.2f %s" % (name, fam, cfg[0], cfg[1], tb, sid, ks, ms, rel,
                                                                                  "<-" if pick[0] == sid else ""))
            tot_n += min(ms, tb) if pick[0] == sid else tb
        else:
            print("%-28s %-6s (%2d,%2d)  %9.4f   -" % (name, fam, cfg[0], cfg[1], tb))
            tot_n += tb
        tot_b += tb
        out_cfg.append([name, int(pick[0]), int(pick[1])])
    print("serial sum: configured %.3f ms, with split winners %.3f ms" % (tot_b, tot_n))
    for i, (_, c, k) in enumerate(out_cfg):
        plan.set_config(i, c, k)
    g.parts = None                          # the two-stream parts are cut from the plan's configurations at their first run
    g.run()
    torch.cuda.synchronize()
    d = (g.output_nchw() - ref).abs().max().item()
    print("whole forward, winners against configured: L-inf %.3e" % d)
    for i in range(n):                      # every layer that can on its fastest split candidate: the worst case for the distance
        c = sorted(cand.get(i, []))
        if c:
            plan.set_config(i, c[0][1], c[0][2])
    g.parts = None
    g.run()
    torch.cuda.synchronize()
    d2 = (g.output_nchw() - ref).abs().max().item()
    nsplit = sum(1 for _, _, fam, _ in plan.resolved() if fam in ("split", "wino2s", "tp2s", "stem7s", "k3s"))
    print("whole forward, %d of %d launches on split kernels against configured: L-inf %.3e" % (nsplit, n, d2))
    if args.emit_table:
        # key of launch i = the key the library looks up for it: geometry and precision from the layer, (N, H, W) from the plan record
        import ctypes as C
        nk = lib.w2l_tune_key_ints()
        entries = []
        for i, (name, c, k) in enumerate(out_cfg):
            if lib.w2l_conv_config_family(c) not in (5, 6, 7, 8, 9):
                continue
            _, layer, N_, H_, W_ = plan.records[i]
            key = layer.tune_key(N_, H_, W_, has_res=plan.has_res[i])
            assert lib.w2l_tune_entry_applicable((C.c_int * nk)(*key), c) == 1, (name, key, c)
            entries.append(list(key) + [c, k])
        doc = {"key_ints": nk, "num_configs": lib.w2l_conv_num_tiles(),
               "key": "transposed cin cout kh kw sh sw ph pw oph opw precision has_residual head_c N H W -> config ksplit",
               "note": "tools/split_sweep.py --batch %d --margin %.2f: launches the split-operand implicit GEMM wins" % (B, args.margin),
               "entries": sorted(entries)}
        os.makedirs(os.path.dirname(os.path.abspath(args.emit_table)), exist_ok=True)
        with open(args.emit_table, "w") as fh:
            fh.write(json.dumps(doc, separators=(",", ":")).replace("],[", "],\n[") + "\n")
        print("wrote %s (%d entries)" % (args.emit_table, len(entries)))
    if args.emit:
        os.makedirs(os.path.dirname(os.path.abspath(args.emit)), exist_ok=True)
        with open(args.emit, "w") as fh:
            json.dump({str(B): out_cfg}, fh)
        print("wrote", args.emit)


if __name__ == "__main__":
    main()
