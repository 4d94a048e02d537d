#!/usr/bin/env python
"""Regenerate wav2lip_amd/tune_table.json on a GPU box: the shape-keyed (configuration, split-K) table every default run
replays (include/w2l_hip.h, "tune table").

Stopwatch tuning is opt-in (W2L_AUTOTUNE=1) because a timed choice changes a layer's summation order from box to box; this
tool is the one place that opts in.  It runs the workloads of the BASELINE configurations with autotuning on - generator
inference at the serving batch sizes, the three training steps in fp32 and bf16, the S3FD trunk - lets w2l_plan_autotune
record every winner in the library's table and writes the table out.  Commit the result: from then on every process loads
it (wav2lip_amd/_lib.py) and the same shapes run the same configurations everywhere.

    W2L_AUTOTUNE=1 python tools/make_tune_table.py [--out wav2lip_amd/tune_table.json] [--quick]
    W2L_AUTOTUNE=1 python tools/make_tune_table.py --add-batches 2,3,4,5,6,7

After regenerating the table, re-dump the per-plan launch lists that ride on it (wav2lip_amd/plan_configs.json):
    python tools/batch_sweep.py --batches 1,2,3,4,5,6,7,8,16,32,64,128,256 --add-table <table with the 2..7 entries> \
        --dump-configs wav2lip_amd/plan_configs.json

--add-batches keeps every committed entry bit for bit (the training goldens are anchored to the summation orders those
entries select) and only ADDS entries for generator inference at the listed batch sizes: the heuristic never splits K, so
between the tuned batches 1 and 8 it leaves the bottleneck layers on a handful of workgroups (EXPERIMENTS.md, "batch 2-7
cliff").
"""
import argparse
import os
import sys

os.environ["W2L_AUTOTUNE"] = "1"
ADDITIVE = "--add-batches" in " ".join(sys.argv)
if not ADDITIVE:
    os.environ.setdefault("W2L_TUNE_TABLE", "0")  # start from an empty table: no stale entries survive a regeneration
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "wav2lip_amd", "tune_table.json"))
    ap.add_argument("--quick", action="store_true", help="inference batch 128 and cfg3/cfg4 fp32 only")
    ap.add_argument("--exact", action="store_true", help="the exact table (wav2lip_amd/tune_table_exact.json): generator inference "
                    "plans only, tuned with the F(4x4,3x3) Winograd family switched off (w2l_conv_exclude_families)")
    ap.add_argument("--add-batches", default="", help="comma-separated generator-inference batch sizes to ADD to the committed "
                    "table; every existing entry is kept unchanged")
    ap.add_argument("--rounds", type=int, default=2, help="tuning passes per workload; the LAST pass's winner is kept")
    args = ap.parse_args()
    from wav2lip_amd import _lib, engine, models, optim, train
    from wav2lip_amd import synthetic as synth
    assert engine.AUTOTUNE
    lib = _lib.load()
    if args.exact:
        _lib.check(lib.w2l_conv_exclude_families(1 << _lib.FAMILY_WINO4), "conv_exclude_families")
        if args.out == os.path.join(ROOT, "wav2lip_amd", "tune_table.json"):
            args.out = _lib.EXACT_TABLE_PATH
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    r = np.random.default_rng(0)

    def rand(shape, lo=0., hi=1.):
        return torch.from_numpy(r.uniform(lo, hi, shape).astype(np.float32)).to(dev)

    def note(msg):
        print("[tune] %-52s table entries: %d" % (msg, lib.w2l_tune_count()), flush=True)

    # ---- inference plans (BASELINE configs[0..1] and the batch sizes a serving loop / the tests use)
    G = models.Wav2Lip()
    G.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0))
    G = G.to(dev).eval()
    if args.add_batches:
        before = {tuple(e[:-2]): e for e in _lib.export_tune_table(lib)}
        if not before:
            raise SystemExit("--add-batches extends the committed table, but none was loaded (W2L_TUNE_TABLE=%s)"
                             % os.environ.get("W2L_TUNE_TABLE"))
        nk = lib.w2l_tune_key_ints()
        for B in [int(b) for b in args.add_batches.split(",") if b]:
            g = G.graph(B, 96, 96, dev)
            for _ in range(args.rounds):
                g.plan.autotune(reps=5)
            note("generator inference B=%d (added)" % B)
        for e in before.values():   # a batch size that was already tuned keeps its committed entry
            _lib.check(lib.w2l_tune_set((_lib.C.c_int * nk)(*e[:nk]), e[nk], e[nk + 1]), "tune_set")
        after = _lib.export_tune_table(lib)
        assert all(before.get(tuple(e[:-2]), e) == e for e in after) and len(after) >= len(before)
        torch.cuda.synchronize()
        n = _lib.save_tune_table(lib, args.out, note="tools/make_tune_table.py on %s, rounds=%d; + generator inference at "
                                 "batches %s (--add-batches)" % (torch.cuda.get_device_name(0), args.rounds, args.add_batches))
        print("[tune] wrote %d entries (%d new) to %s" % (n, n - len(before), args.out), flush=True)
        return
    for B in ([128] if args.quick else [128, 256, 64, 32, 16, 8, 1]):
        g = G.graph(B, 96, 96, dev)
        for _ in range(args.rounds):
            g.plan.autotune(reps=3)
        note("generator inference B=%d" % B)
    if args.exact:
        torch.cuda.synchronize()
        n = _lib.save_tune_table(lib, args.out, note="tools/make_tune_table.py --exact on %s, rounds=%d: generator inference, no "
                                 "F(4x4) Winograd" % (torch.cuda.get_device_name(0), args.rounds))
        print("[tune] wrote %d entries to %s" % (n, args.out), flush=True)
        return
    if not args.quick:
        S = models.SyncNet_color().to(dev).eval()
        with torch.no_grad():
            for B in (64, 512):
                S(rand((B, 1, 80, 16), -4, 4), rand((B, 15, 48, 96)))
        note("SyncNet inference B=64,512")
        D = models.Wav2Lip_disc_qual().to(dev).eval()
        with torch.no_grad():
            D(rand((8, 3, 5, 96, 96)))
        note("disc inference 40 frames")

    # ---- training steps (configs 3-5), fp32 and bf16 contractions
    # (the bf16-storage path picks its launch shapes by rule - csrc/conv_bf16.hip pickb - and holds no table entries; "bf16c" is
    # round 2's contraction-only variant over fp32 tensors, kept for A/B)
    for prec in (["f32"] if args.quick else ["f32", "bf16c"]):
        engine.set_train_precision(prec)
        S = models.SyncNet_color().to(dev)
        opt = optim.Adam([p for p in S.parameters() if p.requires_grad], lr=1e-4)
        B = 512
        x, mel, y = rand((B, 15, 48, 96)), rand((B, 1, 80, 16), -4, 4), (rand((B, 1)) > 0.5).float()
        for _ in range(2):
            train.syncnet_train_step(S, opt, x, mel, y)
        note("cfg3 SyncNet step B=512 %s" % prec)
        del opt
        for p in S.parameters():
            p.requires_grad = False
        B, T = 64, 5
        G = models.Wav2Lip().to(dev)
        optG = optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
        gt = rand((B, 3, T, 96, 96))
        xin = torch.cat([gt.clone(), rand((B, 3, T, 96, 96))], dim=1)
        xin[:, :3, :, 48:] = 0.
        indiv, melw = rand((B, T, 1, 80, 16), -4, 4), rand((B, 1, 80, 16), -4, 4)
        for _ in range(2):
            train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03)
        note("cfg4 wav2lip_train step B=64 %s" % prec)
        if not args.quick:
            D = models.Wav2Lip_disc_qual().to(dev)
            optD = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
            for _ in range(2):
                train.hq_train_step(G, D, S, optG, optD, xin, indiv, melw, gt, syncnet_wt=0.03, disc_wt=0.07)
            note("cfg5 hq_wav2lip_train step B=64 %s" % prec)
            del D, optD
        del G, optG, S
        torch.cuda.empty_cache()
    engine.set_train_precision("f32")

    if not args.quick:
        try:
            from wav2lip_amd import face_detection as fd
            fa = fd.FaceAlignment(fd.LandmarksType._2D, device="cuda", state_dict=synth.s3fd_state_dict())
            net = fa.face_detector
            img = torch.from_numpy(r.integers(0, 256, (16, 480, 640, 3), dtype=np.uint8)).to(dev)
            net.dense_boxes(img)
            note("S3FD 16 x 480x640")
        except Exception as e:     # noqa: BLE001 - the detector is a "next" row: its absence must not lose the table
            print("[tune] S3FD skipped: %s" % e, flush=True)

    torch.cuda.synchronize()
    n = _lib.save_tune_table(lib, args.out, note="tools/make_tune_table.py on %s, rounds=%d" %
                             (torch.cuda.get_device_name(0), args.rounds))
    print("[tune] wrote %d entries to %s" % (n, args.out), flush=True)


if __name__ == "__main__":
    main()
