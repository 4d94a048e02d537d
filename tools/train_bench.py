#!/usr/bin/env python
"""Training-step timings on one MI355X for BASELINE configs 3-5 (fp32 in this round):

  cfg3  SyncNet_color forward + cosine loss + backward + Adam, batch 512            (color_syncnet_train.py:149-165)
  cfg4  wav2lip_train step: generator on 64x5 frames + frozen SyncNet + L1, Adam    (wav2lip_train.py:210-230)
  cfg5  hq_wav2lip_train step: cfg4 + discriminator (perceptual, real, fake), 2 Adams  (hq_wav2lip_train.py:212-257)

    python tools/train_bench.py [--cfg 3 4 5] [--steps 5] [--warmup 2] [--batch3 512] [--batch 64]

Prints one JSON line per config: ms/step, samples/s, `nominal_tflops` (SURVEY.md 8d per-sample work: 7.26 GFLOP per SyncNet
pair, 123.9 GFLOP per generator sample, 224 GFLOP per hq sample) and - what `frac` is made of - `executed_tflops`: the FLOPs the
matrix cores execute in a step (w2l_flops_begin / _end: padded tiles and K, Winograd products; per kernel family beside it) over
the step time, divided by the dense MFMA peak of the precision (157.3 fp32, 2500 bf16): frac <= 1 by construction.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK = 157.3


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    e0.record()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    timed.host_ms = (time.perf_counter() - t0) * 1e3 / steps      # host time to ENQUEUE a step (== the step time when host-bound)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


FAMILIES = ["fp32 conv fwd/dgrad", "fp32 wgrad (direct)", "fp32 wgrad (Winograd)", "bf16c conv", "bf16c wgrad", "bf16 conv fwd/dgrad",
            "bf16 wgrad", "head reductions (no MFMA)"]


def executed_flops(step):
    """FLOPs the matrix cores EXECUTE in one step (w2l_flops_begin / _end: padded tiles and K, Winograd products - not the nominal
    direct-convolution count): (total, {kernel family: GFLOP})"""
    import ctypes as C
    from wav2lip_amd import _lib
    lib = _lib.load()
    torch.cuda.synchronize()
    lib.w2l_flops_begin()
    step()
    torch.cuda.synchronize()
    by = (C.c_longlong * 8)()
    tot = int(lib.w2l_flops_end(by))
    return tot, {FAMILIES[i]: round(by[i] / 1e9, 1) for i in range(8) if by[i]}


def node_profile(nets, step):
    """one step with every train graph of `nets` recording an event after each launch group; prints per-phase totals and
    the slowest (block, phase) entries with their nominal TFLOP/s"""
    graphs = [(tag, g) for tag, m in nets.items() for lst in m._train_graphs.graphs.values() for g in lst]
    # a profiled step keeps the weight gradients on the main stream (one timeline), which moves the data gradients to the shared
    # scratch buffers: one untimed step in that mode first, so that the one-item plans of those buffers are tuned outside the
    # measured step
    for _, g in graphs:
        g.profile_begin()
    step()
    for _, g in graphs:
        g.profile_end()
    for _, g in graphs:
        g.profile_begin()
    step()
    rows = []
    for tag, g in graphs:
        rows += [(tag + "." + n, ph, ms, macs) for n, ph, ms, macs in g.profile_end()]
    tot = {}
    for n, ph, ms, macs in rows:
        t = tot.setdefault(ph, [0.0, 0])
        t[0] += ms
        t[1] += macs
    print("per-phase totals (ms, nominal TFLOP/s):", file=sys.stderr)
    for ph, (ms, macs) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print("  %-14s %8.3f ms  %7.1f TF/s" % (ph, ms, 2e-9 * macs / ms if macs else 0.0), file=sys.stderr)
    print("slowest entries:", file=sys.stderr)
    for n, ph, ms, macs in sorted(rows, key=lambda r: -r[2])[:45]:
        print("  %-34s %-12s %8.3f ms  %7.1f TF/s" % (n, ph, ms, 2e-9 * macs / ms if macs else 0.0), file=sys.stderr)
    print("sum %.3f ms" % sum(r[2] for r in rows), file=sys.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, nargs="+", default=[3, 4, 5])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch3", type=int, default=512)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--profile-nodes", action="store_true", help="cfg4: per-block, per-phase HIP-event times of one step")
    ap.add_argument("--precision", choices=("f32", "bf16", "bf16c"), default=None,
                    help="training precision (default: W2L_TRAIN_PRECISION or f32); bf16 = the bf16-storage path (bf16 activations "
                         "and gradients in HBM, bf16 matrix cores, fp32 accumulation / statistics / master weights / Adam); bf16c = "
                         "round 2's contraction-only variant over fp32 tensors (A/B)")
    args = ap.parse_args()
    from wav2lip_amd import engine, models, optim, train
    if args.precision:
        engine.set_train_precision(args.precision)
    prec = engine.TRAIN_PRECISION[0]
    peak = {"f32": PEAK, "bf16": 2500.0, "bf16c": 2500.0}[prec]
    # N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...
    # one process per GPU, per-rank batch, gradients averaged by a GradReducer overlapped with the backward pass (RCCL)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    reducer = None
    if world > 1:
        import torch.distributed as dist
        from wav2lip_amd.sharding import GradReducer
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        reducer = GradReducer(dist)
    torch.manual_seed(0)
    r = np.random.default_rng(rank)
    emit = (lambda d: print(json.dumps(dict(d, n_gpus=world, scaling="weak")), flush=True)) if rank == 0 else (lambda d: None)

    def rand(shape, lo=0., hi=1.):
        return torch.from_numpy(r.uniform(lo, hi, shape).astype(np.float32)).to(dev)

    S = models.SyncNet_color().to(dev)
    if reducer is not None:
        reducer.attach(S)
    if 3 in args.cfg:
        B = args.batch3
        opt = optim.Adam([p for p in S.parameters() if p.requires_grad], lr=1e-4)
        x, mel = rand((B, 15, 48, 96)), rand((B, 1, 80, 16), -4, 4)
        y = (rand((B, 1)) > 0.5).float()
        ms = timed(lambda: train.syncnet_train_step(S, opt, x, mel, y), args.steps, args.warmup)
        tf = 7.26 * B / ms
        ex, fam = executed_flops(lambda: train.syncnet_train_step(S, opt, x, mel, y))
        emit({"cfg": 3, "what": "SyncNet fwd+loss+bwd+Adam, " + prec, "batch_per_gpu": B, "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(timed.host_ms, 3),
              "pairs_per_s": round(world * B / ms * 1e3, 1), "nominal_tflops": round(tf, 2), "executed_tflops": round(ex / ms / 1e9, 2),
              "frac": round(ex / ms / 1e9 / peak, 4), "peak_tflops": peak, "executed_gflop_per_step": round(ex / 1e9, 1),
              "executed_gflop_by_kernel": fam, "mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)})
        del opt
    if 4 in args.cfg or 5 in args.cfg:
        B, T = args.batch, 5
        for p in S.parameters():
            p.requires_grad = False
        GradReducer_detach = reducer.detach if reducer is not None else (lambda *a: None)
        GradReducer_detach(S)          # frozen from here on: nothing to average
        G = models.Wav2Lip().to(dev)
        if reducer is not None:
            reducer.attach(G)
        optG = optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
        gt = rand((B, 3, T, 96, 96))
        xin = torch.cat([gt.clone(), rand((B, 3, T, 96, 96))], dim=1)
        xin[:, :3, :, 48:] = 0.
        indiv, melw = rand((B, T, 1, 80, 16), -4, 4), rand((B, 1, 80, 16), -4, 4)
        if 4 in args.cfg:
            ms = timed(lambda: train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03), args.steps,
                       args.warmup)
            tf = 123.9 * B / ms
            ex, fam = executed_flops(lambda: train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03))
            emit({"cfg": 4, "what": "wav2lip_train step (generator 5 frames/sample + frozen SyncNet + L1), " + prec,
                  "batch_per_gpu": B, "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(timed.host_ms, 3), "samples_per_s": round(world * B / ms * 1e3, 2),
                  "nominal_tflops": round(tf, 2), "executed_tflops": round(ex / ms / 1e9, 2), "frac": round(ex / ms / 1e9 / peak, 4),
                  "peak_tflops": peak, "executed_gflop_per_step": round(ex / 1e9, 1), "executed_gflop_by_kernel": fam,
                  "mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)})
        if 4 in args.cfg and args.profile_nodes and rank == 0 and world == 1:
            node_profile({"G": G, "S": S}, lambda: train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03))
        if 5 in args.cfg:
            D = models.Wav2Lip_disc_qual().to(dev)
            if reducer is not None:
                reducer.attach(D)
            optD = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
            ms = timed(lambda: train.hq_train_step(G, D, S, optG, optD, xin, indiv, melw, gt, syncnet_wt=0.03, disc_wt=0.07),
                       args.steps, args.warmup)
            tf = 224.0 * B / ms
            hq = lambda: train.hq_train_step(G, D, S, optG, optD, xin, indiv, melw, gt, syncnet_wt=0.03, disc_wt=0.07)   # noqa: E731
            ex, fam = executed_flops(hq)
            emit({"cfg": 5, "what": "hq_wav2lip_train step (cfg4 + disc perceptual/real/fake), " + prec, "batch_per_gpu": B,
                  "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(timed.host_ms, 3), "samples_per_s": round(world * B / ms * 1e3, 2), "nominal_tflops": round(tf, 2),
                  "executed_tflops": round(ex / ms / 1e9, 2), "frac": round(ex / ms / 1e9 / peak, 4), "peak_tflops": peak,
                  "executed_gflop_per_step": round(ex / 1e9, 1), "executed_gflop_by_kernel": fam,
                  "mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)})


if __name__ == "__main__":
    main()
