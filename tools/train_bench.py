#!/usr/bin/env python
"""Training-step timings on one MI355X for BASELINE configs 3-5 (fp32 in this round):

  cfg3  SyncNet_color forward + cosine loss + backward + Adam, batch 512            (color_syncnet_train.py:149-165)
  cfg4  wav2lip_train step: generator on 64x5 frames + frozen SyncNet + L1, Adam    (wav2lip_train.py:210-230)
  cfg5  hq_wav2lip_train step: cfg4 + discriminator (perceptual, real, fake), 2 Adams  (hq_wav2lip_train.py:212-257)

    python tools/train_bench.py [--cfg 3 4 5] [--steps 5] [--warmup 2] [--batch3 512] [--batch 64] [--gpus N]
        (--gpus N without WORLD_SIZE in the environment: re-executes itself as N ranks, one per GPU, under torch.distributed.run on
         a free local port - data-parallel samples, gradients averaged by sharding.GradReducer overlapped with the backward pass;
         `--backend gloo --dry-run`: the same launch + bucketed reducer on CPU with rank-coded gradients, no kernels)

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


GOLDEN = os.path.join(ROOT, "tests", "golden", "golden_train_baseline_v1.npz")


def loss_parity(cfg, B, prec, got):
    """In-run parity of the FIRST step of a config (seeded weights of wav2lip_amd/synthetic.py, inputs synthetic.train_batch(cfg, B,
    5)) against the committed golden of the same step on the REAL reference modules (tests/golden/make_golden_train_baseline.py):
    fp32: every loss within 1e-4 of the reference's (the cosine loss through the frozen train-mode SyncNet: 1e-3); bf16: within 3x
    the bf16 error model's spread around the fp64 value (floor 2^-8).  Returns the `parity` object of the JSON line; None when
    the golden does not hold this (cfg, batch)."""
    if not os.path.exists(GOLDEN):
        return None
    g = np.load(GOLDEN)
    tag = "cfg%d" % cfg
    if tag + "_batch" not in g.files or int(g[tag + "_batch"]) != B:
        return None
    rows, ok = {}, True
    for k, v in got.items():
        ref, r64 = float(g["%s_%s" % (tag, k)]), float(g["%s_%s64" % (tag, k)])
        if prec == "f32":
            err, bound = abs(v - ref), (1e-3 if k == "sync" else 1e-4) * abs(ref)
        else:
            spread = max(abs(float(g["%s_%s_noise%d" % (tag, k, s_)]) - r64) for s_ in range(int(g[tag + "_noise_seeds"])))
            err, bound = abs(v - r64), 3 * max(spread, 2.0 ** -8 * abs(r64))
        rows[k] = {"got": round(v, 7), "reference_fp32": round(ref, 7), "fp64": round(r64, 7), "abs_err": float("%.3e" % err),
                   "bound": float("%.3e" % bound)}
        ok = ok and err <= bound
    return {"ok": bool(ok), "against": "tests/golden/golden_train_baseline_v1.npz: the step on the reference's nn.Modules, torch "
            + str(g["torch_version"]) + " CPU (fp32) and the oracle graph in fp64", "step": "first step, seeded weights and inputs",
            "losses": rows}


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
    import re
    grp = {}
    for n, ph, ms, macs in rows:
        k = (re.sub(r"\.\d+(\.\d+)?$", "", n), ph)
        t = grp.setdefault(k, [0.0, 0, 0.0, 0])
        t[0] += ms
        t[1] += 1
        if ms < 0.015:
            t[2] += ms
            t[3] += 1
    print("by sub-network and phase (ms, entries; of which entries under 15 us):", file=sys.stderr)
    for (n, ph), (ms, cnt, sms, scnt) in sorted(grp.items()):
        print("  %-26s %-12s %8.3f ms %4d   | %7.3f ms %4d" % (n, ph, ms, cnt, sms, scnt), file=sys.stderr)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """`--gpus N` outside torch.distributed.run: N ranks of this file, one per GPU, on a free local port; a rank that dies takes
    the job down (the elastic agent stops the others) and this process exits non-zero"""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(cmd, env=env))


def dry_run(args, world, rank):
    """The N > 1 control path of a training step without a device: every rank walks the generator's (cfg 4) or generator +
    discriminator's (cfg 5) parameter blocks last-to-first as a backward pass does, hands rank- and step-coded gradients to the
    SAME bucketed GradReducer, finalizes, and checks every averaged gradient; per-rank step times are gathered and ONE JSON line
    is printed.  Exercises launch, rendezvous (with timeout), bucket composition, asynchronous all-reduce and the average."""
    import datetime
    import time
    import torch.distributed as dist
    from wav2lip_amd import models
    from wav2lip_amd.sharding import GradReducer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    if args.inject_failure == rank:
        sys.exit("train_bench: injected failure on rank %d (launch test)" % rank)
    dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.dist_timeout))
    nets = [models.Wav2Lip()] + ([models.Wav2Lip_disc_qual()] if 5 in args.cfg else [])
    blocks = []                                   # [(network index, block name, {param name: shape})] in backward order
    for ni, net in enumerate(nets):
        per = {}
        for name, p_ in net.named_parameters():
            per.setdefault(name.rsplit(".", 3)[0], {})[name] = tuple(p_.shape)
        blocks += [(ni, b, shapes) for b, shapes in reversed(list(per.items()))]
    reducer = GradReducer(dist, bucket_bytes=args.bucket_mb << 20)
    nparam = sum(int(np.prod(sh)) for _, _, shapes in blocks for sh in shapes.values())
    ok, ms = True, []
    for step in range(args.warmup + args.steps):
        dist.barrier()
        t0 = time.perf_counter()
        for ni, _, shapes in blocks:
            reducer.on_grads({(ni, k): torch.full(sh, float(rank + 1 + step), dtype=torch.float32) for k, sh in shapes.items()})
        nb = len(reducer._inflight) + (1 if reducer._open else 0)
        out = reducer.finalize()
        dt = (time.perf_counter() - t0) * 1e3
        want = (world + 1) / 2.0 + step
        ok = ok and len(out) == sum(len(s_) for _, _, s_ in blocks) and all(
            bool((g == want).all()) and tuple(g.shape) == shapes[k[1]] for ni, _, shapes in blocks for k, g in
            ((kk, out[kk]) for kk in ((ni, n_) for n_ in shapes)))
        if step >= args.warmup:
            ms.append(dt)
    t = torch.tensor([float(np.median(ms))], dtype=torch.float64)
    allms = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allms, t)
    okt = torch.tensor([1 if ok else 0])
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"dry_run": True, "cfg": 5 if 5 in args.cfg else 4, "n_gpus": world, "scaling": "weak",
                          "collective_world_size": dist.get_world_size(), "collective_backend": dist.get_backend(),
                          "reduced_parameters": nparam, "buckets_per_step": nb, "bucket_mb": args.bucket_mb,
                          "gradients_verified": bool(okt.item()),
                          "per_rank_ms_per_step": [round(float(v.item()), 3) for v in allms],
                          "ms_per_step": round(max(float(v.item()) for v in allms), 3)}), flush=True)
    dist.destroy_process_group()
    if not okt.item():
        sys.exit("dry run: averaged gradients are wrong")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, nargs="+", default=[3, 4, 5])
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 without WORLD_SIZE in the environment: re-execute as N ranks under "
                    "torch.distributed.run (one per GPU, RCCL), per-rank batch fixed (weak scaling)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="gloo exists for --dry-run only")
    ap.add_argument("--dry-run", action="store_true", help="no kernels: the launch + bucketed GradReducer path on CPU tensors")
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="seconds before a rendezvous / collective gives up (a rank "
                    "that never arrives must end the job with a non-zero exit, not hang it)")
    ap.add_argument("--bucket-mb", type=int, default=32, help="GradReducer bucket size")
    ap.add_argument("--inject-failure", type=int, default=-1, help="(launch test) this rank exits before the rendezvous")
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
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)       # does not return
    if args.backend == "gloo" and not args.dry_run:
        sys.exit("train_bench.py: --backend gloo exists for --dry-run only; the measured reduce is RCCL (nccl)")
    if args.dry_run:
        return dry_run(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))
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
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                timeout=datetime.timedelta(seconds=args.dist_timeout))
        reducer = GradReducer(dist, bucket_bytes=args.bucket_mb << 20)
    torch.manual_seed(0)
    r = np.random.default_rng(rank)
    def emit(d):
        """one JSON line per config on rank 0; N > 1: the slowest rank's step time is the job's, per-rank times beside it"""
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([d["ms_per_step"]], device=dev, dtype=torch.float64)
            allt = torch.zeros(world, device=dev, dtype=torch.float64)
            dist.all_gather_into_tensor(allt, t)
            d = dict(d, per_rank_ms_per_step=[round(float(v), 3) for v in allt.tolist()], ms_per_step=round(float(allt.max()), 3),
                     collective_world_size=dist.get_world_size(), collective_backend=dist.get_backend())
            for key, n in (("pairs_per_s", "batch_per_gpu"), ("samples_per_s", "batch_per_gpu")):
                if key in d:
                    d[key] = round(world * d[n] / d["ms_per_step"] * 1e3, 2)
        if rank == 0:
            print(json.dumps(dict(d, n_gpus=world, scaling="weak")), flush=True)

    from wav2lip_amd import synthetic as synth

    def seeded(cls, seed):
        m = cls()
        m.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed))
        return m.to(dev)

    def batch(cfg, B):
        # rank 0 of every job runs the golden's inputs (seed 5); other ranks their own
        return {k: torch.from_numpy(v).to(dev) for k, v in synth.train_batch(cfg, B, 5 + rank).items()}

    failed = []

    def with_parity(d, cfg, B, got):
        par = loss_parity(cfg, B, prec, got) if rank == 0 else None
        if par is not None:
            d["parity"] = par
            if not par["ok"]:
                failed.append(cfg)
        return d

    S = seeded(models.SyncNet_color, 2)
    if reducer is not None:
        reducer.attach(S)
    if 3 in args.cfg:
        B = args.batch3
        opt = optim.Adam([p for p in S.parameters() if p.requires_grad], lr=1e-4)
        b3 = batch(3, B)
        x, mel, y = b3["x"], b3["mel"], b3["y"]
        first = {"loss": float(train.syncnet_train_step(S, opt, x, mel, y))}
        ms = timed(lambda: train.syncnet_train_step(S, opt, x, mel, y), args.steps, args.warmup)
        tf = 7.26 * B / ms
        ex, fam = executed_flops(lambda: train.syncnet_train_step(S, opt, x, mel, y))
        emit(with_parity({"cfg": 3, "what": "SyncNet fwd+loss+bwd+Adam, " + prec, "batch_per_gpu": B, "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(timed.host_ms, 3),
              "pairs_per_s": round(world * B / ms * 1e3, 1), "nominal_tflops": round(tf, 2), "executed_tflops": round(ex / ms / 1e9, 2),
              "frac": round(ex / ms / 1e9 / peak, 4), "peak_tflops": peak, "executed_gflop_per_step": round(ex / 1e9, 1),
              "executed_gflop_by_kernel": fam, "mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}, 3, B, first))
        if args.profile_nodes and rank == 0 and world == 1 and 4 not in args.cfg:
            node_profile({"S": S}, lambda: train.syncnet_train_step(S, opt, x, mel, y))
        del opt
    if 4 in args.cfg or 5 in args.cfg:
        B, T = args.batch, 5
        GradReducer_detach = reducer.detach if reducer is not None else (lambda *a: None)
        GradReducer_detach(S)          # frozen from here on: nothing to average
        S = seeded(models.SyncNet_color, 2)      # the expert as loaded from its checkpoint, not as cfg 3's steps left it
        for p in S.parameters():
            p.requires_grad = False
        G = seeded(models.Wav2Lip, 0)
        if reducer is not None:
            reducer.attach(G)
        optG = optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
        b4 = batch(4, B)
        gt, xin, indiv, melw = b4["gt"], b4["x"], b4["indiv_mels"], b4["mel"]
        if 4 in args.cfg:
            first = dict(zip(("loss", "l1", "sync"), (float(v) for v in train.wav2lip_train_step(
                G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03))))
            ms = timed(lambda: train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03), args.steps,
                       args.warmup)
            tf = 123.9 * B / ms
            ex, fam = executed_flops(lambda: train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03))
            emit(with_parity({"cfg": 4, "what": "wav2lip_train step (generator 5 frames/sample + frozen SyncNet + L1), " + prec,
                  "batch_per_gpu": B, "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(timed.host_ms, 3), "samples_per_s": round(world * B / ms * 1e3, 2),
                  "nominal_tflops": round(tf, 2), "executed_tflops": round(ex / ms / 1e9, 2), "frac": round(ex / ms / 1e9 / peak, 4),
                  "peak_tflops": peak, "executed_gflop_per_step": round(ex / 1e9, 1), "executed_gflop_by_kernel": fam,
                  "mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}, 4, B, first))
        if 4 in args.cfg and args.profile_nodes and rank == 0 and world == 1:
            node_profile({"G": G, "S": S}, lambda: train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03))
        if 5 in args.cfg:
            D = seeded(models.Wav2Lip_disc_qual, 4)
            if reducer is not None:
                reducer.attach(D)
            optD = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
            if 4 in args.cfg:      # the golden's hq step starts from the seeded generator, not from cfg 4's trained one
                G.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0))
                optG = optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
            first = {k: float(v) for k, v in train.hq_train_step(G, D, S, optG, optD, xin, indiv, melw, gt, syncnet_wt=0.03,
                                                                 disc_wt=0.07).items()}
            ms = timed(lambda: train.hq_train_step(G, D, S, optG, optD, xin, indiv, melw, gt, syncnet_wt=0.03, disc_wt=0.07),
                       args.steps, args.warmup)
            tf = 224.0 * B / ms
            hq = lambda: train.hq_train_step(G, D, S, optG, optD, xin, indiv, melw, gt, syncnet_wt=0.03, disc_wt=0.07)   # noqa: E731
            ex, fam = executed_flops(hq)
            emit(with_parity({"cfg": 5, "what": "hq_wav2lip_train step (cfg4 + disc perceptual/real/fake), " + prec, "batch_per_gpu": B,
                  "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(timed.host_ms, 3), "samples_per_s": round(world * B / ms * 1e3, 2), "nominal_tflops": round(tf, 2),
                  "executed_tflops": round(ex / ms / 1e9, 2), "frac": round(ex / ms / 1e9 / peak, 4), "peak_tflops": peak,
                  "executed_gflop_per_step": round(ex / 1e9, 1), "executed_gflop_by_kernel": fam,
                  "mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}, 5, B, first))
    if failed:
        sys.exit("train_bench: in-run loss parity FAILED for cfg %s (see the `parity` objects above)" % failed)


if __name__ == "__main__":
    main()
