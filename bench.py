#!/usr/bin/env python
"""bench.py — BASELINE.json metric: face-frames/sec of the Wav2Lip generator hot path, fp32, on N MI355X.

A "step" is one pass of the hot path over one batch (BASELINE configs[1]: 128 synthetic 96x96 BGR crops + 128
mel windows, random-init weights): w2l_datagen_pack -> w2l_mel_gather -> 53 fused conv launches (generator) ->
w2l_frames_to_u8, inputs already resident in HBM; with N > 1 every rank processes its own 128-frame shard and the
uint8 frames are all-gathered over RCCL (the path's one exchange step) — weak scaling.
Successive batches alternate between `--pipeline` (default 2) independent (buffer set, HIP stream) pairs per GPU, so that the
low-occupancy layers of one batch (deep encoder / early decoder levels) overlap the chip-filling layers of the other — what a
serving loop does; every batch is still a full 128-frame pass and K steps are K batches.  `--pipeline 1` is strictly serial.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 128] [--no-cpu-baseline] [--profile-layers]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (bound = fp32 MFMA, 157.3 TFLOP/s
dense; achieved = algorithmic FLOP of the conv launches / their HIP-event time inside the timed region) and, at
N == 1, `cpu_baseline` (the oracle = the reference's CPU path restated, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

GFLOP_PER_FRAME = 7.934          # 3 966 984 192 nominal MACs x 2
PEAK_FP32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="frames per GPU per step (BASELINE config: 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--tune-cache", default=None, help="JSON file of tuned launch configurations: loaded if it "
                    "exists (no autotune launches, for clean rocprof runs), else written after autotuning")
    ap.add_argument("--profile-layers", action="store_true", help="print per-launch HIP-event times to stderr")
    ap.add_argument("--no-train-configs", action="store_true", help="skip the BASELINE configs 3/4 training-step timings that are "
                    "appended (N = 1 only) as `other_configs` from a tools/train_bench.py subprocess")
    ap.add_argument("--pipeline", type=int, default=2, help="batches in flight per GPU: successive 128-frame batches alternate "
                    "between this many (buffer set, stream) pairs, so the low-occupancy layers of one batch overlap the heavy "
                    "layers of the other, as in a serving loop; 1 = strictly one batch at a time")
    return ap.parse_args()


def cpu_baseline(sd, seconds):
    """The oracle's generator forward (the reference's CPU path restated, oracle/models_ref.py) on the host cores.
    torch CPU scales badly past a few dozen threads on small convs, so a few thread counts are probed first (one
    batch each) and the sample is timed at the best one; `cores` is the thread count actually used."""
    from oracle import datagen_ref, models_ref     # the ONLY oracle use in this file: the timed CPU baseline
    from wav2lip_amd import synthetic as synth
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    bs = 16     # the CPU's best-throughput batch in the survey
    img, mel = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(synth.face_crops_u8(bs, seed=11),
                                                                    synth.mel_windows(bs, seed=11)))
    img, mel = torch.from_numpy(img), torch.from_numpy(mel)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    best_t, best_dt = None, None
    for th in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        models_ref.wav2lip_forward(sd_cpu, mel[:2], img[:2])            # warm-up
        t0 = time.perf_counter()
        models_ref.wav2lip_forward(sd_cpu, mel, img)
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best_t, best_dt = th, dt
        if dt > 6.0:
            break
    torch.set_num_threads(best_t)
    n, t0 = 0, time.perf_counter()
    while True:
        models_ref.wav2lip_forward(sd_cpu, mel, img)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 64:
            break
    return {"value": round(n * bs / dt, 2), "unit": "face-frames/sec", "cores": best_t, "kind": "port",
            "sample": "%d batches of %d frames, oracle.models_ref.wav2lip_forward (torch CPU fp32, %d of %d host "
                      "threads, best of a probe), %.1f s" % (n, bs, best_t, avail, dt)}


def hbm_traffic(batch):
    """HBM bytes per step (all launches of one 128-frame pass) from the committed rocprofv3 PMC passes (FETCH_SIZE x2 +
    WRITE_SIZE, profiles/r01/traffic.json, produced by tools/gpu_round.sh + tools/collect_profiles.sh): PMC counters
    cannot be read from inside the timed process, so this is the last measured value, or null when the batch differs"""
    path = os.path.join(ROOT, "profiles", "r01", "traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
        return int(t["hbm_bytes_per_step"]) if int(t.get("frames_per_step", 0)) == batch else None
    except (OSError, ValueError, KeyError):
        return None


def train_configs():
    """BASELINE configs[2] / configs[3] (SyncNet step at batch 512, wav2lip_train step at batch 64) timed by
    tools/train_bench.py in a subprocess, fp32 contractions and the bf16 ones the configs name: reported next to the headline
    metric, never part of `value`.  Any failure is reported as a string instead of breaking the bench line."""
    import subprocess
    out = []
    for prec in ("f32", "bf16"):
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "--cfg", "3", "4", "--steps", "3",
                                "--warmup", "2", "--precision", prec], capture_output=True, text=True, timeout=300)
            lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
            if not lines:
                raise RuntimeError((p.stderr or "no output")[-300:])
            out += lines
        except Exception as e:      # noqa: BLE001 - the headline line must survive
            out.append({"precision": prec, "error": str(e)[:300]})
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (see docstring)" % args.gpus)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU path to measure)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from wav2lip_amd import synthetic as synth    # synthetic weights/inputs (no datasets/checkpoints offline)
    from wav2lip_amd import audio, models
    from wav2lip_amd.inference import Wav2LipRunner, mel_chunk_starts
    from wav2lip_amd.sharding import PipelinedFrameGatherer

    B = args.batch
    G = models.Wav2Lip()
    sd = synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0)
    G.load_state_dict(sd)
    G = G.to(dev).eval()
    runner = Wav2LipRunner(G, batch_size=B)

    # synthetic inputs resident in HBM: B uint8 crops + a mel spectrogram of random 16 kHz audio with B windows
    faces = torch.from_numpy(synth.face_crops_u8(B, seed=100 + rank)).to(dev)
    fps = 25.0
    nsamp = int(16000 * (B + 8) / fps)
    mel = audio.melspectrogram_device(synth.noise_wav(nsamp, seed=200 + rank), dev)
    starts = torch.tensor(mel_chunk_starts(mel.shape[1], fps)[:B], dtype=torch.int32, device=dev)
    assert starts.numel() == B
    gather = (PipelinedFrameGatherer(dist, world, (B, 96, 96, 3), torch.uint8, dev, depth=max(2, args.pipeline))
              if world > 1 else None)

    g = G.graph(B, 96, 96, dev)
    if args.tune_cache and os.path.exists(args.tune_cache):
        g.plan.load_configs(args.tune_cache)
    elif args.tune_cache and rank == 0:
        g.plan.autotune()
        g.plan.save_configs(args.tune_cache)
    lib = runner.lib
    from wav2lip_amd import engine
    from wav2lip_amd._lib import check, current_stream, ptr
    from wav2lip_amd.models.wav2lip import _GeneratorGraph
    if not g.plan.tuned and engine.AUTOTUNE:
        g.plan.autotune()
    # `depth` independent (buffer set, stream) pairs: batch i runs on pair i % depth.  Every pair holds a full plan over its own
    # buffers with the configurations tuned once on the first.
    depth = max(1, args.pipeline)
    graphs = [g] + [_GeneratorGraph(G, B, 96, 96, dev) for _ in range(depth - 1)]
    for gg in graphs[1:]:
        for i, (_, t_, k_) in enumerate(g.plan.configs()):
            gg.plan.set_config(i, t_, k_)
        gg.plan.tuned = True
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] if depth == 1 else [torch.cuda.Stream(device=dev) for _ in range(depth)]
    outs_u8 = [torch.empty((B, 96, 96, 3), dtype=torch.uint8, device=dev) for _ in range(depth)]
    counter = [0]

    def step():
        k = counter[0] % depth
        counter[0] += 1
        gg = graphs[k]
        with torch.cuda.stream(streams[k]):
            s = current_stream()
            check(lib.w2l_datagen_pack(s, B, 96, ptr(faces), ptr(gg.x_in), 8, 8), "datagen_pack")
            check(lib.w2l_mel_gather(s, ptr(mel), mel.shape[1], ptr(starts), B, ptr(gg.mel_in), 4, 4), "mel_gather")
            gg.run()                 # face / audio encoders on two streams, joined before the decoder; same launches as gg.plan
            dst = gather.slot() if gather is not None else outs_u8[k]
            check(lib.w2l_frames_to_u8(s, B, 96, 96, gg.out.ptr, gg.out.cs, ptr(dst)), "frames_to_u8")
            if gather is not None:
                gather.submit()      # asynchronous: this batch's frames cross xGMI while the next batch is computed

    def fence():
        if gather is not None:
            for st in streams:
                with torch.cuda.stream(st):
                    gather.drain()   # every all-gather of the timed region has completed before the clock stops
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev_b, ev_e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_b.record(main_stream)         # HIP events bracket the timed region on the launch streams: every stream starts behind
    for st in streams:               # ev_b and ev_e is recorded after all of them have been joined
        st.wait_stream(main_stream)
    for i in range(args.steps):
        step()
    for st in streams:
        main_stream.wait_stream(st)
    ev_e.record(main_stream)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    conv_ms = ev_b.elapsed_time(ev_e) / args.steps     # GPU time per batch over the timed region (all launches of the step)
    macs = g.plan.macs()
    flop_step = 2.0 * macs
    achieved = flop_step / (conv_ms * 1e-3) / 1e12
    frames = world * B * args.steps
    result = {
        "metric": "face-frames/sec (96x96, mel T=16)",
        "value": round(frames / dt, 1),
        "unit": "face-frames/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "Wav2Lip generator fp32 inference, batch=%d synthetic 96x96x6 crops + random mel per GPU "
                               "(BASELINE configs[1]); datagen pack + mel gather + generator + uint8 frames%s"
                               % (B, " + RCCL all-gather of uint8 frames" if world > 1 else ""),
                   "frames_per_gpu_per_step": B, "parallelism": "dp%d" % world, "batches_in_flight_per_gpu": depth,
                   "weights": "random-init (wav2lip_amd.synthetic seed 0)"},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": hbm_traffic(B),
                     "kernel": "conv_igemm_f32_kernel + conv_wino_f32_kernel (all %d fused conv launches of one generator pass)"
                               % len(g.plan.records),
                     "algorithmic_gflop_per_step": round(flop_step / 1e9, 2),
                     "gflop_per_frame": round(flop_step / 1e9 / B, 4),
                     "gpu_ms_per_step": round(conv_ms, 3)},
    }
    if args.profile_layers and rank == 0:
        prof = g.plan.profile(reps=3)
        tot = sum(p[1] for p in prof)
        for name, ms, m in prof:
            sys.stderr.write("%-34s %8.3f ms %6.1f%%  %7.2f TFLOP/s\n" % (name, ms, 100 * ms / tot, 2 * m / ms / 1e9))
        sys.stderr.write("sum %.3f ms\n" % tot)
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(sd, args.cpu_seconds)
    if world == 1 and not args.no_train_configs and not args.no_cpu_baseline:
        result["other_configs"] = train_configs()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
