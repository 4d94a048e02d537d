#!/usr/bin/env python
"""bench.py — BASELINE.json metric: face-frames/sec of the Wav2Lip generator hot path, fp32, on N MI355X.

A "step" is one pass of the hot path over one batch (BASELINE configs[1]: 128 synthetic 96x96 BGR crops + 128
mel windows, random-init weights): w2l_datagen_pack -> w2l_mel_gather -> 53 fused conv launches (generator) ->
w2l_frames_to_u8, inputs already resident in HBM; with N > 1 every rank processes its own 128-frame shard and the
uint8 frames are all-gathered over RCCL (the path's one exchange step) — weak scaling.
The timed loop is the PRODUCT's: `wav2lip_amd.inference.PipelinedRunner(depth=--pipeline).submit`, the loop inference.lipsync and
inference.main run.  Successive batches alternate between `--pipeline` (default 4) independent (buffer set, HIP stream) lanes per GPU, so that the
low-occupancy layers of one batch (deep encoder / early decoder levels) overlap the chip-filling layers of the other — what a
serving loop does; every batch is still a full 128-frame pass and K steps are K batches.  `--pipeline 1` is strictly serial.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 128] [--no-cpu-baseline] [--profile-layers]
        (N > 1 without WORLD_SIZE in the environment: re-executes itself as N ranks under torch.distributed.run, free port)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (bound = fp32 MFMA, 157.3 TFLOP/s dense;
achieved = FLOPs the matrix cores EXECUTE per step - Winograd layers at 16 (F(2x2)) or 9 (F(4x4)) instead of 36 products per 2x2 outputs, padded
tiles included - over the HIP-event time of the timed region, so frac <= 1; the nominal direct-convolution rate and the
dominant kernel's own fraction are reported beside it) and, at N == 1, `cpu_baseline` (the reference's CPU path timed on the
host cores on a bounded sample) and `parity` (the timed path's last batch checked against that CPU forward).
Launch configurations are the committed tune table + heuristic (bit-reproducible); `--autotune` opts into stopwatch tuning.
"""
import argparse
import datetime
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
PEAK_BF16_MFMA_TFLOPS = 2516.6   # the same table: dense bf16 = 16x the fp32 matrix rate (1024 FLOP/clk/SIMD; "~2.5 PF")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="the timed region of exactly --steps steps is repeated this many "
                    "times (each bracketed by barrier + synchronize) and the MEDIAN window is reported, with min / max beside it")
    ap.add_argument("--batch", type=int, default=128, help="frames per GPU per step (BASELINE config: 128)")
    ap.add_argument("--sustained-seconds", type=float, default=2.5, help="after the --windows timed regions, ONE more timed region of "
                    "as many steps as fill this many seconds of GPU time (same fences, max over ranks) with the shader-clock probe "
                    "running across all of it: `windows.sustained_value` - the steady-state rate of a long clip; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=14.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--autotune", action="store_true", help="opt into stopwatch tuning of the launch configurations (default: "
                    "the committed shape-keyed tune table + heuristic, bit-reproducible); recorded in config.launch_configs")
    ap.add_argument("--tune-cache", default=None, help="JSON file of launch configurations: loaded if it exists, else "
                    "written after --autotune (A/B runs and rocprof passes of one session share one set of choices)")
    ap.add_argument("--profile-layers", action="store_true", help="print per-launch HIP-event times to stderr")
    ap.add_argument("--no-train-configs", action="store_true", help="skip the BASELINE configs 3/4 training-step timings that are "
                    "appended (N = 1 only) as `other_configs` from a tools/train_bench.py subprocess")
    ap.add_argument("--pipeline", type=int, default=4, help="batches in flight per GPU: successive 128-frame batches alternate "
                    "between this many (buffer set, stream) pairs, so the low-occupancy layers of one batch overlap the heavy "
                    "layers of the other, as in a serving loop; 1 = strictly one batch at a time")
    ap.add_argument("--launch-threads", type=int, default=1, help="accepted for old command lines and ignored: the timed loop is the "
                    "product's PipelinedRunner, which one host thread drives (several enqueueing threads measured the same, round 3)")
    ap.add_argument("--exact", action="store_true", help="run the launch table tuned without the F(4x4,3x3) Winograd kernel "
                    "(wav2lip_amd/tune_table_exact.json, W2L_EXACT=1): F(2x2) / implicit-GEMM launches only - about half the "
                    "rounding error against the reference, at the frames/s this run then reports")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="torch.distributed backend of the N > 1 exchange "
                    "step: nccl (= RCCL over xGMI, the measured path) or gloo (CPU; only with --dry-run)")
    ap.add_argument("--dry-run", action="store_true", help="no kernels: every rank fills its frame slot with a (rank, step) "
                    "pattern instead of running the generator, then the same partition / pipelined all-gather / fence / max-over-"
                    "ranks timing code runs and the gathered order is verified (the CPU test of the N > 1 launch path)")
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="seconds before the rendezvous / a collective gives up: a rank "
                    "that never arrives ends the job with a non-zero exit instead of hanging rank 0 at the barrier")
    ap.add_argument("--inject-failure", type=int, default=-1, help="(launch test) this rank exits before the rendezvous")
    return ap.parse_args()


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` outside torch.distributed.run (the driver's SCALE command): re-execute this file as N ranks,
    one per GPU, under torch.distributed.run on a free local port; rank 0's JSON line is this process's stdout."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(cmd, env=env))


def dry_run(args, world, rank, dist):
    """--dry-run: the N > 1 control path without a device - contiguous frame shards (shard_range), `--pipeline` rotating
    (send, receive) buffer pairs, asynchronous all-gathers, drain + barrier fence, max-over-ranks wall time, per-rank rates -
    with a (rank, step)-coded byte pattern in place of the generator's frames, verified after the gather on every rank."""
    from wav2lip_amd.sharding import PipelinedFrameGatherer, shard_range
    B = args.batch
    dev = torch.device("cpu")
    n_total = world * B * args.steps                       # the notional clip: every rank owns a contiguous run of frames
    lo, hi = shard_range(n_total, rank, world)
    assert hi - lo == B * args.steps
    gather = PipelinedFrameGatherer(dist, world, (B, 4, 4, 3), torch.uint8, dev, depth=max(2, args.pipeline))
    ok = True

    def check(recv, step):
        return all(bool((recv[r * B:(r + 1) * B] == (17 * r + step) % 251).all()) for r in range(world))

    def fence():
        last = gather.drain()
        dist.barrier()
        return last

    for w in range(args.warmup):
        gather.slot().fill_((17 * rank + w) % 251)
        gather.submit()
    fence()
    wall = []
    for _ in range(max(1, args.windows)):
        t0 = time.perf_counter()
        pending = []
        for i in range(args.steps):
            gather.slot().fill_((17 * rank + i) % 251)
            pending.append((i, gather.submit()))
            if len(pending) > gather.depth - 1:            # the oldest receive buffer is complete before its pair is reused
                j, recv = pending.pop(0)
                gather.work[(gather.i - len(pending) - 1) % gather.depth].wait()
                ok = ok and check(recv, j)
        last = fence()
        ok = ok and check(last, args.steps - 1)
        dt_local = time.perf_counter() - t0
        t = torch.tensor([dt_local], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall.append((float(t.item()), dt_local))
    med = sorted(range(len(wall)), key=lambda i: wall[i][0])[len(wall) // 2]
    dt, dt_local = wall[med]
    rates = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(rates, torch.tensor([B * args.steps / dt_local], dtype=torch.float64))
    okt = torch.tensor([1 if ok else 0])
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({
            "metric": "face-frames/sec (96x96, mel T=16)", "value": round(world * B * args.steps / dt, 1), "unit": "face-frames/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "dry-run pattern",
            "dry_run": True, "gather_verified": bool(okt.item()), "frame_shards": [list(shard_range(n_total, r, world)) for r in range(world)],
            "per_rank_frames_per_s": [round(float(r.item()), 1) for r in rates],
            "config": {"workload": "DRY RUN: partition + pipelined all-gather + fence only, no kernels (not a measurement)",
                       "frames_per_gpu_per_step": B, "parallelism": "dp%d" % world, "batches_in_flight_per_gpu": gather.depth,
                       "collective_world_size": dist.get_world_size(), "collective_backend": dist.get_backend()}}), flush=True)
    dist.destroy_process_group()
    if not okt.item():
        sys.exit("dry run: gathered frames out of order")


def host_cpu():
    """(model name, logical CPUs available to this process) of the host the CPU baseline runs on"""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, avail


def cpu_baseline(sd, seconds, check_faces=None, check_mels=None):
    """The reference's CPU path timed on this host: the REAL `models.Wav2Lip` imported from /root/reference when that
    checkout is present (`kind: "reference"`; build container), otherwise the oracle's restatement of it
    (oracle/models_ref.py, pinned bit-for-bit to the reference by tests/golden/make_golden.py; `kind: "port"`; the GPU box has
    no /root/reference).  SURVEY.md 8(d): fp32, torch.no_grad, eval mode; thread counts are probed up to every logical CPU
    (torch CPU convs scale badly past a few dozen threads, so the best of the probe is used and `cores` says which);
    reported at B=128 (the BASELINE batch) and at the CPU's best batch (B=16); `value` is the better.
    When `check_faces` / `check_mels` (uint8 crops, mel windows of a batch the GPU just processed) are given, the same CPU
    forward also returns its float32 prediction for them: bench.py's in-run parity check against the timed HIP path.
    This function is the ONLY place in this file that touches oracle/."""
    from oracle import datagen_ref, models_ref
    from wav2lip_amd import synthetic as synth
    model_name, avail = host_cpu()
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    kind, fwd = "port", (lambda mel, img: models_ref.wav2lip_forward(sd_cpu, mel, img))
    ref_root = "/root/reference"
    if os.path.isdir(os.path.join(ref_root, "models")):
        try:
            sys.path.insert(0, ref_root)
            import importlib
            ref_models = importlib.import_module("models")
            net = ref_models.Wav2Lip()
            net.load_state_dict(sd_cpu)
            net.eval()

            def fwd(mel, img, net=net):     # noqa: F811 - the reference module replaces the port
                with torch.no_grad():
                    return net(mel, img)
            kind = "reference"
        except Exception:       # noqa: BLE001 - fall back to the pinned port, and say so in `kind`
            kind = "port"
        finally:
            if ref_root in sys.path:
                sys.path.remove(ref_root)

    def inputs(bs, seed):
        img, mel = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(synth.face_crops_u8(bs, seed=seed),
                                                                        synth.mel_windows(bs, seed=seed)))
        return torch.from_numpy(mel), torch.from_numpy(img)

    mel16, img16 = inputs(16, 11)
    probe = {}
    budget = time.perf_counter() + 0.35 * seconds
    for th in sorted({min(avail, c) for c in (4, 8, 16, 32, 64, 128, 256, avail)}):
        torch.set_num_threads(th)
        fwd(mel16[:2], img16[:2])            # warm-up at this thread count
        t0 = time.perf_counter()
        fwd(mel16, img16)
        probe[th] = time.perf_counter() - t0
        # past the knee torch's CPU convs only get slower (measured on a 256-thread EPYC 9575F host: 0.16 s at 16 threads,
        # 1.4 s at 128, 39.6 s at 256 for the same 16 frames): stop climbing once a count is 2.5x off the best so far
        if time.perf_counter() > budget or probe[th] > 2.5 * min(probe.values()):
            break
    best_t = min(probe, key=probe.get)
    torch.set_num_threads(best_t)

    def rate(mel, img, secs, max_n):
        n, t0 = 0, time.perf_counter()
        while True:
            fwd(mel, img)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= secs or n >= max_n:
                return n * len(mel) / dt, n, dt

    r16, n16, dt16 = rate(mel16, img16, 0.3 * seconds, 64)
    mel128, img128 = inputs(128, 12)
    r128, n128, dt128 = rate(mel128, img128, 0.3 * seconds, 8)
    # the BASELINE batch at higher thread counts as well (the probe above climbs at B=16 and stops at its knee): ONE forward of
    # 128 frames per count, abandoned once a count takes more than 8 s
    probe128 = {str(best_t): round(128.0 / r128, 3)}
    for th in (32, 64, 128):
        if th > avail or th == best_t:
            continue
        torch.set_num_threads(th)
        fwd(mel16[:2], img16[:2])
        t0 = time.perf_counter()
        fwd(mel128, img128)
        probe128[str(th)] = round(time.perf_counter() - t0, 3)
        if probe128[str(th)] > 8.0:
            break
    b128 = min(probe128, key=probe128.get)
    r128 = max(r128, 128.0 / probe128[b128])
    torch.set_num_threads(best_t)
    # the other two CPU legs of the path beside the model - audio.melspectrogram (STFT + mel basis, restated numpy,
    # oracle/audio_ref.py) over the audio of 128 frames at 25 fps, and datagen (resize-free crop batch -> masked 6-channel float
    # input + mel windows, oracle/datagen_ref.py) for 128 frames
    from oracle import audio_ref
    wav = synth.noise_wav(int(16000 * (128 + 8) / 25.0), seed=200)
    audio_ref.melspectrogram(wav[:16000])
    t0 = time.perf_counter()
    nmel = 0
    while time.perf_counter() - t0 < 0.06 * seconds or nmel < 2:
        audio_ref.melspectrogram(wav)
        nmel += 1
    mel_ms = (time.perf_counter() - t0) * 1e3 / nmel
    f128, m128 = synth.face_crops_u8(128, seed=3), synth.mel_windows(128, seed=3)
    t0 = time.perf_counter()
    ndg = 0
    while time.perf_counter() - t0 < 0.06 * seconds or ndg < 2:
        datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(f128, m128))
        ndg += 1
    dg_ms = (time.perf_counter() - t0) * 1e3 / ndg
    out = {"value": round(max(r16, r128), 2), "unit": "face-frames/sec", "cores": best_t, "kind": kind,
           "host_cpu": model_name, "host_logical_cpus": avail,
           "at_batch_128": round(r128, 2), "at_batch_16": round(r16, 2),
           "thread_probe_s_per_16_frames": {str(k): round(v, 3) for k, v in sorted(probe.items())},
           "thread_probe_s_per_128_frames": probe128,
           "mel_ms_per_128_frames": round(mel_ms, 2), "datagen_ms_per_128_frames": round(dg_ms, 2),
           "with_mel_and_datagen_at_batch_128": round(128.0 / (128.0 / r128 + (mel_ms + dg_ms) * 1e-3), 2),
           "pin": ("oracle.models_ref == /root/reference models.Wav2Lip by torch.equal on torch 2.10.0+rocm7.0 CPU (tests/golden/"
                   "make_golden.py, outputs committed as tests/golden/golden_v1.npz and re-checked by tests/test_oracle.py); this "
                   "run: torch %s" % torch.__version__) if kind == "port" else "the reference's own modules",
           "sample": "%s Wav2Lip forward, torch CPU fp32, eval, no_grad, %d of %d logical CPUs (best of the probe): "
                     "%d batches of 16 in %.1f s and %d batches of 128 in %.1f s"
                     % ("/root/reference models.Wav2Lip" if kind == "reference" else "oracle.models_ref (reference restated)",
                        best_t, avail, n16, dt16, n128, dt128)}
    pred = None
    if check_faces is not None:
        img, mel = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(check_faces, check_mels))
        pred = fwd(torch.from_numpy(mel), torch.from_numpy(img)).numpy()
    return out, pred, datagen_ref.frames_to_u8


def source_fingerprint():
    """sha256 (first 16 hex digits) over the kernel sources: PMC-derived numbers are only valid for the code they were
    measured on"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "wav2lip_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    table = os.path.join(ROOT, "wav2lip_amd", "tune_table.json")
    if os.path.exists(table):
        with open(table, "rb") as fh:
            h.update(b"tune_table\0" + fh.read())
    return h.hexdigest()[:16]


def hbm_traffic(batch):
    """HBM bytes per step (all launches of one 128-frame pass) from the newest committed rocprofv3 PMC passes (FETCH_SIZE x2,
    the guide's gfx950 correction, + WRITE_SIZE; profiles/r*/traffic.json written by tools/collect_profiles.sh).  PMC counters
    cannot be read from inside the timed process, so the value is a stored measurement: it is reported only when the file
    was measured on exactly these kernel sources + tune table (source_fingerprint) at this batch, otherwise null."""
    fp = source_fingerprint()
    prof = os.path.join(ROOT, "profiles")
    for rnd in sorted((d for d in os.listdir(prof) if d.startswith("r")), reverse=True) if os.path.isdir(prof) else []:
        try:
            with open(os.path.join(prof, rnd, "traffic.json")) as fh:
                t = json.load(fh)
        except (OSError, ValueError):
            continue
        if t.get("source_fingerprint") == fp and int(t.get("frames_per_step", 0)) == batch:
            return int(t["hbm_bytes_per_step"])
        return None     # the newest measurement is of other code: stale, not reported
    return None


def train_configs():
    """BASELINE configs[2] / [3] / [4] (SyncNet step at batch 512, wav2lip_train step and hq_wav2lip_train GAN step at batch 64) timed by
    tools/train_bench.py in a subprocess, fp32 contractions and the bf16 ones the configs name: reported next to the headline
    metric, never part of `value`.  Any failure is reported as a string instead of breaking the bench line."""
    import subprocess
    out = []
    for prec in ("f32", "bf16"):
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "--cfg", "3", "4", "5", "--steps", "3",
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
    if args.exact:
        os.environ["W2L_EXACT"] = "1"       # read by wav2lip_amd/_lib.py when the library is loaded (below, and in every rank)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)       # does not return: N ranks of this file under torch.distributed.run
    if world != args.gpus:
        args.gpus = world
    if args.backend == "gloo" and not args.dry_run:
        sys.exit("bench.py: --backend gloo exists for --dry-run only; the measured exchange step is RCCL (nccl)")
    if args.dry_run:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if args.inject_failure == rank:
            sys.exit("bench.py: injected failure on rank %d (launch test)" % rank)
        dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.dist_timeout))
        return dry_run(args, world, rank, dist)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU path to measure)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                timeout=datetime.timedelta(seconds=args.dist_timeout))

    from wav2lip_amd import synthetic as synth    # synthetic weights/inputs (no datasets/checkpoints offline)
    from wav2lip_amd import audio, models
    from wav2lip_amd.inference import PipelinedRunner, mel_chunk_starts
    from wav2lip_amd.sharding import PipelinedFrameGatherer

    B = args.batch
    G = models.Wav2Lip()
    sd = synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0)
    G.load_state_dict(sd)
    G = G.to(dev).eval()
    # THE PRODUCT'S RUNNER is what is timed: inference.PipelinedRunner (the loop of inference.lipsync / inference.main) with
    # `--pipeline` lanes - each lane a Wav2LipRunner with its own generator buffers on its own HIP stream
    depth = max(1, args.pipeline)
    runner = PipelinedRunner(G, batch_size=B, depth=depth)

    # synthetic inputs resident in HBM: B uint8 crops + a mel spectrogram of random 16 kHz audio with B windows
    faces_host = synth.face_crops_u8(B, seed=100 + rank)
    faces = torch.from_numpy(faces_host).to(dev)
    fps = 25.0
    nsamp = int(16000 * (B + 8) / fps)
    mel = audio.melspectrogram_device(synth.noise_wav(nsamp, seed=200 + rank), dev)
    starts_host = mel_chunk_starts(mel.shape[1], fps)[:B]
    starts = torch.tensor(starts_host, dtype=torch.int32, device=dev)
    assert starts.numel() == B
    gather = (PipelinedFrameGatherer(dist, world, (B, 96, 96, 3), torch.uint8, dev, depth=max(2, args.pipeline))
              if world > 1 else None)

    # launch configurations: the committed shape-keyed tune table + heuristic (bit-reproducible) unless --autotune
    g = G.graph(B, 96, 96, dev)
    config_source = ("exact tune table (wav2lip_amd/tune_table_exact.json, no F(4x4) Winograd) + heuristic" if args.exact
                     else "tune table (wav2lip_amd/tune_table.json) + heuristic")
    from wav2lip_amd import _lib as _w2l_lib, engine
    if _w2l_lib.NO_SPLIT and not args.exact:
        config_source = ("tune table (wav2lip_amd/tune_table.json) with its split-operand entries replaced by their fp32-pipe "
                         "predecessors (tune_table_nosplit.json, W2L_SPLIT=0) + heuristic")
    if engine.plan_configs_enabled() and engine.plan_config_source("generator_96", B) is not None:
        config_source = ("per-plan launch list of batch %d (wav2lip_amd/plan_configs.json: batch %d is not in the tune table)"
                         % (engine.plan_config_source("generator_96", B), B))
    if args.tune_cache and os.path.exists(args.tune_cache):
        g.plan.load_configs(args.tune_cache)
        config_source = "file " + os.path.basename(args.tune_cache)
    elif args.autotune:
        g.plan.autotune()
        config_source = "stopwatch autotune in this process"
        if args.tune_cache and rank == 0:
            g.plan.save_configs(args.tune_cache)
    lib = runner.lanes[0].lib
    from wav2lip_amd._lib import check, current_stream, ptr
    # every lane holds a full plan over its own buffers with the same launch configurations as the first
    graphs = [g] + [lane._graph(B) for lane in runner.lanes[1:]]
    assert graphs[0] is runner.lanes[0]._graph(B)
    for gg in graphs[1:]:
        for i, (_, t_, k_) in enumerate(g.plan.configs()):
            gg.plan.set_config(i, t_, k_)
        gg.plan.tuned = g.plan.tuned
    main_stream = torch.cuda.current_stream()
    streams = runner.streams
    outs_u8 = [torch.empty((B, 96, 96, 3), dtype=torch.uint8, device=dev) for _ in range(depth)]
    counter = [0]
    nthreads = 1                     # one host thread enqueues (the product's loop); --launch-threads is accepted and ignored

    def step():
        """one batch through the product's runner: w2l_datagen_pack + w2l_mel_gather + the generator plan + w2l_frames_to_u8 on
        lane (step % depth); with N > 1 the frames land in the gatherer's send slot and the all-gather is issued behind them"""
        k = runner.n % depth
        counter[0] += 1
        if gather is None:
            runner.submit(faces, mel=mel, starts=starts, out=outs_u8[k])
            return
        with torch.cuda.stream(streams[k]):
            dst = gather.slot()
        runner.submit(faces, mel=mel, starts=starts, out=dst)
        with torch.cuda.stream(streams[k]):
            gather.submit()          # asynchronous: this batch's frames cross xGMI while the next batch is computed

    def run_steps(n):
        for _ in range(n):
            step()

    def fence():
        if gather is not None:
            for st in streams:
                with torch.cuda.stream(st):
                    gather.drain()   # every all-gather of the timed region has completed before the clock stops
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not a step: every (buffer set, stream) pair runs its plan once so that lazily allocated workspaces exist before
    # the W warmup steps (with W < depth a pair would otherwise see its first launch inside the timed region)
    for k, gg in enumerate(graphs):
        with torch.cuda.stream(streams[k]):
            gg.run()
    torch.cuda.synchronize()
    assert counter[0] == 0 and runner.n == 0
    run_steps(args.warmup)
    fence()
    # The timed region: EXACTLY --steps steps between two (barrier + synchronize) fences, wall clock, max over ranks.  It is
    # repeated --windows times and the median window is the reported one (a single 20 x 6 ms window is a 0.12 s sample).
    wall, gpu_ms, local_wall = [], [], []
    for _ in range(max(1, args.windows)):
        ev_b, ev_e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev_b.record(main_stream)     # HIP events bracket the timed region on the launch streams: every stream starts behind
        for st in streams:           # ev_b and ev_e is recorded after all of them have been joined
            st.wait_stream(main_stream)
        run_steps(args.steps)
        for st in streams:
            main_stream.wait_stream(st)
        ev_e.record(main_stream)
        fence()
        dt = dt_local = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        wall.append(dt)
        local_wall.append(dt_local)
        gpu_ms.append(ev_b.elapsed_time(ev_e) / args.steps)
    # The sustained window: one more timed region (same fences, wall clock, max over ranks) long enough to fill
    # --sustained-seconds of GPU time - a long clip is minutes of steady state, the --steps windows above are 0.1 s bursts.
    # The shader clock is SAMPLED across all of it: a 2 ms probe (s_memtime ticks per 100 MHz s_memrealtime tick) on its own stream
    # every `probe_every` steps.  (A probe that spins through the whole window was measured first, profiles/r04/b_*: its one wave
    # keeps a CU from taking a 512-thread workgroup that needs every register of the CU, the persistent kernels then run their
    # last workgroup alone, and the window read 14.0k frames/s instead of 24.6k.  Sampled, the probes hold one CU for ~2 % of the
    # window.)  EVERY rank runs the window (its steps contain the all-gather); rank 0 alone launches the probes.
    clock_mhz, sustained = None, None
    sust_steps = int(np.ceil(args.sustained_seconds * 1e3 / max(1e-3, sorted(gpu_ms)[len(gpu_ms) // 2]))) if args.sustained_seconds > 0 else 0
    sust_steps = max(args.steps, sust_steps) if sust_steps else args.steps
    if dist is not None:                                    # the same step count on every rank
        t = torch.tensor([sust_steps], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sust_steps = int(t.item())
    probe_every = max(4, sust_steps // 24)
    nprobe = (sust_steps + probe_every - 1) // probe_every if rank == 0 else 0
    probe_stream = torch.cuda.Stream(device=dev) if rank == 0 else None
    ticks = torch.zeros((max(1, nprobe), 2), dtype=torch.int64, device=dev)
    fence()
    t0 = time.perf_counter()
    done = 0
    while done < sust_steps:
        if rank == 0:
            # behind the newest enqueued step: the host runs ahead of the GPU, an unordered probe stream would take all its
            # samples in the first milliseconds of the window
            probe_stream.wait_stream(streams[(counter[0] - 1) % depth])
            with torch.cuda.stream(probe_stream):
                check(lib.w2l_clock_probe(current_stream(), 2000, ptr(ticks[done // probe_every])), "clock_probe")
        n = min(probe_every, sust_steps - done)
        run_steps(n)
        done += n
    fence()
    dt_s = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt_s], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())
    clock_samples = None
    if rank == 0:
        tk = ticks.tolist()
        mhz = [100.0 * c / r for c, r in tk if r > 0]
        if mhz:
            clock_mhz = round(100.0 * sum(c for c, r in tk if r > 0) / sum(r for c, r in tk if r > 0), 1)
            clock_samples = {"n": len(mhz), "min_mhz": round(min(mhz), 1), "max_mhz": round(max(mhz), 1)}
    if args.sustained_seconds > 0:
        sustained = {"steps": sust_steps, "seconds": round(dt_s, 3), "value": round(world * B * sust_steps / dt_s, 1),
                     "ms_per_step": round(dt_s / sust_steps * 1e3, 3), "clock_samples": clock_samples}
    # ---- three more rates beside `value` (one GPU; none of them is `value`):
    #   with_mel_and_d2h    the same runner and depth, every step ALSO computes the mel spectrogram of its 128 frames' audio
    #                       (audio.melspectrogram_device on a wav resident in HBM: north_star names the STFT as hot path) and copies
    #                       its uint8 frames to pinned host memory (what inference.lipsync / main do with every batch)
    #   one_in_flight       the product's runner at depth 1: strictly one batch at a time
    #   exact               this file with --exact in a subprocess (the library reads W2L_EXACT when it is loaded)
    extra = {}
    if world == 1 and not args.no_cpu_baseline:
        def windows_of(step_fn, nwin=3):
            rates = []
            for _ in range(nwin):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step_fn()
                torch.cuda.synchronize()
                rates.append(B * args.steps / (time.perf_counter() - t0))
            return round(sorted(rates)[len(rates) // 2], 1)

        wav_dev = torch.from_numpy(synth.noise_wav(nsamp, seed=300)).to(dev)
        host_u8 = [torch.empty((B, 96, 96, 3), dtype=torch.uint8).pin_memory() for _ in range(depth)]

        def step_mel_d2h():
            k = runner.n % depth
            counter[0] += 1
            mel_i = audio.melspectrogram_device(wav_dev, dev)                  # on the caller's stream; the lane waits for it
            runner.submit(faces, mel=mel_i, starts=starts, out=outs_u8[k])
            with torch.cuda.stream(streams[k]):
                host_u8[k].copy_(outs_u8[k], non_blocking=True)                # ordered before the lane's next batch
        for _ in range(depth):
            step_mel_d2h()
        extra["value_with_mel_and_d2h"] = windows_of(step_mel_d2h)
        extra["with_mel_and_d2h_note"] = ("per step: w2l_melspectrogram of %d samples (HBM-resident wav) + the step + %d bytes of uint8 "
                                          "frames to pinned host memory; same runner, %d batches in flight" % (nsamp, B * 96 * 96 * 3, depth))
        # `mel` is restored for the parity check below: the last batch of the loop above ran on mel_i of another wav
        step()
        torch.cuda.synchronize()
        if depth > 1:
            one = PipelinedRunner(G, batch_size=B, depth=1)      # lane 0's buffers, one stream
            one_out = torch.empty((B, 96, 96, 3), dtype=torch.uint8, device=dev)
            for _ in range(3):
                one.submit(faces, mel=mel, starts=starts, out=one_out)
            extra["one_in_flight_value"] = windows_of(lambda: one.submit(faces, mel=mel, starts=starts, out=one_out))
            step()               # lane 0 of the timed runner holds the frames of its own last batch again (parity below)
            torch.cuda.synchronize()
        else:
            extra["one_in_flight_value"] = None
        if not args.exact and depth > 1 and not os.environ.get("W2L_TUNE_TABLE"):
            # strictly serial serving with the launch table tuned for it (wav2lip_amd/tune_table_one_in_flight.json: the default table
            # plus the conv_wino2s entries that win with ONE batch in flight and lose with four): its measured place
            import subprocess
            try:
                env1 = dict(os.environ, W2L_TUNE_TABLE=os.path.join(ROOT, "wav2lip_amd", "tune_table_one_in_flight.json"))
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-train-configs", "--windows", "3",
                                     "--sustained-seconds", "0", "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch",
                                     str(B), "--pipeline", "1"], capture_output=True, text=True, timeout=240, env=env1)
                extra["one_in_flight_value_with_its_table"] = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])["value"]
            except Exception as e:      # noqa: BLE001
                extra["one_in_flight_value_with_its_table"] = None
                extra["one_in_flight_table_note"] = "failed: " + str(e)[:200]
        if not args.exact:
            import subprocess
            try:
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--exact", "--no-cpu-baseline", "--no-train-configs",
                                     "--windows", "3", "--sustained-seconds", "0", "--steps", str(args.steps), "--warmup",
                                     str(args.warmup), "--batch", str(B), "--pipeline", str(depth)],
                                    capture_output=True, text=True, timeout=240)
                ex = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
                extra["exact_value"] = ex["value"]
                extra["exact_note"] = "python bench.py --exact (no F(4x4,3x3) Winograd launches; " + ex["config"]["launch_configs"] + ")"
            except Exception as e:      # noqa: BLE001 - the headline line must survive
                extra["exact_value"] = None
                extra["exact_note"] = "failed: " + str(e)[:200]
    order = sorted(range(len(wall)), key=lambda i: wall[i])
    med = order[len(order) // 2]
    dt = wall[med]
    step_ms = gpu_ms[med]            # GPU time per batch over the median window (HIP events on the launch streams)
    frames = world * B * args.steps
    per_rank = [round(B * args.steps / local_wall[med], 1)]
    if dist is not None:
        rates = torch.zeros(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(rates, torch.tensor([per_rank[0]], device=dev, dtype=torch.float64))
        per_rank = [round(float(v), 1) for v in rates.tolist()]

    # ---- roofline: FLOPs the matrix cores EXECUTE (padded tiles / K, 16 products per 2x2 tile on Winograd launches) over the
    # event time of the timed region; the nominal direct-convolution count (SURVEY.md 8d, 7.934 GFLOP/frame) beside it
    # Launches of the split-operand families ("split": conv_igemm_bf16_kernel<.., 3>, "wino2s": conv_wino2s_kernel, "tp2s": conv_tp2s_kernel - fp32 operands
    # as three bf16 pieces, fp32-accurate result) report the bf16 matrix-core FLOPs they execute: six piece products per product.
    resolved = g.plan.resolved()
    BF16_FAMS = ("split", "wino2s", "tp2s", "stem7s", "k3s")                # families whose executed FLOPs are bf16 matrix-core FLOPs
    exec_f32 = float(sum(f for _, f, fam, _ in resolved if fam not in BF16_FAMS))
    exec_bf16 = float(sum(f for _, f, fam, _ in resolved if fam in BF16_FAMS))
    exec_flop = exec_f32 + exec_bf16 / 6.0
    nominal_flop = 2.0 * g.plan.macs()
    products_tf = exec_flop / (step_ms * 1e-3) / 1e12          # fp32-accurate products executed per second (x2), whichever pipe
    # ONE meaning for achieved / frac: matrix-core work in fp32-pipe TFLOP/s, a bf16 MFMA FLOP counted at the 1/16 of an fp32
    # MFMA FLOP's pipe time it occupies (2516.6 = 16 x 157.3) - so frac = achieved / 157.3 is exactly the share of the step the
    # matrix cores would need at their peak rates, and no launch can exceed 1
    achieved = (exec_f32 + exec_bf16 / 16.0) / (step_ms * 1e-3) / 1e12
    # the dominant kernel on its own: per-launch HIP events of one serial pass of the plan (outside the timed region)
    prof = g.plan.profile(reps=3)
    fam_ms, fam_fl, fam_n = {}, {}, {}
    for (name, ms, _), (_, fl, fam, _) in zip(prof, resolved):
        fam_ms[fam] = fam_ms.get(fam, 0.) + ms
        fam_fl[fam] = fam_fl.get(fam, 0.) + fl / (16.0 if fam in BF16_FAMS else 1.0)       # fp32-pipe equivalents, as `achieved`
        fam_n[fam] = fam_n.get(fam, 0) + 1
    serial_ms = sum(fam_ms.values())
    dom = max(fam_ms, key=fam_ms.get)
    kname = {"wino": "conv_wino_f32_kernel (Winograd F(2x2,3x3), fp32 MFMA)",
             "wino2": "conv_wino2_f32_kernel (Winograd F(2x2,3x3), position-split waves, fp32 MFMA)",
             "wino4": "conv_wino4_f32_kernel (Winograd F(4x4,3x3), 36 positions split over 8 waves, fp32 MFMA)",
             "tp2": "conv_tp2_f32_kernel (stride-2 transposed 3x3, four phases per workgroup, fp32 MFMA)",
             "igemm": "conv_igemm_f32_kernel (implicit GEMM, fp32 MFMA)",
             "split": "conv_igemm_bf16_kernel<.., 3> (implicit GEMM, fp32 operands as three bf16 pieces, bf16 MFMA)",
             "wino2s": "conv_wino2s_kernel (Winograd F(2x2,3x3), transformed operands as three bf16 pieces, bf16 MFMA)",
             "tp2s": "conv_tp2s_kernel (stride-2 transposed 3x3, four phases per workgroup, operands as three bf16 pieces, bf16 MFMA)",
             "stem7s": "conv_stem7s_kernel (7x7 first layer, region staged and split once, bf16 MFMA)",
             "k3s": "conv_k3s_kernel (direct 3x3 for 32-cout layers + fused head, operands as three bf16 pieces, bf16 MFMA)"}
    dom_tf = fam_fl[dom] / (fam_ms[dom] * 1e-3) / 1e12
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
        "per_rank_frames_per_s": per_rank,
        "timed_loop": "wav2lip_amd.inference.PipelinedRunner(depth=%d).submit - the loop of inference.lipsync / inference.main" % depth,
        "windows": {"n": len(wall), "reported": "median", "value_min": round(frames / max(wall), 1),
                    "value_max": round(frames / min(wall), 1),
                    "sustained_value": sustained["value"] if sustained else None, "sustained": sustained},
        "config": {"workload": "Wav2Lip generator fp32 inference, batch=%d synthetic 96x96x6 crops + random mel per GPU "
                               "(BASELINE configs[1]); datagen pack + mel gather + generator + uint8 frames%s"
                               % (B, " + RCCL all-gather of uint8 frames" if world > 1 else ""),
                   "frames_per_gpu_per_step": B, "parallelism": "dp%d" % world, "batches_in_flight_per_gpu": depth,
                   "launch_threads": nthreads,
                   "weights": "random-init (wav2lip_amd.synthetic seed 0)", "launch_configs": config_source,
                   "collective_world_size": (dist.get_world_size() if dist is not None else 1),
                   "collective_backend": (dist.get_backend() if dist is not None else None)},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": hbm_traffic(B),
                     "what": "matrix-core FLOPs EXECUTED per step (all %d fused conv launches; padded tiles and K, Winograd layers at 16 "
                             "(F(2x2,3x3)) or 9 (F(4x4,3x3)) instead of 36 products per 2x2 outputs) in fp32-pipe TFLOP/s - fp32 MFMA FLOPs + "
                             "bf16 MFMA FLOPs / 16 (a bf16 FLOP occupies 1/16 of an fp32 FLOP's pipe time: 2516.6 = 16 x 157.3) - over "
                             "the GPU time per step (HIP events over the median timed window): frac = the share of the step the matrix "
                             "cores need at their peak rates (the `pipe_time_frac` of earlier rounds)" % len(resolved),
                     "executed_by_pipe": {"fp32_mfma_gflop": round(exec_f32 / 1e9, 2), "bf16_mfma_gflop": round(exec_bf16 / 1e9, 2),
                                          "bf16_launches": sum(1 for _, _, fam, _ in resolved if fam in BF16_FAMS),
                                          "fp32_pipe_time_frac": round(exec_f32 / PEAK_FP32_MFMA_TFLOPS / 1e12 / (step_ms * 1e-3), 4),
                                          "bf16_pipe_time_frac": round(exec_bf16 / PEAK_BF16_MFMA_TFLOPS / 1e12 / (step_ms * 1e-3), 4),
                                          "bf16_pipe_tflops": round(exec_bf16 / (step_ms * 1e-3) / 1e12, 1), "bf16_peak": PEAK_BF16_MFMA_TFLOPS},
                     "fp32_accurate_products_tflops": round(products_tf, 2),
                     "fp32_accurate_products_note": "2 x the fp32-accurate multiply-adds executed per second on either pipe (a split-"
                                                    "operand launch's bf16 FLOPs / 6): the rate earlier rounds divided by 157.3 and "
                                                    "called frac; a rate, not a fraction of any one peak",
                     "executed_gflop_per_step": round(exec_flop / 1e9, 2),
                     "gpu_ms_per_step": round(step_ms, 3),
                     "nominal_tflops": round(nominal_flop / (step_ms * 1e-3) / 1e12, 2),
                     "nominal_gflop_per_step": round(nominal_flop / 1e9, 2),
                     "nominal_gflop_per_frame": round(nominal_flop / 1e9 / B, 4),
                     "dominant_kernel": {"name": kname[dom], "launches_per_step": fam_n[dom],
                                         "executed_gflop_per_step": round(fam_fl[dom] / 1e9, 2),
                                         "ms_per_step_serial": round(fam_ms[dom], 3),
                                         "avg_launch_ms": round(fam_ms[dom] / fam_n[dom], 4),
                                         "share_of_serial_step": round(fam_ms[dom] / serial_ms, 3),
                                         "achieved": round(dom_tf, 2), "frac": round(dom_tf / PEAK_FP32_MFMA_TFLOPS, 4)},
                     "serial_ms_per_step": round(serial_ms, 3),
                     "sustained_clock_mhz": clock_mhz,
                     "frac_at_sustained_clock": (round(achieved / (PEAK_FP32_MFMA_TFLOPS * clock_mhz / 2400.0), 4)
                                                 if clock_mhz else None),
                     "clock_note": "peak 157.3 TFLOP/s is 256 CUs x 256 fp32 MFMA FLOP/clk x 2.4 GHz; sustained_clock_mhz is "
                                   "the mean of 2 ms shader-clock samples taken across the sustained window of this workload",
                     "source_fingerprint": source_fingerprint()},
    }
    # the rate this launch plan would reach if every launch ran its matrix pipe at peak: value / frac (the bound `frac` implies)
    result["bound_frames_per_s"] = round(result["value"] / max(1e-9, result["roofline"]["frac"]), 1)
    result.update(extra)
    nsplit = sum(1 for _, _, fam, _ in resolved if fam in BF16_FAMS)
    if nsplit:
        result["arithmetic"] = ("fp32 tensors, fp32 accumulation; %d of %d conv launches multiply on the bf16 matrix cores with every "
                                "fp32 operand as the exact sum of three bf16 pieces (six piece products per product, dropped terms "
                                "< 2^-24): error against fp64 not above the fp32 MFMA kernels' (tests/test_conv_gpu.py, "
                                "tools/split_bf16_accuracy.py)" % (nsplit, len(resolved)))
    if args.profile_layers and rank == 0:
        for (name, ms, m), (_, fl, fam, cfg) in zip(prof, resolved):
            sys.stderr.write("%-34s %8.3f ms %6.1f%%  nominal %7.2f  executed %7.2f TFLOP/s%s %-5s cfg %d ks %d\n"
                             % (name, ms, 100 * ms / serial_ms, 2 * m / ms / 1e9, fl / ms / 1e9, " (bf16)" if fam in BF16_FAMS else "",
                                fam, cfg[0], cfg[1]))
        sys.stderr.write("sum %.3f ms\n" % serial_ms)
    if world == 1 and not args.no_cpu_baseline:
        # CPU baseline + in-run parity: the frames the timed loop just produced (lane of the last step) against the CPU
        # forward of the same crops / mel windows.  Tolerances: north star 1e-3 L-inf on fp32 pixels; uint8 frames may differ
        # where v*255 sits within rounding of an integer (truncation), bounded at 0.1 % of the bytes.
        ncheck = B                   # every frame of the batch (the CPU forward of 128 frames is ~2.5 s)
        k_last = (counter[0] - 1) % depth
        got_u8 = outs_u8[k_last][:ncheck].cpu().numpy()
        got_f32 = graphs[k_last].output_nchw()[:ncheck].cpu().numpy()
        mel_host = mel.cpu().numpy()
        mels_chk = np.stack([mel_host[:, s_:s_ + 16] for s_ in starts_host[:ncheck]])
        base, pred, to_u8 = cpu_baseline(sd, args.cpu_seconds, faces_host[:ncheck], mels_chk)
        result["cpu_baseline"] = base
        linf = float(np.abs(got_f32 - pred).max())
        mism = int((got_u8.astype(np.int32) != to_u8(pred).astype(np.int32)).sum())
        result["parity"] = {"frames_checked": ncheck, "fp32_linf": linf, "fp32_tolerance": 1e-3,
                            "u8_mismatches": mism, "u8_values": int(got_u8.size), "against": base["kind"]}
        assert linf <= 1e-3 and mism <= got_u8.size // 1000, "timed path failed the in-run parity check: %s" % result["parity"]
    if world == 1 and not args.no_train_configs and not args.no_cpu_baseline:
        result["other_configs"] = train_configs()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
