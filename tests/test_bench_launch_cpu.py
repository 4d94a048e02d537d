"""`python bench.py --gpus N` the way the driver's SCALE step runs it (no torch.distributed.run, no WORLD_SIZE): bench.py must
start its own ranks.  CPU coverage: --backend gloo --dry-run executes the launch, partition, pipelined all-gather, fence and
max-over-ranks timing code with a byte pattern in place of the generator's frames (no kernels) and prints the one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=240,
                       env=env, cwd="/tmp")
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    return json.loads(lines[0])


def test_bench_gpus2_launches_its_own_ranks_and_gathers_in_order():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "7", "--warmup", "2", "--windows", "3", "--batch", "8"])
    assert r["n_gpus"] == 2 and r["dry_run"] is True and r["gather_verified"] is True
    assert r["config"]["collective_world_size"] == 2 and r["config"]["collective_backend"] == "gloo"
    assert r["frame_shards"] == [[0, 56], [56, 112]]              # contiguous chunks of the 2 x 8 x 7 frame list
    assert len(r["per_rank_frames_per_s"]) == 2 and all(v > 0 for v in r["per_rank_frames_per_s"])
    assert r["steps"] == 7 and r["warmup"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    # whole-job aggregate = all ranks' frames over the slowest rank's time: never above the sum of the per-rank rates
    assert r["value"] <= sum(r["per_rank_frames_per_s"]) * 1.001


def test_bench_dry_run_three_ranks_pipeline_two():
    r = _run(["--gpus", "3", "--backend", "gloo", "--dry-run", "--steps", "5", "--warmup", "1", "--windows", "1", "--batch", "4",
              "--pipeline", "2"])
    assert r["n_gpus"] == 3 and r["gather_verified"] is True and r["config"]["batches_in_flight_per_gpu"] == 2
    assert r["frame_shards"] == [[0, 20], [20, 40], [40, 60]]


def test_bench_dry_run_eight_ranks_the_scale_step_shape():
    """N = 8, the widest launch the driver's SCALE step makes (`python bench.py --gpus 8 ...`): eight contiguous shards, four
    batches in flight per rank, every gathered batch in rank order"""
    r = _run(["--gpus", "8", "--backend", "gloo", "--dry-run", "--steps", "6", "--warmup", "2", "--windows", "2", "--batch", "4"])
    assert r["n_gpus"] == 8 and r["gather_verified"] is True and r["config"]["collective_world_size"] == 8
    assert r["frame_shards"] == [[24 * i, 24 * (i + 1)] for i in range(8)] and len(r["per_rank_frames_per_s"]) == 8
    assert r["config"]["batches_in_flight_per_gpu"] == 4


def test_a_failing_rank_ends_the_job_with_a_nonzero_exit_instead_of_hanging():
    """rank 1 exits before the rendezvous (--inject-failure): `python bench.py --gpus 2` and `python tools/train_bench.py --gpus 2`
    must come back non-zero well inside the collective timeout - the elastic agent stops rank 0 - and print no result line"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for script, extra in ((os.path.join(ROOT, "bench.py"), ["--steps", "2", "--warmup", "1", "--windows", "1", "--batch", "4"]),
                          (os.path.join(ROOT, "tools", "train_bench.py"), ["--cfg", "4", "--steps", "1", "--warmup", "0"])):
        t0 = time.time()
        p = subprocess.run([sys.executable, script, "--gpus", "2", "--backend", "gloo", "--dry-run", "--inject-failure", "1",
                            "--dist-timeout", "30"] + extra, capture_output=True, text=True, timeout=200, env=env, cwd="/tmp")
        assert p.returncode != 0, (script, p.stdout[-300:])
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert time.time() - t0 < 120


def test_train_bench_gpus4_launches_its_own_ranks_and_averages_bucketed_gradients():
    """`python tools/train_bench.py --gpus 4 --backend gloo --dry-run`: four ranks, the generator's parameter blocks walked in
    backward order through the bucketed GradReducer, every averaged gradient checked, one JSON line with per-rank step times"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "--gpus", "4", "--backend", "gloo", "--dry-run",
                        "--cfg", "4", "--steps", "1", "--warmup", "0", "--bucket-mb", "16"], capture_output=True, text=True,
                       timeout=240, env=env, cwd="/tmp")
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    r = json.loads(lines[0])
    assert r["n_gpus"] == 4 and r["collective_world_size"] == 4 and r["collective_backend"] == "gloo" and r["dry_run"] is True
    assert r["gradients_verified"] is True and r["reduced_parameters"] == 36298035 and r["buckets_per_step"] >= 5
    assert len(r["per_rank_ms_per_step"]) == 4 and r["ms_per_step"] == max(r["per_rank_ms_per_step"])


def test_no_collective_work_sits_under_a_rank_condition():
    """bench.py's measured path: a `step()` (it submits the all-gather), a `fence()` (it drains and barriers) or a `dist.` /
    `gather.` call lexically inside `if rank == 0:` / `if rank != 0:` would be executed by one rank only and hang every N > 1
    run (the rank-0-only clock-probe window of an earlier revision did exactly that, on a path no single-GPU box exercises).
    Calls that only READ the process group (world size, backend name) are allowed."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    readers = {"get_world_size", "get_backend", "get_rank"}

    def mentions_rank(test):
        return any(isinstance(n, ast.Name) and n.id == "rank" for n in ast.walk(test))

    def offending_calls(nodes):
        bad = []
        for stmt in nodes:
            for n in ast.walk(stmt):
                if not isinstance(n, ast.Call):
                    continue
                f = n.func
                if isinstance(f, ast.Name) and f.id in ("step", "step_on", "run_steps", "fence"):
                    bad.append((f.id, n.lineno))
                if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in ("dist", "gather") \
                        and f.attr not in readers:
                    bad.append((f.value.id + "." + f.attr, n.lineno))
        return bad
    found = []
    for node in ast.walk(main):
        if isinstance(node, ast.If) and mentions_rank(node.test):
            found += offending_calls(node.body) + offending_calls(node.orelse)
        if isinstance(node, ast.IfExp) and mentions_rank(node.test):
            found += offending_calls([ast.Expr(node.body), ast.Expr(node.orelse)])
    assert not found, "collective work under a rank condition in bench.main(): %s" % found
