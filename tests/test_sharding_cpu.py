"""N > 1 path on CPU: contiguous-chunk partition and in-order frame all-gather, 2 gloo processes."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from wav2lip_amd.sharding import (FrameGatherer, PipelinedFrameGatherer, gather_frames_in_order, shard_counts,
                                  shard_range)


def test_partition_covers_every_item_once():
    for n in (0, 1, 7, 8, 72, 129, 1000):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert sum(shard_counts(n, w)) == n
            assert max(shard_counts(n, w)) <= -(-n // w)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_total * 2 * 2 * 3, dtype=torch.int64).remainder(251).to(torch.uint8).view(n_total, 2, 2, 3)
        lo, hi = shard_range(n_total, rank, world)
        got = gather_frames_in_order(dist, full[lo:hi].clone(), n_total, rank, world)
        ok_ragged = torch.equal(got, full)
        g = FrameGatherer(dist, world, torch.device("cpu"))
        mine = torch.full((4, 2, 2, 3), rank + 1, dtype=torch.uint8)
        allf = g.all_gather(mine)
        ok_equal = all(bool((allf[4 * r:4 * r + 4] == r + 1).all()) for r in range(world)) and allf.shape[0] == 4 * world
        pg = PipelinedFrameGatherer(dist, world, (4, 2, 2, 3), torch.uint8, torch.device("cpu"))
        ok_pipe = True
        outs = []
        for step in range(5):                                     # 5 batches through 2 rotating buffer pairs
            pg.slot().fill_(10 * step + rank + 1)
            outs.append((step, pg.submit()))
            if step >= 1:                                         # batch step-1 is complete once its slot is reused or drained
                pass
        last = pg.drain()
        ok_pipe = all(bool((last[4 * r:4 * r + 4] == 10 * 4 + r + 1).all()) for r in range(world))
        prev = pg.recv[(pg.i - 2) % pg.depth]
        ok_pipe = ok_pipe and all(bool((prev[4 * r:4 * r + 4] == 10 * 3 + r + 1).all()) for r in range(world))
        pg4 = PipelinedFrameGatherer(dist, world, (4, 2, 2, 3), torch.uint8, torch.device("cpu"), depth=4)   # bench.py --pipeline 4
        for step in range(9):
            pg4.slot().fill_(step + rank + 1)
            pg4.submit()
        pg4.drain()
        for back in range(4):                                     # the four newest batches are all still held, each in its own pair
            buf = pg4.recv[(pg4.i - 1 - back) % pg4.depth]
            ok_pipe = ok_pipe and all(bool((buf[4 * r:4 * r + 4] == (8 - back) + r + 1).all()) for r in range(world))
        q.put((rank, ok_ragged, ok_equal and ok_pipe))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_preserves_frame_order():
    world, n_total = 2, 7          # ragged: rank 0 owns 4 frames, rank 1 owns 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]
