"""Multi-GPU data parallelism of the inference path (new design; the reference has none).

Frames are independent given the weights, so the frame/mel-window list is cut into contiguous per-rank chunks
(rank r owns [r*ceil(n/W), min(n, (r+1)*ceil(n/W)))), every rank runs datagen + generator locally with replicated
weights, and the ONE exchange step is an all-gather of the generated uint8 NHWC crops (27 648 B per frame — 4x
smaller than fp32) in frame order for the single writer.  One process per GPU, torch.distributed backend "nccl"
(= RCCL over xGMI) on the device path, "gloo" in the CPU tests of the partition/ordering logic.

Both product entry points use it: `trainer.main_*` (init_from_env -> broadcast_state -> GradReducer.attach) and
`inference.lipsync` / `inference.main` (shard_range over the mel chunks -> local runner -> gather_frames_in_order, one writer).
"""
import datetime
import os

import torch


class Ranks:
    """what a process knows about its place in the job: `dist` is torch.distributed (None in a single-process run), `device`
    the HIP device of this rank (cuda:LOCAL_RANK; the CPU under the gloo tests)"""

    def __init__(self, dist, rank, world, local_rank, device, owned=False):
        self.dist, self.rank, self.world, self.local_rank, self.device = dist, rank, world, local_rank, device
        self.owned = owned        # this object created the process group (close() then destroys it)

    @property
    def writer(self):
        return self.rank == 0

    def close(self):
        if self.owned and self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
            self.owned = False


def init_from_env(backend="nccl", timeout_s=1800.0, env=None):
    """The launch contract of `python -m torch.distributed.run --nproc-per-node N -m wav2lip_amd.trainer ...` (and of
    wav2lip_amd.inference): WORLD_SIZE / RANK / LOCAL_RANK / MASTER_ADDR / MASTER_PORT come from the environment, every rank
    binds cuda:LOCAL_RANK BEFORE it creates the process group (RCCL picks its device from the current one) and joins it with
    backend "nccl" (= RCCL over xGMI).  Without WORLD_SIZE (or WORLD_SIZE=1) nothing is initialised and `dist` is None: the
    single-process behaviour of the reference's scripts.  `backend="gloo"` is the CPU test path (no device is bound)."""
    env = os.environ if env is None else env
    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    local = int(env.get("LOCAL_RANK", str(rank)))
    on_gpu = backend != "gloo" or torch.cuda.is_available()
    if on_gpu:
        if not torch.cuda.is_available():
            raise RuntimeError("wav2lip_amd: no HIP device (this engine has no CPU path)")
        if local >= torch.cuda.device_count():
            raise RuntimeError("wav2lip_amd: LOCAL_RANK %d but only %d HIP device(s) are visible" % (local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world <= 1:
        return Ranks(None, 0, 1, local, device)
    if not (0 <= rank < world):
        raise RuntimeError("wav2lip_amd: RANK %d outside WORLD_SIZE %d" % (rank, world))
    import torch.distributed as dist
    owned = not dist.is_initialized()
    if owned:
        kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
        if backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return Ranks(dist, rank, world, local, device, owned)


def broadcast_state(dist, *modules, src=0, bucket_bytes=64 << 20):
    """Every rank starts from rank `src`'s parameters AND buffers (BatchNorm running statistics, num_batches_tracked): each rank's
    constructor draws its own random initialisation, and averaged gradients applied to different weights are not data
    parallelism.  Called once after the checkpoints are loaded (what DistributedDataParallel's constructor does).  Tensors of
    one dtype travel flattened in ~64 MB buckets - a handful of large broadcasts instead of one per tensor - and are copied back
    IN PLACE, which bumps their version counters so that the packed device weights are rebuilt (engine.param_version)."""
    if dist is None or dist.get_world_size() == 1:
        return 0
    seen, by_dtype = set(), {}
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            if id(t) not in seen:
                seen.add(id(t))
                by_dtype.setdefault(t.dtype, []).append(t)
    n_coll = 0
    for dtype in sorted(by_dtype, key=str):          # the same order on every rank
        bucket, size = [], 0
        items = by_dtype[dtype]
        for i, t in enumerate(items):
            bucket.append(t)
            size += t.numel() * t.element_size()
            if size >= bucket_bytes or i == len(items) - 1:
                with torch.no_grad():
                    flat = torch.cat([b.detach().reshape(-1) for b in bucket])
                    if _host_staged(dist, flat):
                        wire = flat.cpu()
                        dist.broadcast(wire, src=src)
                        flat.copy_(wire)
                    else:
                        dist.broadcast(flat, src=src)
                    off = 0
                    for b in bucket:
                        b.copy_(flat[off:off + b.numel()].view_as(b))
                        off += b.numel()
                n_coll += 1
                bucket, size = [], 0
    return n_coll


def shard_range(n_items, rank, world):
    """contiguous chunk of rank `rank`: (begin, end)"""
    per = -(-n_items // world)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def shard_counts(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


class FrameGatherer:
    """all-gather of equal-sized uint8 frame batches into one preallocated [world*B, H, W, 3] buffer"""

    def __init__(self, dist, world, device):
        self.dist, self.world, self.device = dist, world, device
        self._buf = None

    def all_gather(self, frames_u8):
        shape = (self.world * frames_u8.shape[0],) + tuple(frames_u8.shape[1:])
        if self._buf is None or tuple(self._buf.shape) != shape:
            self._buf = torch.empty(shape, dtype=frames_u8.dtype, device=frames_u8.device)
        self.dist.all_gather_into_tensor(self._buf, frames_u8.contiguous())
        return self._buf


class PipelinedFrameGatherer:
    """The same exchange, overlapped with the next batch's compute: `depth` (send, receive) buffer pairs are used in
    rotation and every collective is issued asynchronously (RCCL runs it on its own stream), so the all-gather of batch i
    travels over xGMI while the generator works on batch i+1.  `slot()` hands out the send buffer to write the frames into
    (after making the compute stream wait for the collective that last used it); `submit()` launches the all-gather;
    `drain()` waits for everything in flight and returns the newest receive buffer."""

    def __init__(self, dist, world, shape, dtype, device, depth=2):
        self.dist, self.world, self.depth = dist, world, depth
        self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(depth)]
        self.recv = [torch.empty((world * shape[0],) + tuple(shape[1:]), dtype=dtype, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.i = 0

    def slot(self):
        k = self.i % self.depth
        if self.work[k] is not None:
            self.work[k].wait()        # stream-side wait on the device backends; the host does not block
            self.work[k] = None
        return self.send[k]

    def submit(self):
        k = self.i % self.depth
        self.work[k] = self.dist.all_gather_into_tensor(self.recv[k], self.send[k], async_op=True)
        self.i += 1
        return self.recv[k]

    def drain(self):
        for k in range(self.depth):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
        return self.recv[(self.i - 1) % self.depth] if self.i else None


def gather_frames_in_order(dist, local_frames, n_total, rank, world):
    """Ragged variant for a real clip: rank r holds the frames of shard_range(n_total, r, world); returns the
    [n_total, ...] tensor in frame order on every rank (chunks are padded to the largest one for the collective)."""
    counts = shard_counts(n_total, world)
    per = max(counts)
    pad = torch.zeros((per,) + tuple(local_frames.shape[1:]), dtype=local_frames.dtype, device=local_frames.device)
    pad[:local_frames.shape[0]] = local_frames
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def _host_staged(dist, t):
    """gloo moves host memory: under it (the CPU tests, and the two-ranks-on-one-GPU test) device tensors are staged through
    the host; under nccl (= RCCL) the collective reads and writes HBM directly over xGMI"""
    return t.is_cuda and dist.get_backend() == "gloo"


def gather_shards_to_writer(dist, local_frames, n_total, rank, world, device=None, chunk=256, writer=0):
    """The exchange step of sharded inference for a clip of any length: rank r holds the uint8 frames of
    shard_range(n_total, r, world) (a numpy array / list of arrays on the host, or a device tensor); the shards travel in rounds of
    at most `chunk` frames per rank - one fixed-shape all-gather per round, so device memory is bounded by world x chunk frames
    whatever the clip length, and every rank issues the same number of collectives even when the last shard is short or empty.
    Returns the n_total frames in frame order as one numpy array on the writer rank, None on the others."""
    import numpy as np
    counts = shard_counts(n_total, world)
    per = max(counts)
    mine = counts[rank]
    if mine == 0:
        local = None
    elif isinstance(local_frames, torch.Tensor):
        local = local_frames
    else:
        local = torch.from_numpy(np.ascontiguousarray(np.asarray(local_frames)))
    if (0 if local is None else local.shape[0]) != mine or (local is not None and (local.dim() != 4 or local.dtype != torch.uint8)):
        raise ValueError("rank %d must hold the %d uint8 [n,H,W,C] frames of its shard of %d frames over %d ranks"
                         % (rank, mine, n_total, world))
    if n_total == 0:
        return np.zeros((0, 0, 0, 3), dtype=np.uint8) if rank == writer else None
    dev = device if device is not None else (local.device if local is not None else torch.device("cpu"))
    staged = torch.device(dev).type == "cuda" and dist.get_backend() == "gloo"
    cdev = torch.device("cpu") if staged else torch.device(dev)
    # the frame shape comes from rank 0, whose shard is never empty when n_total > 0 (an empty shard does not know it)
    shape = torch.tensor(list(local.shape[1:]) if rank == 0 else [0, 0, 0], dtype=torch.int64, device=cdev)
    dist.broadcast(shape, src=0)
    fshape = tuple(int(v) for v in shape.cpu())
    send = torch.zeros((chunk,) + fshape, dtype=torch.uint8, device=cdev)
    recv = torch.empty((world * chunk,) + fshape, dtype=torch.uint8, device=cdev)
    parts = [[] for _ in range(world)]
    for lo in range(0, per, chunk):
        n = max(0, min(chunk, mine - lo))
        if n:
            send[:n].copy_(local[lo:lo + n])
        dist.all_gather_into_tensor(recv, send)
        if rank == writer:
            host = recv.view((world, chunk) + fshape).cpu().numpy()
            for r in range(world):
                nr = max(0, min(chunk, counts[r] - lo))
                if nr:
                    parts[r].append(host[r, :nr].copy())
    if rank != writer:
        return None
    return np.concatenate([a for r in range(world) for a in parts[r]], axis=0)


def all_gather_batch(dist, t):
    """[B, ...] per rank -> [world*B, ...] in rank order on every rank: the optional frame all-gather of BASELINE
    configs[4] (hq_wav2lip_train step: the discriminator sees the global batch; train.hq_train_step(gather_frames=...)).
    One `all_gather_into_tensor` over RCCL/xGMI on the device path (B x 3 x 5 x 96 x 96 fp32 = 35 MB per rank at B = 64: one
    large collective, the size xGMI's per-link ring wants)."""
    world = dist.get_world_size()
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out


# ---------------------------------------------------------------- training: data-parallel gradient averaging
def grad_buckets(params, bucket_bytes=32 << 20):
    """cut the parameter list (reverse order = the order backward produces gradients) into buckets of about
    `bucket_bytes`: lists of parameters.  Pure host logic."""
    buckets, cur, size = [], [], 0
    for p in reversed(list(params)):
        n = p.numel() * 4
        if cur and size + n > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += n
    if cur:
        buckets.append(cur)
    return buckets


def allreduce_gradients(dist, params, bucket_bytes=32 << 20):
    """Average `.grad` over all ranks (the DDP step of BASELINE configs 4/5; the reference used single-process
    nn.DataParallel, hq_wav2lip_train.py has none).  Gradients are flattened into ~32 MB buckets — large enough to run
    the xGMI ring at its per-link rate, small enough to pipeline — and every bucket is all-reduced asynchronously, so
    bucket k+1's flatten overlaps bucket k's collective; results are averaged and scattered back in place.
    Parameters without a gradient on this rank contribute zeros (every rank must issue the same collectives)."""
    world = dist.get_world_size()
    if world == 1:
        return
    params = [p for p in params]
    work = []
    for bucket in grad_buckets(params, bucket_bytes):
        for p in bucket:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        wire = flat.cpu() if _host_staged(dist, flat) else flat
        work.append((dist.all_reduce(wire, async_op=True), flat, bucket, wire))
    for handle, flat, bucket, wire in work:
        handle.wait()
        if wire is not flat:
            flat.copy_(wire)
        flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p))
            off += n


class GradReducer:
    """Gradient averaging OVERLAPPED with the backward pass.  A network's backward is one autograd node that walks its
    blocks last-to-first (wav2lip_amd/autograd.py); after every block it hands the fresh parameter gradients to
    `on_grads`.  They are appended to the open bucket; when the bucket reaches `bucket_bytes` it is flattened and its
    all-reduce is issued asynchronously, so the collective of the late layers' gradients runs over xGMI while the HIP
    kernels of the earlier layers' backward are still executing.  `finalize` flushes the last bucket, waits, divides by the
    world size and returns every gradient as a view into its bucket (no copy back).  Every rank must run the same model:
    the block order, hence the bucket composition, is then identical everywhere.

        reducer = GradReducer(torch.distributed)          # once
        reducer.attach(model)                              # backward() of `model` now returns averaged gradients
    """

    def __init__(self, dist, bucket_bytes=32 << 20):
        self.dist = dist
        self.world = dist.get_world_size()
        self.bucket_bytes = bucket_bytes
        self._open, self._open_bytes, self._inflight = [], 0, []

    def attach(self, *modules):
        for m in modules:
            m._train_graphs.reducer = self
        return self

    @staticmethod
    def detach(*modules):
        for m in modules:
            m._train_graphs.reducer = None

    def on_grads(self, grads):
        """grads: {key: tensor} produced by one block"""
        for key, g in grads.items():
            self._open.append((key, g))
            self._open_bytes += g.numel() * g.element_size()
        if self._open_bytes >= self.bucket_bytes:
            self._launch()

    def _launch(self):
        if not self._open:
            return
        flat = torch.cat([g.reshape(-1) for _, g in self._open])
        wire = flat.cpu() if (self.world > 1 and _host_staged(self.dist, flat)) else flat
        work = self.dist.all_reduce(wire, async_op=True) if self.world > 1 else None
        self._inflight.append((work, flat, [(k, g.shape, g.numel()) for k, g in self._open], wire))
        self._open, self._open_bytes = [], 0

    def finalize(self):
        """-> {key: averaged gradient}"""
        self._launch()
        out = {}
        for work, flat, items, wire in self._inflight:
            if work is not None:
                work.wait()
                if wire is not flat:
                    flat.copy_(wire)
                flat.div_(self.world)
            off = 0
            for key, shape, n in items:
                out[key] = flat[off:off + n].view(shape)
                off += n
        self._inflight = []
        return out
