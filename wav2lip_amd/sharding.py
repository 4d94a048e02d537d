"""Multi-GPU data parallelism of the inference path (new design; the reference has none).

Frames are independent given the weights, so the frame/mel-window list is cut into contiguous per-rank chunks
(rank r owns [r*ceil(n/W), min(n, (r+1)*ceil(n/W)))), every rank runs datagen + generator locally with replicated
weights, and the ONE exchange step is an all-gather of the generated uint8 NHWC crops (27 648 B per frame — 4x
smaller than fp32) in frame order for the single writer.  One process per GPU, torch.distributed backend "nccl"
(= RCCL over xGMI) on the device path, "gloo" in the CPU tests of the partition/ordering logic.
"""
import torch


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
        work.append((dist.all_reduce(flat, async_op=True), flat, bucket))
    for handle, flat, bucket in work:
        handle.wait()
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
        work = self.dist.all_reduce(flat, async_op=True) if self.world > 1 else None
        self._inflight.append((work, flat, [(k, g.shape, g.numel()) for k, g in self._open]))
        self._open, self._open_bytes = [], 0

    def finalize(self):
        """-> {key: averaged gradient}"""
        self._launch()
        out = {}
        for work, flat, items in self._inflight:
            if work is not None:
                work.wait()
                flat.div_(self.world)
            off = 0
            for key, shape, n in items:
                out[key] = flat[off:off + n].view(shape)
                off += n
        self._inflight = []
        return out
