"""`s3fd` (face_detection/detection/sfd/net_s3fd.py:22-129) with the reference's attribute names and state-dict keys, run as
fused HIP launches: every convolution is a `w2l_conv` layer (bias + ReLU folded into the launch; 3x3 / stride 1 layers go
through the Winograd kernel), max-pools, L2Norm and the box decode are the glue kernels of csrc/detect.hip.
Activations are NHWC fp32; the detection heads write (conf, loc) maps that `w2l_s3fd_decode` turns into dense
(x1, y1, x2, y2, score) tables, one per pyramid level."""
import ctypes as C

import numpy as np
import torch
from torch import nn

from .. import engine
from .._lib import ACT_NONE, ACT_RELU, check, current_stream, load, ptr
from ..engine import Act

# (name, cin, cout, kernel, stride, padding), net_s3fd.py:25-48
BACKBONE = [("conv1_1", 3, 64, 3, 1, 1), ("conv1_2", 64, 64, 3, 1, 1), "pool",
            ("conv2_1", 64, 128, 3, 1, 1), ("conv2_2", 128, 128, 3, 1, 1), "pool",
            ("conv3_1", 128, 256, 3, 1, 1), ("conv3_2", 256, 256, 3, 1, 1), ("conv3_3", 256, 256, 3, 1, 1), "tap:conv3_3", "pool",
            ("conv4_1", 256, 512, 3, 1, 1), ("conv4_2", 512, 512, 3, 1, 1), ("conv4_3", 512, 512, 3, 1, 1), "tap:conv4_3", "pool",
            ("conv5_1", 512, 512, 3, 1, 1), ("conv5_2", 512, 512, 3, 1, 1), ("conv5_3", 512, 512, 3, 1, 1), "tap:conv5_3", "pool",
            ("fc6", 512, 1024, 3, 1, 3), ("fc7", 1024, 1024, 1, 1, 0), "tap:fc7",
            ("conv6_1", 1024, 256, 1, 1, 0), ("conv6_2", 256, 512, 3, 2, 1), "tap:conv6_2",
            ("conv7_1", 512, 128, 1, 1, 0), ("conv7_2", 128, 256, 3, 2, 1), "tap:conv7_2"]
# (feature, L2Norm module or None, conf channels), net_s3fd.py:50-66
HEADS = [("conv3_3", "conv3_3_norm", 4), ("conv4_3", "conv4_3_norm", 2), ("conv5_3", "conv5_3_norm", 2), ("fc7", None, 2),
         ("conv6_2", None, 2), ("conv7_2", None, 2)]


class L2Norm(nn.Module):
    """parameter container of net_s3fd.py:6-19 (`weight` initialised to `scale`)"""

    def __init__(self, n_channels, scale=1.0):
        super().__init__()
        self.n_channels, self.scale, self.eps = n_channels, scale, 1e-10
        self.weight = nn.Parameter(torch.full((n_channels,), float(scale)))


class _Graph:
    """buffers + launch list for one (batch, height, width)"""

    def __init__(self, model, B, H, W, device):
        self.lib = load()
        self.B, self.H, self.W = B, H, W
        self.x_in = engine.new_buf(B, H, W, 4, device, zero=True)
        self.ops = []          # ("convs", Plan) | ("pool", src, dst) | ("l2norm", src, weight, dst)
        self.keep = []
        plan = None

        def flush():
            nonlocal plan
            if plan is not None:
                self.ops.append(("convs", plan))
                plan = None

        def conv(name, src, act, cout_buf=None):
            nonlocal plan
            layer = model._layer(name, act)
            ho, wo = layer.out_hw(src.H, src.W)
            ct = (layer.cout + 3) // 4 * 4
            dst = Act(engine.new_buf(B, ho, wo, ct, device, zero=(ct != layer.cout)), 0, layer.cout)
            if plan is None:
                plan = engine.Plan()
            plan.add(name, layer, src, dst)
            return dst

        x = Act(self.x_in, 0, 4)
        taps = {}
        for item in BACKBONE:
            if item == "pool":
                flush()
                if x.H < 2 or x.W < 2:
                    raise RuntimeError("image too small for the S3FD pyramid")
                dst = Act(engine.new_buf(B, x.H // 2, x.W // 2, x.C, device), 0, x.C)
                self.ops.append(("pool", x, dst))
                x = dst
            elif isinstance(item, str):
                taps[item[4:]] = x
            else:
                x = conv(item[0], x, ACT_RELU)
        flush()
        self.levels = []       # (conf Act, loc Act, ncls, stride)
        for i, (feat, norm, ncls) in enumerate(HEADS):
            f = taps[feat]
            src = f
            prefix = feat
            if norm is not None:
                flush()
                nb = Act(engine.new_buf(B, f.H, f.W, f.C, device), 0, f.C)
                self.ops.append(("l2norm", f, getattr(model, norm).weight, nb))
                src = nb
                prefix = norm
            conf = conv(prefix + "_mbox_conf", src, ACT_NONE)
            loc = conv(prefix + "_mbox_loc", src, ACT_NONE)
            self.levels.append((conf, loc, ncls, 2 ** (i + 2)))
        flush()
        self.dense = [torch.empty((B, c.H * c.W, 5), device=device, dtype=torch.float32) for c, _, _, _ in self.levels]

    def run(self):
        s = current_stream()
        lib = self.lib
        for op in self.ops:
            if op[0] == "convs":
                op[1].run()
            elif op[0] == "pool":
                _, a, d = op
                check(lib.w2l_maxpool2x2(s, a.N, a.H, a.W, a.C, a.ptr, a.cs, d.ptr, d.cs), "maxpool2x2")
            else:
                _, a, w, d = op
                check(lib.w2l_l2norm_scale(s, a.N * a.H * a.W, a.C, a.ptr, a.cs, ptr(w.detach()), d.ptr, d.cs), "l2norm_scale")

    def decode(self):
        s = current_stream()
        for (conf, loc, ncls, stride), out in zip(self.levels, self.dense):
            check(self.lib.w2l_s3fd_decode(s, self.B, conf.H, conf.W, stride, conf.ptr, conf.cs, ncls, loc.ptr, loc.cs,
                                           ptr(out)), "s3fd_decode")
        return self.dense


class s3fd(nn.Module):
    def __init__(self):
        super().__init__()
        for item in BACKBONE:
            if not isinstance(item, str):
                name, cin, cout, k, st, p = item
                setattr(self, name, nn.Conv2d(cin, cout, kernel_size=k, stride=st, padding=p))
        self.conv3_3_norm = L2Norm(256, scale=10)
        self.conv4_3_norm = L2Norm(512, scale=8)
        self.conv5_3_norm = L2Norm(512, scale=5)
        cins = {"conv3_3": 256, "conv4_3": 512, "conv5_3": 512, "fc7": 1024, "conv6_2": 512, "conv7_2": 256}
        for feat, norm, ncls in HEADS:
            prefix = norm or feat
            setattr(self, prefix + "_mbox_conf", nn.Conv2d(cins[feat], ncls, kernel_size=3, stride=1, padding=1))
            setattr(self, prefix + "_mbox_loc", nn.Conv2d(cins[feat], 4, kernel_size=3, stride=1, padding=1))
        self._layers = {}
        self._graphs = {}
        self._version = None

    def _layer(self, name, act):
        key = (name, act)
        if key not in self._layers:
            self._layers[key] = engine.FusedConv(getattr(self, name), None, act)
        return self._layers[key]

    def _graph(self, B, H, W, device):
        ver = engine.param_version(self)
        if ver != self._version:
            self._layers, self._graphs, self._version = {}, {}, ver
        key = (B, H, W, str(device))
        g = self._graphs.get(key)
        if g is None:
            g = _Graph(self, B, H, W, torch.device(device))
            self._graphs = {key: g}        # one live geometry: VGG activations of a video frame batch are large
        return g

    def forward(self, x):
        """x: float NCHW (already mean-subtracted RGB) -> [cls1, reg1, ..., cls6, reg6] NCHW, cls1 with the background
        max-out applied (net_s3fd.py:68-129)"""
        engine.require_cuda(x, "input")
        x = x.contiguous().float()
        B, Cn, H, W = x.shape
        g = self._graph(B, H, W, x.device)
        lib = load()
        s = current_stream()
        check(lib.w2l_nchw_to_nhwc(s, B, Cn, H, W, ptr(x), ptr(g.x_in), 4, 4), "nchw_to_nhwc")
        g.run()
        outs = []
        for conf, loc, ncls, _ in g.levels:
            for a in (conf, loc):
                y = torch.empty((B, a.C, a.H, a.W), device=x.device, dtype=torch.float32)
                check(lib.w2l_nhwc_to_nchw(s, B, a.C, a.H, a.W, a.ptr, a.cs, ptr(y)), "nhwc_to_nchw")
                outs.append(y)
        chunk = torch.chunk(outs[0], 4, 1)          # max-out background label (3 of the 4 conf channels of level 1)
        outs[0] = torch.cat([torch.max(torch.max(chunk[0], chunk[1]), chunk[2]), chunk[3]], dim=1)
        return outs

    def dense_boxes(self, images_bgr_u8):
        """images: torch uint8 [B,H,W,3] BGR on the device -> per level torch float32 [B, FH*FW, 5] (x1,y1,x2,y2,score):
        api.py:62 (BGR->RGB) + detect.py:57-84 for every position"""
        if images_bgr_u8.dtype != torch.uint8 or images_bgr_u8.dim() != 4 or images_bgr_u8.shape[3] != 3:
            raise RuntimeError("dense_boxes: images must be uint8 [B,H,W,3]")
        engine.require_cuda(images_bgr_u8, "images")
        img = images_bgr_u8.contiguous()
        B, H, W = img.shape[:3]
        g = self._graph(B, H, W, img.device)
        check(load().w2l_s3fd_pack(current_stream(), B * H * W, ptr(img), ptr(g.x_in), 4), "s3fd_pack")
        g.run()
        return g.decode()


def nms_batch(table, gate, thresh):
    """sfd_detector.py:39-45 + bbox.py:44-64 on the device for a batch: table = torch float32 [B, P, 5] (x1, y1, x2, y2, score)
    on the HIP device; per image the rows with score > gate compete (`w2l_s3fd_nms`).  Returns (keep int32 [B, P], counts int32
    [B]): keep[b, :counts[b]] are the kept row indices of image b, best score first."""
    engine.require_cuda(table, "box table")
    if table.dtype != torch.float32 or table.dim() != 3 or table.shape[2] != 5:
        raise RuntimeError("nms_batch: table must be float32 [B, P, 5]")
    table = table.contiguous()
    B, P = table.shape[:2]
    keep = torch.empty((B, P), device=table.device, dtype=torch.int32)
    counts = torch.empty((B,), device=table.device, dtype=torch.int32)
    scratch = torch.empty((B * P * 12 + 8,), device=table.device, dtype=torch.uint8)
    check(load().w2l_s3fd_nms(current_stream(), B, P, ptr(table), float(gate), float(thresh), ptr(keep), ptr(counts),
                              ptr(scratch), scratch.numel()), "s3fd_nms")
    over = (counts < 0).nonzero().flatten().tolist()
    for b in over:
        # more rows above the gate than the device pass holds (> 262 144: a multi-megapixel frame with a permissive gate): this
        # image's survivors go through the same greedy pass on the host - any table size works, as in the reference
        rows = (table[b, :, 4] > gate).nonzero().flatten()
        kept = _nms_host(table[b].index_select(0, rows).cpu().numpy(), thresh)
        idx = rows[torch.as_tensor(kept, dtype=torch.long, device=rows.device)].to(torch.int32)
        keep[b, :len(kept)] = idx
        counts[b] = len(kept)
    return keep, counts


def _nms_host(dets, thresh):
    """bbox.py:44-64 in its float32 operation order (numpy): kept row indices of `dets` [n, 5], best score first.  Only the
    overflow route of nms_batch comes here."""
    d = np.asarray(dets, dtype=np.float32)
    x1, y1, x2, y2, sc = (d[:, i] for i in range(5))
    one = np.float32(1)
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    order = np.lexsort((np.arange(len(d)), sc))[::-1]        # score descending, equal scores: the later row first (as the device pass)
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + one)
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + one)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        order = rest[ovr <= np.float32(thresh)]
    return keep


def nms(dets, thresh):
    """`nms(dets, thresh)` of bbox.py:44-64: dets [n, 5] (numpy or torch) -> list of kept row indices, best score first.
    Runs on the HIP device (every row competes: the gate is -inf)."""
    if 0 == len(dets):
        return []
    t = torch.as_tensor(np.ascontiguousarray(dets) if isinstance(dets, np.ndarray) else dets, dtype=torch.float32)
    keep, counts = nms_batch(t.to("cuda").reshape(1, -1, 5), float("-inf"), thresh)
    return keep[0, :int(counts[0].item())].cpu().tolist()
