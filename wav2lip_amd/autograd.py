"""Training-mode execution of the mirrored modules: the torch autograd graph that `loss.backward()` walks in the
reference's training loops (wav2lip_train.py:220-231, color_syncnet_train.py:155-165, hq_wav2lip_train.py:221-256)
restated as ONE autograd node per network whose forward and backward are sequences of HIP launches.

Per block (models/conv.py:5-44), forward in train mode:
    z = conv(x) + bias                      w2l_conv_forward (scale 1, shift bias, no activation)
    mean, rstd (+ running stats)            w2l_bn_train_stats
    y = relu(gamma*(z-mean)*rstd + beta (+x))   w2l_affine_act
and backward:
    g = dy*[y>0]; dgamma, dbeta, dz         w2l_bn_train_bwd        (g is also the residual branch's gradient)
    dbias = sum dz                          w2l_col_sum
    dW                                      w2l_conv_wgrad
    dx = conv_transpose(dz, W) (+ g)        w2l_conv_forward on the transposed geometry (same weight tensor)
Blocks in eval mode (the frozen SyncNet of wav2lip_train.py:187-190,196) run the BN-folded inference launch forward and
only propagate the data gradient.  nonorm blocks (the discriminator) and bare conv heads skip the BN steps.

Buffers are static per (network, batch, size): every activation keeps its own NHWC buffer (saved for backward), every
forward buffer has a gradient twin, and the generator's skip concats are channel slices of shared buffers in both.
A gradient region that already holds a contribution from another consumer is accumulated through the kernel's residual
input; nothing is ever zero-filled per step.
"""
import ctypes as C
import os

import torch

from . import _lib, bf16, engine
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ConvGeom, check, current_stream, ptr
from .bf16 import ActB, ConvB, round8
from .engine import Act


def _round4(c):
    return (c + 3) // 4 * 4


_CONST = {}


def const_vec(device, n, value):
    key = (str(device), n, float(value))
    t = _CONST.get(key)
    if t is None:
        t = torch.full((n,), float(value), device=device, dtype=torch.float32)
        _CONST[key] = t
    return t


class RawConv:
    """a `w2l_conv` handle built from explicit geometry and device tensors (weight in torch layout)"""

    def __init__(self, geom, weight, scale, shift, precision="f32"):
        self._lib = _lib.load()
        self.geom = geom
        h = C.c_void_p()
        check(self._lib.w2l_conv_create(C.byref(geom), ptr(weight), ptr(scale), ptr(shift), current_stream(),
                                        C.byref(h)), "conv_create")
        self.handle = h
        if precision != "f32":
            check(self._lib.w2l_conv_set_precision(h, {"bf16c": _lib.PREC_BF16}[precision]), "conv_set_precision")
        self.cin, self.cout = geom.cin, geom.cout
        self._plans = {}    # launch signature -> one-item w2l_plan carrying the autotuned (tile, split-K)

    def update(self, weight=None, scale=None, shift=None):
        check(self._lib.w2l_conv_update(self.handle, ptr(weight), ptr(scale), ptr(shift), current_stream()),
              "conv_update")

    def out_hw(self, H, W):
        ho, wo = C.c_int(), C.c_int()
        check(self._lib.w2l_conv_out_hw(C.byref(self.geom), H, W, C.byref(ho), C.byref(wo)), "conv_out_hw")
        return ho.value, wo.value

    def _plan(self, x, y, res):
        lib = self._lib
        p = C.c_void_p()
        check(lib.w2l_plan_create(C.byref(p)), "plan_create")
        check(lib.w2l_plan_add_conv(p, self.handle, x.N, x.H, x.W, x.ptr, x.cs, y.ptr, y.cs,
                                    res.ptr if res is not None else None, res.cs if res is not None else 0), "plan_add_conv")
        return p

    def run(self, x, y, res=None):
        """enqueue the layer; with autotuning on (default) the first launch of each (buffers, shape) signature times the
        tile / split-K candidates on the device (w2l_plan_autotune) and later launches replay the winner"""
        if not engine.AUTOTUNE:
            check(self._lib.w2l_conv_forward(self.handle, current_stream(), x.N, x.H, x.W, x.ptr, x.cs, y.ptr, y.cs,
                                             res.ptr if res is not None else None, res.cs if res is not None else 0),
                  "conv_forward")
            return
        lib = self._lib
        key = (x.ptr.value, x.N, x.H, x.W, x.cs, y.ptr.value, y.cs, res.ptr.value if res is not None else 0,
               res.cs if res is not None else 0)
        p = self._plans.get(key)
        if p is None:
            p = self._plan(x, y, res)
            if res is not None and res.buf.data_ptr() == y.buf.data_ptr():
                # accumulating launch (y += conv): repeated timing runs would corrupt the gradient, so tune a twin of the
                # launch on a scratch copy of the output and transfer the configuration
                tmp = torch.empty_like(y.buf)
                ty = Act(tmp, y.off, y.C)
                tr = Act(tmp, res.off, res.C)
                q = self._plan(x, ty, tr)
                check(lib.w2l_plan_autotune(q, current_stream(), 2), "plan_autotune")
                t, k = C.c_int(), C.c_int()
                check(lib.w2l_plan_get_config(q, 0, C.byref(t), C.byref(k)), "plan_get_config")
                check(lib.w2l_plan_set_config(p, 0, t.value, k.value), "plan_set_config")
                lib.w2l_plan_destroy(q)
                del tmp
            else:
                check(lib.w2l_plan_autotune(p, current_stream(), 2), "plan_autotune")
            self._plans[key] = p
        check(lib.w2l_plan_run(p, current_stream()), "plan_run")

    def __del__(self):
        try:
            for p in getattr(self, "_plans", {}).values():
                self._lib.w2l_plan_destroy(p)
            if getattr(self, "handle", None):
                self._lib.w2l_conv_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def describe(blk):
    """(conv, bn, act, transposed, residual) of a mirrored block or a PlainConv adapter"""
    if hasattr(blk, "conv_block"):
        conv = blk.conv_block[0]
        bn = blk.conv_block[1] if blk._norm else None
        return conv, bn, blk._act, blk._transposed, bool(blk.residual)
    return blk._conv, None, blk._act, False, False


def _ver(t):
    return None if t is None else (t.data_ptr(), t._version, engine.PARAM_EPOCH[0])


class Node:
    """one block of a TrainGraph: x slice -> y slice"""

    def __init__(self, graph, name, blk, x, y):
        self.graph, self.name, self.blk, self.x, self.y = graph, name, blk, x, y
        conv, bn, act, transposed, residual = describe(blk)
        self.conv, self.bn, self.act, self.transposed, self.residual = conv, bn, act, transposed, residual
        dev = graph.device
        self.lib = graph.lib
        kh, kw = engine._pair(conv.kernel_size)
        sh, sw = engine._pair(conv.stride)
        ph, pw = engine._pair(conv.padding)
        oph, opw = engine._pair(conv.output_padding) if transposed else (0, 0)
        cin, cout = conv.in_channels, conv.out_channels
        self.cin, self.cout = cin, cout
        self.cout_p = _round4(cout)
        if bn is None:
            self.kind = "plain"
        elif bn.training:
            self.kind = "bn"
            if bn.momentum is None or not bn.track_running_stats or not bn.affine:
                raise NotImplementedError("BatchNorm2d variants other than the reference's default are not on the hot path")
        else:
            self.kind = "bn_eval"
        fwd_act = ACT_NONE if self.kind == "bn" else act
        self.geom = ConvGeom(int(transposed), cin, cout, kh, kw, sh, sw, ph, pw, oph, opw, fwd_act)
        ones, zeros = const_vec(dev, cout, 1.0), const_vec(dev, cout, 0.0)
        self.fold_scale = torch.ones(cout, device=dev)
        self.fold_shift = torch.zeros(cout, device=dev)
        self.precision = engine.TRAIN_PRECISION[0]
        self.fwd = RawConv(self.geom, conv.weight.detach(), ones, zeros, self.precision)
        ho, wo = self.fwd.out_hw(x.H, x.W)
        if (y.H, y.W, y.N) != (ho, wo, x.N) or y.C != cout:
            raise RuntimeError("train graph %s: output slice %s does not match %s" %
                               (name, (y.N, y.H, y.W, y.C), (x.N, ho, wo, cout)))
        if x.C < cin or x.cs - x.off < _round4(cin):
            raise RuntimeError("train graph %s: input slice too narrow" % name)
        self.rows = y.N * y.H * y.W
        # data gradient = the same weight tensor read with the other interpretation (conv <-> transposed conv)
        if not transposed:
            dg = ConvGeom(1, cout, cin, kh, kw, sh, sw, ph, pw, (x.H + 2 * ph - kh) % sh, (x.W + 2 * pw - kw) % sw, ACT_NONE)
        else:
            dg = ConvGeom(0, cout, cin, kh, kw, sh, sw, ph, pw, 0, 0, ACT_NONE)
        self.dgrad_geom = dg
        self.dgrad = None          # built on first backward that needs it
        if self.kind == "bn":
            self.z = engine.new_buf(y.N, y.H, y.W, cout, dev)
            self.mean = torch.empty(cout, device=dev)
            self.rstd = torch.empty(cout, device=dev)
            self.scale = torch.empty(cout, device=dev)
            self.shift = torch.empty(cout, device=dev)
        self._seen = None
        self._own_dz = None

    # ---- parameters -> packed handles (weights change every optimiser step)
    def parameters(self):
        """the parameters this node can produce gradients for"""
        return (self.conv.weight, self.conv.bias) + ((self.bn.weight, self.bn.bias) if self.bn is not None else ())

    def _zero_bias_grad(self):
        """the bias of a conv in front of batch statistics has an exactly zero gradient: a slice of the backward pass's one zero
        buffer (TrainGraph.zero_slice)"""
        return self.graph.zero_slice(self.cout)

    def refresh(self):
        conv, bn = self.conv, self.bn
        seen = (_ver(conv.weight), _ver(conv.bias)) + (
            (_ver(bn.weight), _ver(bn.bias), _ver(bn.running_mean), _ver(bn.running_var)) if self.kind == "bn_eval" else ())
        if seen == self._seen:
            return
        w = conv.weight.detach()
        bias = conv.bias.detach() if conv.bias is not None else const_vec(self.graph.device, self.cout, 0.0)
        if self.kind == "bn_eval":
            check(self.lib.w2l_bn_fold(current_stream(), self.cout, ptr(bias), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                                       ptr(bn.running_mean), ptr(bn.running_var), float(bn.eps), ptr(self.fold_scale),
                                       ptr(self.fold_shift)), "bn_fold")
            self.fwd.update(w, self.fold_scale, self.fold_shift)
        else:
            self.fwd.update(w, None, bias)
        if self.dgrad is not None:
            self.dgrad.update(w)
        self._seen = seen

    def forward(self):
        s = current_stream()
        x, y = self.x, self.y
        tick = self.graph.tick
        res = x if self.residual else None
        if self.kind != "bn":
            self.fwd.run(x, y, res)
            tick(self, "fwd.conv")
            return
        if self.rows <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" %
                             str([y.N, self.cout, y.H, y.W]))
        bn = self.bn
        z = Act(self.z, 0, self.cout)
        self.fwd.run(x, z)
        tick(self, "fwd.conv")
        check(self.lib.w2l_bn_train_stats(s, self.rows, self.cout, z.ptr, z.cs, ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                                          float(bn.eps), float(bn.momentum), ptr(bn.running_mean), ptr(bn.running_var),
                                          ptr(self.mean), ptr(self.rstd), ptr(self.scale), ptr(self.shift)), "bn_train_stats")
        tick(self, "fwd.bn_stats")
        check(self.lib.w2l_affine_act(s, self.rows, self.cout, z.ptr, z.cs, ptr(self.scale), ptr(self.shift),
                                      res.ptr if res is not None else None, res.cs if res is not None else 0, self.act,
                                      y.ptr, y.cs), "affine_act")
        tick(self, "fwd.bn_apply")
        self.graph._bn_counters.append(bn.num_batches_tracked)   # incremented together at the end of the forward

    def backward(self, gy, gx, accumulate, want):
        """gy: gradient slice of y (overwritten with the masked gradient when the block is residual); gx: gradient slice
        of x or None; accumulate: gx already holds another consumer's contribution; want: parameter -> bool.
        Returns {parameter: gradient tensor}."""
        s = current_stream()
        lib = self.lib
        x, y = self.x, self.y
        dev = self.graph.device
        Cp = self.cout_p
        lane = getattr(self, "lane", 0)
        wstream = self.graph.wgrad_stream_for_step()
        if wstream is not None:
            # weight gradients run on their own stream (below): this block's dz must then outlive the next block's backward,
            # so it gets a buffer of its own instead of a pooled one
            if self._own_dz is None:
                self._own_dz = torch.zeros((y.N, y.H, y.W, Cp), device=dev, dtype=torch.float32)
                self.graph.bytes += self._own_dz.numel() * 4
            dz_buf = self._own_dz
        else:
            dz_buf = self.graph.scratch(y.N, y.H, y.W, Cp, lane)
        dz = Act(dz_buf, 0, self.cout)
        grads = {}
        tick = self.graph.tick
        g_ptr = gy.ptr if self.residual else None
        if self.kind == "bn":
            bn = self.bn
            dgamma = torch.empty(self.cout, device=dev)
            dbeta = torch.empty(self.cout, device=dev)
            check(lib.w2l_bn_train_bwd(s, self.rows, self.cout, gy.ptr, gy.cs, y.ptr, y.cs, ptr(self.z), self.cout, self.act,
                                       ptr(self.mean), ptr(self.rstd), ptr(self.scale), ptr(dgamma), ptr(dbeta),
                                       dz.ptr, dz.cs, g_ptr, gy.cs), "bn_train_bwd")
            if want(bn.weight):
                grads[bn.weight.data_ptr()] = dgamma
            if want(bn.bias):
                grads[bn.bias.data_ptr()] = dbeta
        elif self.kind == "bn_eval":
            check(lib.w2l_act_bwd(s, self.rows, self.cout, gy.ptr, gy.cs, y.ptr, y.cs, self.act, ptr(self.fold_scale),
                                  dz.ptr, dz.cs, g_ptr, gy.cs), "act_bwd")
        else:
            check(lib.w2l_act_bwd(s, self.rows, Cp, gy.ptr, gy.cs, y.ptr, y.cs, self.act, None, dz.ptr, dz.cs,
                                  None, 0), "act_bwd")
        tick(self, "bwd.bn_act")
        if self.kind != "bn_eval":   # eval-mode blocks are frozen: data gradient only
            conv = self.conv
            if want(conv.weight):
                prec = _lib.PREC_BF16 if self.precision == "bf16c" else _lib.PREC_F32
                if wstream is not None:
                    # The weight gradient is off the critical path (nothing in this backward pass consumes it) and
                    # compute-bound, while the next things on the critical path — this block's data gradient and the previous
                    # block's BatchNorm backward — are partly HBM-bound: it goes to a side stream behind an event that marks
                    # dz complete, and the two kinds of work share the chip.
                    ready = torch.cuda.Event()
                    ready.record(torch.cuda.current_stream())
                    with torch.cuda.stream(wstream):
                        wstream.wait_event(ready)
                        dw = torch.empty_like(conv.weight)
                        check(lib.w2l_conv_wgrad_prec(C.byref(self.geom), current_stream(), x.N, x.H, x.W, x.ptr, x.cs, dz.ptr,
                                                      dz.cs, ptr(dw), prec), "conv_wgrad")
                else:
                    dw = torch.empty_like(conv.weight)
                    check(lib.w2l_conv_wgrad_prec(C.byref(self.geom), s, x.N, x.H, x.W, x.ptr, x.cs, dz.ptr, dz.cs, ptr(dw),
                                                  prec), "conv_wgrad")
                grads[conv.weight.data_ptr()] = dw
                tick(self, "bwd.wgrad")
            if conv.bias is not None and want(conv.bias):
                if self.kind == "bn":
                    # a bias in front of a batch-statistics BatchNorm has gradient sum(dz) = 0 in exact arithmetic (the mean
                    # subtraction removes it); torch returns rounding noise here (~1e-9 of the weight gradient), we return 0
                    grads[conv.bias.data_ptr()] = self._zero_bias_grad()
                else:
                    db = torch.empty(Cp, device=dev)
                    check(lib.w2l_col_sum(s, self.rows, Cp, dz.ptr, dz.cs, ptr(db)), "col_sum")
                    grads[conv.bias.data_ptr()] = db[:self.cout]
                    tick(self, "bwd.bias")
        if gx is not None:
            if self.dgrad is None:
                dg = self.dgrad_geom
                self.dgrad = RawConv(dg, self.conv.weight.detach(), const_vec(dev, dg.cout, 1.0), const_vec(dev, dg.cout, 0.0),
                                     self.precision)
            dzin = Act(dz_buf, 0, Cp)
            if accumulate:
                self.dgrad.run(dzin, gx, gx)
                if self.residual:
                    check(lib.w2l_add_rows(s, x.N * x.H * x.W, self.cin, gx.ptr, gx.cs, gy.ptr, gy.cs, gx.ptr, gx.cs), "add_rows")
            else:
                self.dgrad.run(dzin, gx, Act(gy.buf, gy.off, self.cout) if self.residual else None)
            tick(self, "bwd.dgrad")
        if wstream is None:
            self.graph.release_scratch(dz_buf, lane)
        return grads


# W2L_BWD_SUMS_IN_DGRAD=0: every BatchNorm block reduces its own backward sums (the stand-alone pass over dy, z, y) - A/B switch
BWD_SUMS_IN_DGRAD = [os.environ.get("W2L_BWD_SUMS_IN_DGRAD", "1") != "0"]
# W2L_THIN_1X1=0: the 32 -> 3 output layer of a bf16 graph runs on the implicit GEMM like every other layer - A/B switch
THIN_1X1 = [os.environ.get("W2L_THIN_1X1", "1") != "0"]
# backward passes skip the nodes nobody wants a gradient from (TrainGraph.backward); W2L_BWD_PRUNE=0 is the A/B switch
BWD_PRUNE = [os.environ.get("W2L_BWD_PRUNE", "1") != "0"]
# a data-gradient launch that carries a ReLU block's BatchNorm-backward sums also stores the MASKED gradient (W2L_BNBWD_STORE_MASKED);
# W2L_STORE_MASKED_G=0 is the A/B switch
STORE_MASKED_G = [os.environ.get("W2L_STORE_MASKED_G", "1") != "0"]
# exactly-zero parameter gradients are slices of one fresh zero buffer per backward pass (TrainGraph.zero_slice); W2L_ZERO_POOL=0: A/B
ZERO_POOL = [os.environ.get("W2L_ZERO_POOL", "1") != "0"]
# activation blocks without BatchNorm: dz comes out of the data-gradient launch that completes their dy (W2L_ACT_BWD_IN_DGRAD=0: A/B)
ACT_BWD_IN_DGRAD = [os.environ.get("W2L_ACT_BWD_IN_DGRAD", "1") != "0"]


class NodeB:
    """one block of a bf16-STORAGE TrainGraph (engine.TRAIN_PRECISION "bf16"): the same block arithmetic as `Node` over NHWC
    bf16 buffers.  x, y, the pre-BatchNorm conv output z and the gradients dy / dz are bf16 in HBM; the contractions run on the
    bf16 matrix cores with fp32 accumulation (w2l_convb_forward for the forward and the data gradient, w2l_conv_wgrad_bf16 for
    the weight gradient, which comes out fp32); BatchNorm statistics, per-channel vectors and parameter gradients are fp32."""

    def __init__(self, graph, name, blk, x, y):
        self.graph, self.name, self.blk, self.x, self.y = graph, name, blk, x, y
        conv, bn, act, transposed, residual = describe(blk)
        self.conv, self.bn, self.act, self.transposed, self.residual = conv, bn, act, transposed, residual
        dev = graph.device
        self.lib = graph.lib
        kh, kw = engine._pair(conv.kernel_size)
        sh, sw = engine._pair(conv.stride)
        ph, pw = engine._pair(conv.padding)
        oph, opw = engine._pair(conv.output_padding) if transposed else (0, 0)
        cin, cout = conv.in_channels, conv.out_channels
        self.cin, self.cout = cin, cout
        self.cout_p = round8(cout)
        if bn is None:
            self.kind = "plain"
        elif bn.training:
            self.kind = "bn"
            if bn.momentum is None or not bn.track_running_stats or not bn.affine:
                raise NotImplementedError("BatchNorm2d variants other than the reference's default are not on the hot path")
        else:
            self.kind = "bn_eval"
        fwd_act = ACT_NONE if self.kind == "bn" else act
        self.geom = ConvGeom(int(transposed), cin, cout, kh, kw, sh, sw, ph, pw, oph, opw, fwd_act)
        self.precision = "bf16"
        # the generator's output layer (32 -> 3, 1x1, models/wav2lip.py:83-85): HBM-bound row kernels, not a 128x32 GEMM tile
        self.thin = (THIN_1X1[0] and self.kind == "plain" and not transposed and not residual and (kh, kw, sh, sw, ph, pw) == (1, 1, 1, 1, 0, 0)
                     and cin <= 32 and cout <= 4)
        self.fwd = None if self.thin else ConvB(self.geom, conv.weight)
        ho, wo = (x.H, x.W) if self.thin else self.fwd.out_hw(x.H, x.W)
        if (y.H, y.W, y.N) != (ho, wo, x.N) or y.C != cout:
            raise RuntimeError("train graph %s: output slice %s does not match %s" %
                               (name, (y.N, y.H, y.W, y.C), (x.N, ho, wo, cout)))
        if x.C < cin or x.cs - x.off < round8(cin) or y.cs - y.off < self.cout_p:
            raise RuntimeError("train graph %s: input / output slice too narrow" % name)
        self.rows = y.N * y.H * y.W
        if not transposed:
            dg = ConvGeom(1, cout, cin, kh, kw, sh, sw, ph, pw, (x.H + 2 * ph - kh) % sh, (x.W + 2 * pw - kw) % sw, ACT_NONE)
        else:
            dg = ConvGeom(0, cout, cin, kh, kw, sh, sw, ph, pw, 0, 0, ACT_NONE)
        self.dgrad_geom = dg
        self.dgrad = None
        Cp = self.cout_p
        # per-channel fp32 vectors carry Cp entries (pad entries zero): the elementwise kernels load them 8 at a time
        self.fold_scale = torch.zeros(Cp, device=dev)
        self.fold_shift = torch.zeros(Cp, device=dev)
        if self.kind == "bn":
            self.z = bf16.new_buf(y.N, y.H, y.W, cout, dev)
            graph.bytes += self.z.numel() * 2
            self.mean = torch.zeros(Cp, device=dev)
            self.rstd = torch.zeros(Cp, device=dev)
            self.scale = torch.zeros(Cp, device=dev)
            self.shift = torch.zeros(Cp, device=dev)
        self._seen = None
        self._own_dz = None
        self.sums_for = None      # the "bn" block whose dy this node's data gradient completes (TrainGraph._plan_bwd_fusion)
        self._bwd_sums = None     # (dgamma, dbeta) already reduced in the epilogue of the launch that wrote this block's dy

    def parameters(self):
        """the parameters this node can produce gradients for"""
        return (self.conv.weight, self.conv.bias) + ((self.bn.weight, self.bn.bias) if self.bn is not None else ())

    def _zero_bias_grad(self):
        """the bias of a conv in front of batch statistics has an exactly zero gradient: a slice of the backward pass's one zero
        buffer (TrainGraph.zero_slice)"""
        return self.graph.zero_slice(self.cout)

    def _state(self):
        conv, bn = self.conv, self.bn
        return (_ver(conv.weight), _ver(conv.bias)) + (
            (_ver(bn.weight), _ver(bn.bias), _ver(bn.running_mean), _ver(bn.running_var)) if self.kind == "bn_eval" else ())

    def stale_weights(self):
        """(layer handle, master weight) pairs whose bf16 slabs are older than the weight; TrainGraph re-packs the pairs of all
        its nodes in one launch (ConvB.update_many) and then calls refresh(packed=True)"""
        if self.thin or self._state() == self._seen:
            return []
        return [(self.fwd, self.conv.weight)] + ([(self.dgrad, self.conv.weight)] if self.dgrad is not None else [])

    def refresh(self, packed=False):
        conv, bn = self.conv, self.bn
        seen = self._state()
        if seen == self._seen:
            return
        if not packed and not self.thin:
            self.fwd.update(conv.weight)
            if self.dgrad is not None:
                self.dgrad.update(conv.weight)
        if self.kind == "bn_eval":
            bias = conv.bias.detach() if conv.bias is not None else None
            check(self.lib.w2l_bn_fold(current_stream(), self.cout, ptr(bias), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                                       ptr(bn.running_mean), ptr(bn.running_var), float(bn.eps), ptr(self.fold_scale),
                                       ptr(self.fold_shift)), "bn_fold")
        self._seen = seen

    def forward(self):
        s = current_stream()
        x, y = self.x, self.y
        tick = self.graph.tick
        res = x if self.residual else None
        bias = self.conv.bias.detach() if self.conv.bias is not None else None
        if self.thin:
            check(self.lib.w2l_thin1x1_forward_bf16(s, self.rows, self.cin, self.cout, x.ptr, x.cs, ptr(self.conv.weight.detach()),
                                                    ptr(bias), self.act, y.ptr, y.cs), "thin1x1_forward_bf16")
            tick(self, "fwd.conv")
            return
        if self.kind == "plain":
            self.fwd.run(x, y, res, None, bias)
            tick(self, "fwd.conv")
            return
        if self.kind == "bn_eval":
            self.fwd.run(x, y, res, self.fold_scale, self.fold_shift)
            tick(self, "fwd.conv")
            return
        if self.rows <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" %
                             str([y.N, self.cout, y.H, y.W]))
        bn = self.bn
        Cp = self.cout_p
        z = ActB(self.z, 0, self.cout)
        # the conv and the batch statistics of its output in one call: the sums come out of the conv epilogue
        self.fwd.run_bn(x, z, bias, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                        self.mean, self.rstd, self.scale, self.shift)
        tick(self, "fwd.conv")
        check(self.lib.w2l_affine_act_bf16(s, self.rows, Cp, z.ptr, z.cs, ptr(self.scale), ptr(self.shift),
                                           res.ptr if res is not None else None, res.cs if res is not None else 0, self.act,
                                           y.ptr, y.cs), "affine_act_bf16")
        tick(self, "fwd.bn_apply")
        self.graph._bn_counters.append(bn.num_batches_tracked)   # incremented together at the end of the forward

    def backward(self, gy, gx, accumulate, want):
        s = current_stream()
        lib = self.lib
        x, y = self.x, self.y
        dev = self.graph.device
        Cp = self.cout_p
        lane = getattr(self, "lane", 0)
        wstream = self.graph.wgrad_stream_for_step()
        if wstream is not None:
            if self._own_dz is None:
                self._own_dz = torch.zeros((y.N, y.H, y.W, Cp), device=dev, dtype=torch.bfloat16)
                self.graph.bytes += self._own_dz.numel() * 2
            dz_buf = self._own_dz
        else:
            dz_buf = self.graph.scratch(y.N, y.H, y.W, Cp, lane)
        dz = ActB(dz_buf, 0, self.cout)
        grads = {}
        tick = self.graph.tick
        g_ptr = gy.ptr if self.residual else None
        fused_db = None
        if self.kind == "bn":
            bn = self.bn
            dgamma = torch.empty(Cp, device=dev)
            dbeta = torch.empty(Cp, device=dev)
            # a ReLU block without residual: the mask is recomputed from z (the forward's own z*scale + shift), y is not read
            skip_y = (not self.residual) and self.act == ACT_RELU
            sums, self._bwd_sums = self._bwd_sums, None
            if sums is not None:
                # the launch that completed this block's dy (the data gradient of its consumer) left the two column sums behind:
                # only the elementwise half runs
                dgamma, dbeta, premasked = sums
                if premasked:
                    # ... and for a ReLU block it stored g = dy * mask in place of dy: no mask to rebuild (the block's output is
                    # not read), no g to write for the residual path (it is where the data gradient below expects it)
                    check(lib.w2l_bn_train_bwd_apply_bf16(s, self.rows, Cp, gy.ptr, gy.cs, None, 0, ptr(self.z), Cp, ACT_NONE,
                                                          ptr(self.mean), ptr(self.rstd), ptr(self.scale), ptr(self.shift),
                                                          ptr(dgamma), ptr(dbeta), dz.ptr, dz.cs, None, 0),
                          "bn_train_bwd_apply_bf16")
                else:
                    check(lib.w2l_bn_train_bwd_apply_bf16(s, self.rows, Cp, gy.ptr, gy.cs, None if skip_y else y.ptr, y.cs,
                                                          ptr(self.z), Cp, self.act, ptr(self.mean), ptr(self.rstd), ptr(self.scale),
                                                          ptr(self.shift), ptr(dgamma), ptr(dbeta), dz.ptr, dz.cs, g_ptr, gy.cs),
                          "bn_train_bwd_apply_bf16")
            else:
                check(lib.w2l_bn_train_bwd_bf16(s, self.rows, Cp, self.cout, gy.ptr, gy.cs, None if skip_y else y.ptr, y.cs,
                                                ptr(self.z), Cp, self.act, ptr(self.mean), ptr(self.rstd), ptr(self.scale),
                                                ptr(self.shift), ptr(dgamma), ptr(dbeta), dz.ptr, dz.cs, g_ptr, gy.cs),
                      "bn_train_bwd_bf16")
            if want(bn.weight):
                grads[bn.weight.data_ptr()] = dgamma[:self.cout]
            if want(bn.bias):
                grads[bn.bias.data_ptr()] = dbeta[:self.cout]
        elif self.kind == "bn_eval":
            check(lib.w2l_act_bwd_bf16(s, self.rows, Cp, gy.ptr, gy.cs, y.ptr, y.cs, self.act, ptr(self.fold_scale),
                                       dz.ptr, dz.cs, g_ptr, gy.cs), "act_bwd_bf16")
        else:
            sums, self._bwd_sums = self._bwd_sums, None
            if sums is not None and sums[2]:
                # the launch that completed this block's dy already multiplied it by act'(y): gy IS dz (no launch, no copy; the
                # gradient buffer of this block's output is not written again before the next backward pass)
                dz = ActB(gy.buf, gy.off, self.cout)
                fused_db = sums[0]
            else:
                check(lib.w2l_act_bwd_bf16(s, self.rows, Cp, gy.ptr, gy.cs, y.ptr, y.cs, self.act, None, dz.ptr, dz.cs,
                                           None, 0), "act_bwd_bf16")
        tick(self, "bwd.bn_act")
        if self.thin:
            conv = self.conv
            if want(conv.weight) or (conv.bias is not None and want(conv.bias)):
                dw = torch.empty_like(conv.weight)
                db = torch.empty(self.cout, device=dev) if conv.bias is not None else None
                check(lib.w2l_thin1x1_wgrad_bf16(s, self.rows, self.cin, self.cout, x.ptr, x.cs, dz.ptr, dz.cs, ptr(dw), ptr(db)),
                      "thin1x1_wgrad_bf16")
                if want(conv.weight):
                    grads[conv.weight.data_ptr()] = dw
                if db is not None and want(conv.bias):
                    grads[conv.bias.data_ptr()] = db
                tick(self, "bwd.wgrad")
            if gx is not None:
                check(lib.w2l_thin1x1_dgrad_bf16(s, self.rows, self.cin, self.cout, dz.ptr, dz.cs, ptr(conv.weight.detach()),
                                                 gx.ptr if accumulate else None, gx.cs if accumulate else 0, gx.ptr, gx.cs),
                      "thin1x1_dgrad_bf16")
                tick(self, "bwd.dgrad")
            if wstream is None:
                self.graph.release_scratch(dz_buf, lane)
            return grads
        if self.kind != "bn_eval":
            conv = self.conv
            if want(conv.weight):
                def wgrad(stream_ptr):
                    dw = torch.empty_like(conv.weight)
                    check(lib.w2l_conv_wgrad_bf16(C.byref(self.geom), stream_ptr, x.N, x.H, x.W, x.ptr, x.cs, dz.ptr, dz.cs,
                                                  ptr(dw)), "conv_wgrad_bf16")
                    return dw
                if wstream is not None:
                    ready = torch.cuda.Event()
                    ready.record(torch.cuda.current_stream())
                    with torch.cuda.stream(wstream):
                        wstream.wait_event(ready)
                        dw = wgrad(current_stream())
                else:
                    dw = wgrad(s)
                grads[conv.weight.data_ptr()] = dw
                tick(self, "bwd.wgrad")
            if conv.bias is not None and want(conv.bias):
                if self.kind == "bn":
                    grads[conv.bias.data_ptr()] = self._zero_bias_grad()   # exactly zero in front of batch statistics
                elif self.kind == "plain" and fused_db is not None:
                    grads[conv.bias.data_ptr()] = fused_db[:self.cout]      # the column sums came with dz
                else:
                    db = torch.empty(Cp, device=dev)
                    check(lib.w2l_col_sum_bf16(s, self.rows, Cp, dz.ptr, dz.cs, ptr(db)), "col_sum_bf16")
                    grads[conv.bias.data_ptr()] = db[:self.cout]
                    tick(self, "bwd.bias")
        if gx is not None:
            if self.dgrad is None:
                self.dgrad = ConvB(self.dgrad_geom, self.conv.weight)
            m = self.sums_for if BWD_SUMS_IN_DGRAD[0] else None
            res = gx if accumulate else (ActB(gy.buf, gy.off, self.cout) if self.residual else None)
            if m is not None and not (accumulate and self.residual):
                # this launch writes the final dy of block m (its first consumer in forward order = its last writer here): the
                # BatchNorm-backward column sums of m come out of the same epilogue
                if m.kind == "plain":
                    # ... with the column sums of that dz = the gradient of m's conv bias, when m has one that wants it
                    mb = m.conv.bias
                    db = torch.empty(m.cout_p, device=dev) if (mb is not None and want(mb)) else None
                    fused = self.dgrad.run_actbwd(dz, gx, res, m.y, m.act, db)
                    m._bwd_sums = (db, None, True) if fused else None
                else:
                    mCp = m.cout_p
                    dgamma, dbeta = torch.empty(mCp, device=dev), torch.empty(mCp, device=dev)
                    m_skip_y = (not m.residual) and m.act == ACT_RELU
                    premask = STORE_MASKED_G[0] and m.act == ACT_RELU
                    fused = self.dgrad.run_bnbwd(dz, gx, res, ActB(m.z, 0, m.cout), None if m_skip_y else m.y, m.act, m.mean, m.rstd,
                                                 m.scale, m.shift, dgamma, dbeta, store_masked=premask)
                    m._bwd_sums = (dgamma, dbeta, premask) if fused else None
            elif accumulate:
                self.dgrad.run(dz, gx, gx)
                if self.residual:
                    check(lib.w2l_add_rows_bf16(s, x.N * x.H * x.W, round8(self.cin), gx.ptr, gx.cs, gy.ptr, gy.cs, gx.ptr, gx.cs),
                          "add_rows_bf16")
            else:
                self.dgrad.run(dz, gx, res)
            tick(self, "bwd.dgrad")
        if wstream is None:
            self.graph.release_scratch(dz_buf, lane)
        return grads


class TrainGraph:
    """Static buffers + node list of one network for one (batch, height, width)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.lib = _lib.load()
        # bf16-storage graph: NHWC bf16 buffers (channel counts rounded up to 8), NodeB blocks, bf16 boundary conversions
        self.bf16 = engine.TRAIN_PRECISION[0] == "bf16"
        self.dtype = torch.bfloat16 if self.bf16 else torch.float32
        self.esize = 2 if self.bf16 else 4
        self.nodes = []
        self.inputs = []     # (Act, channels) filled from NCHW tensors
        self.outputs = []    # Act
        self.grad_of = {}    # id(forward buffer) -> gradient buffer
        self._bufs = []
        self._scratch = {}
        self.busy = False
        self.ticket = 0
        self.bytes = 0
        self._phase_ends = []
        self._side = (torch.cuda.Stream(device=self.device) if self.device.type == "cuda" and engine.TWO_STREAM_ENCODERS
                      else None)
        self._wstream = (torch.cuda.Stream(device=self.device) if self.device.type == "cuda" and engine.WGRAD_OVERLAP
                         else None)
        self._wstream_on = False
        self.events = None   # profiling: list of (node name, phase, cuda event) when enabled (W2L_TRAIN_PROFILE=1)
        self._bn_counters = []   # num_batches_tracked of the BatchNorms that ran in train mode during this forward

    def zero_slice(self, n):
        """n zeros for a gradient that is exactly zero (conv biases in front of batch statistics).  ONE zero-filled buffer per backward
        pass, handed out in slices: torch's AccumulateGrad keeps a gradient it is the only holder of (a fresh view is one) and CLONES
        a tensor somebody else still references - the per-node cached zero tensors of rounds 2-5 were cloned every step, ~50
        device-to-device copies behind the generator's backward pass."""
        if not ZERO_POOL[0]:           # A/B: a cached tensor per size (cloned by AccumulateGrad every step, as before)
            cache = self.__dict__.setdefault("_zero_cache", {})
            if n not in cache:
                cache[n] = torch.zeros(n, device=self.device)
            return cache[n]
        pool = getattr(self, "_zero_pool", None)
        if pool is None or self._zero_off + n > pool.numel():
            total = sum(nd.cout for nd in self.nodes if getattr(nd, "kind", None) == "bn")
            pool = self._zero_pool = torch.zeros(max(total, n), device=self.device)
            self._zero_off = 0
        out = pool[self._zero_off:self._zero_off + n]
        self._zero_off += n
        return out

    def wgrad_stream_for_step(self):
        """the side stream weight gradients go to during the current backward pass, or None (profiling timeline, gradient
        reducer attached, or the feature switched off)"""
        return self._wstream if self._wstream_on else None

    def tick(self, node, phase):
        if self.events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.events.append((node.name, phase, ev, node))

    def profile_begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.events = [("", "begin", ev, None)]

    def profile_mark(self, label):
        """re-anchor the clock (work between graph phases, e.g. the loss, is not attributed to a node)"""
        if self.events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.events.append(("", label, ev, None))

    def profile_end(self):
        """[(node name, phase, ms, MMAC)] since profile_begin(); conv/wgrad/dgrad phases carry the layer's nominal MACs"""
        torch.cuda.synchronize()
        evs, self.events = self.events, None
        out = []
        for (_, _, e0, _), (name, phase, e1, node) in zip(evs, evs[1:]):
            if node is None:
                continue
            macs = 0
            if phase in ("fwd.conv", "bwd.wgrad", "bwd.dgrad"):
                lib = self.lib
                macs = int(lib.w2l_conv_macs(C.byref(node.geom), node.x.N, node.x.H, node.x.W))
            out.append((name, phase, e0.elapsed_time(e1), macs))
        return out

    def rnd(self, c):
        """channel count rounded to the layout's granule: 4 floats or 8 bf16 = 16 bytes"""
        return round8(c) if self.bf16 else _round4(c)

    def act(self, buf, off, C_):
        return ActB(buf, off, C_) if self.bf16 else Act(buf, off, C_)

    def buffer(self, N, H, W, Cn):
        Ct = self.rnd(Cn)
        b = torch.zeros((N, H, W, Ct), device=self.device, dtype=self.dtype)
        self._bufs.append(b)
        self.grad_of[id(b)] = None   # allocated lazily: inference-only use of a graph never pays for it
        self.bytes += b.numel() * self.esize
        return b

    def grad_act(self, act, C_=None):
        g = self.grad_of[id(act.buf)]
        if g is None:
            g = torch.zeros_like(act.buf)
            self.grad_of[id(act.buf)] = g
            self.bytes += g.numel() * self.esize
        return self.act(g, act.off, act.C if C_ is None else C_)

    def scratch(self, N, H, W, Cp, lane=0):
        key = (N, H, W, Cp, lane)     # one pool per lane: lanes run on different streams
        lst = self._scratch.setdefault(key, [])
        if lst:
            return lst.pop()
        self.bytes += self.esize * N * H * W * Cp
        return torch.zeros(key[:4], device=self.device, dtype=self.dtype)

    def release_scratch(self, buf, lane=0):
        self._scratch[tuple(buf.shape) + (lane,)].append(buf)

    def add(self, name, blk, x, y, lane=0):
        n = (NodeB if self.bf16 else Node)(self, name, blk, x, y)
        n.lane = lane
        self.nodes.append(n)
        return y

    def chain(self, name, blocks, x, final_dst=None, lane=0):
        """blocks applied in sequence; each output gets its own buffer (kept for backward) unless it is `final_dst`"""
        for j, blk in enumerate(blocks):
            conv = describe(blk)[0]
            probe = RawConvShape(conv, describe(blk)[3])
            ho, wo = probe.out_hw(x.H, x.W)
            if j == len(blocks) - 1 and final_dst is not None:
                y = final_dst
            else:
                y = self.act(self.buffer(x.N, ho, wo, conv.out_channels), 0, conv.out_channels)
            x = self.add("%s.%d" % (name, j), blk, x, y, lane)
        return x

    # ---- lanes: an independent branch (the audio encoder) is recorded with lane=1 and executed on a side stream,
    # concurrently with the lane-0 nodes of the same PHASE; phases are maximal runs of nodes between `barrier()` marks
    def barrier(self):
        """everything recorded so far completes (all lanes) before anything recorded later starts"""
        self._phase_ends.append(len(self.nodes))

    def _phases(self):
        ends = sorted(set(self._phase_ends + [len(self.nodes)]))
        lo = 0
        for hi in ends:
            if hi > lo:
                yield self.nodes[lo:hi]
            lo = hi

    def _run_lanes(self, nodes, fn):
        """call fn(node) for every node: lane 0 on the current stream, lane 1 on the side stream, joined at the end"""
        side_nodes = [n for n in nodes if n.lane == 1]
        if not side_nodes or self._side is None:
            for n in nodes:
                fn(n)
            return
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            for n in side_nodes:
                fn(n)
        for n in nodes:
            if n.lane != 1:
                fn(n)
        main.wait_stream(self._side)

    def _plan_bwd_fusion(self):
        """bf16 graphs: for every batch-statistics block m, the node whose data gradient COMPLETES m's dy - the first node in
        forward order (= the last in a backward pass) that reads m's output, provided it reads exactly m's slice (a reader of a
        wider concat slice, or of a part, writes other channels with the same launch: no fusion) on the same lane.  That node's
        data-gradient launch also reduces m's two BatchNorm-backward column sums in its epilogue (NodeB.backward)."""
        self._bwd_planned = True
        if not self.bf16:
            return
        for i, m in enumerate(self.nodes):
            kind = getattr(m, "kind", None)
            # batch-statistics blocks (sums + masked store) and, this round, activation blocks WITHOUT BatchNorm (the discriminator's
            # conv + LeakyReLU): the launch that completes their dy stores dz = dy * act'(y) directly (w2l_convb_forward_actbwd)
            plain_act = (kind == "plain" and ACT_BWD_IN_DGRAD[0] and getattr(m, "act", ACT_NONE) in (ACT_RELU, ACT_LEAKY)
                         and not getattr(m, "residual", False) and not getattr(m, "thin", False))
            if kind != "bn" and not plain_act:
                continue
            lo, hi = m.y.off, m.y.off + m.cout
            readers = [n for n in self.nodes[i + 1:] if n.x.buf is m.y.buf and n.x.off < hi and lo < n.x.off + n.cin]
            if not readers:
                continue
            n = readers[0]
            if n.x.off == lo and n.cin == m.cout and n.kind != "bn_eval" and n.lane == m.lane and (n.x.N, n.x.H, n.x.W) == (m.y.N, m.y.H, m.y.W):
                n.sums_for = m

    # ---- execution
    def forward(self, tensors):
        s = current_stream()
        self._bn_counters = []       # a forward that raised half way leaves its list behind: never count those layers twice
        for (act, cch), t in zip(self.inputs, tensors):
            engine.require_cuda(t, "input")
            t = t.detach().contiguous().float()
            to_nhwc = self.lib.w2l_nchw_to_nhwc_bf16 if self.bf16 else self.lib.w2l_nchw_to_nhwc
            check(to_nhwc(s, act.N, cch, act.H, act.W, ptr(t), act.ptr, act.cs, act.cs - act.off), "nchw_to_nhwc")
        self.profile_mark("inputs")

        if self.bf16:
            # one launch re-packs the bf16 weight slabs of every layer an optimiser step has touched
            stale = [n for n in self.nodes if isinstance(n, NodeB)]
            pairs = [p for n in stale for p in n.stale_weights()]
            if pairs:
                bf16.ConvB.update_many(pairs)
                for n in stale:
                    n.refresh(packed=True)
                self.profile_mark("repack")

        def fwd(n):
            n.refresh()
            n.forward()
        if self.events is not None:            # per-node profiling needs one timeline
            for n in self.nodes:
                fwd(n)
        else:
            for phase in self._phases():
                self._run_lanes(phase, fwd)
        if self._bn_counters:        # models/conv.py BatchNorm2d.num_batches_tracked += 1, for every layer in one multi-tensor launch
            torch._foreach_add_(self._bn_counters, 1)
            self._bn_counters = []
        outs = []
        for o in self.outputs:
            y = torch.empty((o.N, o.C, o.H, o.W), device=self.device, dtype=torch.float32)
            to_nchw = self.lib.w2l_nhwc_bf16_to_nchw if self.bf16 else self.lib.w2l_nhwc_to_nchw
            check(to_nchw(s, o.N, o.C, o.H, o.W, o.ptr, o.cs, ptr(y)), "nhwc_to_nchw")
            outs.append(y)
        return outs

    def backward(self, gouts, input_needs, want, reducer=None):
        s = current_stream()
        if not getattr(self, "_bwd_planned", False):
            self._plan_bwd_fusion()
        for n in self.nodes:                       # sums left by a backward pass that did not reach their block: stale
            if getattr(n, "_bwd_sums", None) is not None:
                n._bwd_sums = None
        self._wstream_on = self._wstream is not None and self.events is None and reducer is None
        if self._wstream_on:
            self._wstream.wait_stream(torch.cuda.current_stream())   # last step's optimiser reads of dW are ordered before
        written = {}   # id(grad buffer) -> list of (lo, hi) channel intervals holding a gradient

        def covered(a):
            iv = written.get(id(a.buf), [])
            lo, hi = a.off, a.off + a.C
            if any(l <= lo and hi <= h for l, h in iv):
                return True
            if any(lo < h and l < hi for l, h in iv):
                raise RuntimeError("train graph: partially overlapping gradient slices")
            return False

        def mark(a):
            written.setdefault(id(a.buf), []).append((a.off, a.off + a.C))

        for o, g in zip(self.outputs, gouts):
            ga = self.grad_act(o)
            if g is None:
                ga.buf[..., ga.off:ga.off + self.rnd(ga.C)].zero_()
            else:
                g = g.contiguous().float()
                to_nhwc = self.lib.w2l_nchw_to_nhwc_bf16 if self.bf16 else self.lib.w2l_nchw_to_nhwc
                check(to_nhwc(s, o.N, o.C, o.H, o.W, ptr(g), ga.ptr, ga.cs, self.rnd(o.C)), "nchw_to_nhwc")
            mark(ga)
        # Which buffers carry a gradient anybody asked for (what torch's engine decides per tensor with requires_grad): a graph
        # input that needs one, or the output of a node that owns a wanted parameter or reads such a buffer.  A node outside that
        # set is skipped entirely - e.g. the audio encoder of the frozen SyncNet in wav2lip_train.py:187-190,216: its input (the
        # mel) needs no gradient and its parameters are frozen, so the reference's autograd never walks it either.
        input_bufs = {id(a.buf): need for (a, _), need in zip(self.inputs, input_needs)}
        buf_req = {id(a.buf): bool(need) for (a, _), need in zip(self.inputs, input_needs)}
        node_req = {}
        for n in self.nodes:
            owns = any(p is not None and want(p) for p in n.parameters())
            r = owns or buf_req.get(id(n.x.buf), False) or not BWD_PRUNE[0]
            node_req[id(n)] = r
            buf_req[id(n.y.buf)] = buf_req.get(id(n.y.buf), False) or r
        grads = {}
        self._zero_pool = None         # a fresh zero buffer per backward pass (zero_slice): its slices become parameter gradients
        self.backward_nodes = []       # names of the nodes the last backward pass ran (tests)
        self.profile_mark("gouts")
        def bwd(n):
            gy = self.grad_act(n.y)
            if not covered(gy):
                return     # nothing downstream used this output
            if not node_req[id(n)]:
                return     # no parameter of this node and nothing upstream of it wants a gradient
            need_x = buf_req.get(id(n.x.buf), False) if BWD_PRUNE[0] else input_bufs.get(id(n.x.buf), True)
            self.backward_nodes.append(n.name)
            gx = self.grad_act(n.x, n.cin) if need_x else None
            acc = covered(gx) if gx is not None else False
            fresh = n.backward(gy, gx, acc, want)
            if reducer is not None:
                reducer.on_grads(fresh)        # bucketed all-reduce of these gradients starts while earlier blocks run
            else:
                grads.update(fresh)
            if gx is not None and not acc:
                mark(gx)

        # Host bookkeeping (written intervals, gradient dict) runs in reverse node order either way; with lanes only the
        # LAUNCHES of the side-lane nodes of a phase go to the side stream (their buffers are disjoint from lane 0's).
        if self.events is not None or self._side is None or reducer is not None:
            for n in reversed(self.nodes):
                bwd(n)
        else:
            for phase in reversed(list(self._phases())):
                self._run_lanes(list(reversed(phase)), bwd)
        din = []
        for (a, cch), need in zip(self.inputs, input_needs):
            if not need:
                din.append(None)
                continue
            ga = self.grad_act(a, cch)
            t = torch.empty((a.N, cch, a.H, a.W), device=self.device, dtype=torch.float32)
            if not covered(ga):
                t.zero_()
            else:
                to_nchw = self.lib.w2l_nhwc_bf16_to_nchw if self.bf16 else self.lib.w2l_nhwc_to_nchw
                check(to_nchw(s, a.N, cch, a.H, a.W, ga.ptr, ga.cs, ptr(t)), "nhwc_to_nchw")
            din.append(t)
        if reducer is not None:
            grads = reducer.finalize()
        if self._wstream_on:
            torch.cuda.current_stream().wait_stream(self._wstream)    # every weight gradient is complete before it is returned
            self._wstream_on = False
        return din, grads


class RawConvShape:
    """output-size arithmetic of a conv module without building a handle"""

    def __init__(self, conv, transposed):
        self.k = engine._pair(conv.kernel_size)
        self.s = engine._pair(conv.stride)
        self.p = engine._pair(conv.padding)
        self.op = engine._pair(conv.output_padding) if transposed else (0, 0)
        self.t = transposed

    def out_hw(self, H, W):
        if self.t:
            return ((H - 1) * self.s[0] - 2 * self.p[0] + self.k[0] + self.op[0],
                    (W - 1) * self.s[1] - 2 * self.p[1] + self.k[1] + self.op[1])
        return ((H + 2 * self.p[0] - self.k[0]) // self.s[0] + 1, (W + 2 * self.p[1] - self.k[1]) // self.s[1] + 1)


# ---------------------------------------------------------------- graph builders
def build_generator(model, N, H, W, device):
    """models/wav2lip.py:87-125 with every activation kept; skip concats are channel slices of shared buffers"""
    from .models.conv import PlainConv
    from ._lib import ACT_SIGMOID
    g = TrainGraph(device)
    enc, dec = model.face_encoder_blocks, model.face_decoder_blocks
    x_in = g.act(g.buffer(N, H, W, 6), 0, 8)
    mel_in = g.act(g.buffer(N, 80, 16, 1), 0, g.rnd(1))
    g.inputs = [(mel_in, 1), (x_in, 6)]
    enc_hw, h, w = [], H, W
    for blk in enc:
        for b in blk:
            h, w = RawConvShape(*describe(b)[0:4:3]).out_hw(h, w)
        enc_hw.append((h, w, describe(blk[-1])[0].out_channels))
    nb = len(dec)
    cats = []
    for i, blk in enumerate(dec):
        eh, ew, ec = enc_hw[nb - 1 - i]
        dc = describe(blk[-1])[0].out_channels
        cats.append((g.buffer(N, eh, ew, dc + ec), dc, ec))
    x = x_in
    for i, blk in enumerate(enc):
        buf, dc, ec = cats[nb - 1 - i]
        x = g.chain("face_encoder_blocks.%d" % i, list(blk), x, g.act(buf, dc, ec))
    a = g.chain("audio_encoder", list(model.audio_encoder), mel_in, lane=1)   # independent of the face encoder: side stream
    if (a.H, a.W) != (1, 1):
        raise RuntimeError("audio encoder must reduce the mel window to 1x1, got %dx%d" % (a.H, a.W))
    g.barrier()                                                                # the decoder needs both encoders
    x = a
    for i, blk in enumerate(dec):
        buf, dc, ec = cats[i]
        g.chain("face_decoder_blocks.%d" % i, list(blk), x, g.act(buf, 0, dc))
        x = g.act(buf, 0, dc + ec)
    head = PlainConv(model.output_block[1], ACT_SIGMOID)
    out = g.chain("output_block", [model.output_block[0], head], x)
    g.outputs = [out]
    g.keep = head
    return g


def build_block(blk, N, H, W, device):
    """one stand-alone block (models/conv.py:5-44) as a one-node train graph: `blk(x)` in train mode / under autograd"""
    g = TrainGraph(device)
    cin = describe(blk)[0].in_channels
    x_in = g.act(g.buffer(N, H, W, cin), 0, g.rnd(cin))
    g.inputs = [(x_in, cin)]
    g.outputs = [g.chain("block", [blk], x_in)]
    return g


def build_syncnet(model, N, H, W, device):
    g = TrainGraph(device)
    face_in = g.act(g.buffer(N, H, W, 15), 0, 16)
    mel_in = g.act(g.buffer(N, 80, 16, 1), 0, g.rnd(1))
    g.inputs = [(mel_in, 1), (face_in, 15)]
    f = g.chain("face_encoder", list(model.face_encoder), face_in)
    a = g.chain("audio_encoder", list(model.audio_encoder), mel_in, lane=1)    # independent branch: side stream
    for o in (f, a):
        if (o.H, o.W) != (1, 1):
            raise RuntimeError("SyncNet encoders must end at 1x1, got %dx%d" % (o.H, o.W))
    g.outputs = [a, f]
    return g


def build_disc(model, N, H, W, device):
    from .models.conv import PlainConv
    from ._lib import ACT_SIGMOID
    g = TrainGraph(device)
    x_in = g.act(g.buffer(N, H, W, 3), 0, g.rnd(3))
    g.inputs = [(x_in, 3)]
    x = x_in
    for i, blk in enumerate(model.face_encoder_blocks):
        x = g.chain("face_encoder_blocks.%d" % i, list(blk), x)
    head = PlainConv(model.binary_pred[0], ACT_SIGMOID)
    g.outputs = [g.chain("binary_pred", [head], x)]
    g.keep = head
    return g


# ---------------------------------------------------------------- the autograd node
class GraphCache:
    """train graphs of one module keyed by input geometry; a graph is busy from a grad-recording forward until its
    backward, so two live forwards (D(real) / D(fake)) get two buffer sets"""

    MAX_LIVE = 4

    def __init__(self, builder):
        self.builder = builder
        self.graphs = {}
        self.reducer = None     # sharding.GradReducer: multi-GPU gradient averaging overlapped with the backward pass

    def acquire(self, model, key, *shape):
        mode = tuple(m.training for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d))
        lst = self.graphs.setdefault(key + (mode, engine.TRAIN_PRECISION[0]), [])
        for g in lst:
            if not g.busy:
                break
        else:
            if len(lst) >= self.MAX_LIVE:
                raise RuntimeError("wav2lip_amd: %d forward passes of one module are waiting for backward(); "
                                   "call backward() or run inference under torch.no_grad()" % len(lst))
            g = self.builder(model, *shape)
            lst.append(g)
        g.busy = True
        g.ticket += 1
        return g


class GraphFn(torch.autograd.Function):
    """forward(cache entry, n_inputs, *inputs, *parameters) -> NCHW outputs; backward returns input and parameter grads"""

    @staticmethod
    def forward(ctx, graph, n_in, *tensors):
        outs = graph.forward(tensors[:n_in])
        ctx.graph, ctx.ticket, ctx.n_in = graph, graph.ticket, n_in
        ctx.params = tensors[n_in:]
        if not any(ctx.needs_input_grad):
            graph.busy = False
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        g = ctx.graph
        if g.ticket != ctx.ticket or not g.busy:
            raise RuntimeError("wav2lip_amd: the saved activations of this forward were overwritten by a later forward "
                               "of the same module (static buffers); call backward() before reusing the module")
        needs = ctx.needs_input_grad[2:]
        n_in = ctx.n_in
        want_ptrs = {p.data_ptr() for p, need in zip(ctx.params, needs[n_in:]) if need}
        din, grads = g.backward(gouts, needs[:n_in], lambda p: p.data_ptr() in want_ptrs, getattr(g, "reducer", None))
        g.busy = False
        dparams = []
        for p, need in zip(ctx.params, needs[n_in:]):
            gp = grads.get(p.data_ptr()) if need else None
            if need and gp is None:
                gp = torch.zeros_like(p)
            dparams.append(gp)
        return (None, None) + tuple(din) + tuple(dparams)


def needs_graph(module, inputs):
    """True when the call must run on the recorded (train-graph) path: a BatchNorm is in batch-statistics mode, or grad
    mode is on and (an input requires grad, or the module is in train mode with a trainable parameter).
    Otherwise the inference plan runs (BN folded, no saved activations)."""
    if any(m.training for m in module.modules() if isinstance(m, torch.nn.BatchNorm2d)):
        return True   # batch statistics (and running-stat updates) exist only on the recorded path
    if not torch.is_grad_enabled():
        return False
    if any(t.requires_grad for t in inputs):
        return True
    return module.training and any(p.requires_grad for p in module.parameters())


def run_graph(cache, model, key, shape, inputs):
    for t in inputs:
        engine.require_cuda(t, "input")
    g = cache.acquire(model, key, *shape)
    g.reducer = cache.reducer
    params = [p for p in model.parameters()]
    try:
        return GraphFn.apply(g, len(inputs), *inputs, *params)
    except Exception:
        g.busy = False
        raise


# ---------------------------------------------------------------- small differentiable ops of the training scripts
class L2NormRows(torch.autograd.Function):
    """F.normalize(x, p=2, dim=1) on [N, C] (models/syncnet.py:62-63)"""

    @staticmethod
    def forward(ctx, x):
        engine.require_cuda(x, "x")
        x = x.contiguous().float()
        N, Cn = x.shape
        y = torch.empty_like(x)
        check(_lib.load().w2l_l2norm_rows(current_stream(), N, Cn, ptr(x), Cn, ptr(y)), "l2norm_rows")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        N, Cn = x.shape
        dy = dy.contiguous().float()
        dx = torch.empty_like(x)
        check(_lib.load().w2l_l2norm_bwd(current_stream(), N, Cn, ptr(x), Cn, ptr(dy), ptr(dx), Cn), "l2norm_bwd")
        return dx


class CosineBCE(torch.autograd.Function):
    """BCELoss(cosine_similarity(a, v).unsqueeze(1), y), mean (wav2lip_train.py:179-184)"""

    @staticmethod
    def forward(ctx, a, v, y):
        engine.require_cuda(a, "a")
        a = a.contiguous().float()
        v = v.contiguous().float()
        y = y.contiguous().float().view(-1).to(a.device)
        N, Cn = a.shape
        cos = torch.empty(N, device=a.device, dtype=torch.float32)
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        check(_lib.load().w2l_cosine_bce(current_stream(), N, Cn, ptr(a), ptr(v), ptr(y), ptr(cos), ptr(loss)), "cosine_bce")
        ctx.save_for_backward(a, v, y)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        a, v, y = ctx.saved_tensors
        N, Cn = a.shape
        gout = gout.contiguous().float().view(1)
        da, dv = torch.empty_like(a), torch.empty_like(v)
        check(_lib.load().w2l_cosine_bce_bwd(current_stream(), N, Cn, ptr(a), ptr(v), ptr(y), ptr(gout), ptr(da), ptr(dv)),
              "cosine_bce_bwd")
        return da, dv, None


class BCEMean(torch.autograd.Function):
    """F.binary_cross_entropy(p, y) / nn.BCELoss()(p, y), mean (models/wav2lip.py:171, hq_wav2lip_train.py:249,253)"""

    @staticmethod
    def forward(ctx, p, y):
        engine.require_cuda(p, "p")
        shape = p.shape
        p = p.contiguous().float().view(-1)
        y = y.contiguous().float().view(-1).to(p.device)
        out = torch.empty(1, device=p.device, dtype=torch.float32)
        check(_lib.load().w2l_bce_mean(current_stream(), p.numel(), ptr(p), ptr(y), ptr(out)), "bce_mean")
        ctx.save_for_backward(p, y)
        ctx.shape = shape
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        p, y = ctx.saved_tensors
        gout = gout.contiguous().float().view(1)
        dp = torch.empty_like(p)
        check(_lib.load().w2l_bce_bwd(current_stream(), p.numel(), ptr(p), ptr(y), ptr(gout), ptr(dp)), "bce_bwd")
        return dp.view(ctx.shape), None


class L1Mean(torch.autograd.Function):
    """nn.L1Loss()(a, b), mean (wav2lip_train.py:191,227)"""

    @staticmethod
    def forward(ctx, a, b):
        engine.require_cuda(a, "a")
        a = a.contiguous().float()
        b = b.contiguous().float().to(a.device)
        if a.shape != b.shape:
            raise RuntimeError("l1 loss: shapes differ: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        check(_lib.load().w2l_l1_mean(current_stream(), a.numel(), ptr(a), ptr(b), ptr(out)), "l1_mean")
        ctx.save_for_backward(a, b)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        gout = gout.contiguous().float().view(1)
        da = torch.empty_like(a)
        check(_lib.load().w2l_l1_bwd(current_stream(), a.numel(), ptr(a), ptr(b), ptr(gout), ptr(da)), "l1_bwd")
        db = -da if ctx.needs_input_grad[1] else None
        return da, db
