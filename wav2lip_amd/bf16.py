"""Host side of the bf16-STORAGE training path (BASELINE configs[3] / [4] name bf16): thin marshalling over the `w2l_convb_*`,
`w2l_conv_wgrad_bf16` and `w2l_*_bf16` entry points of libw2l_hip.so (include/w2l_hip.h, "training in bf16").

Layout: NHWC bf16, channel stride in ELEMENTS and a multiple of 8 (16-byte pixel rows), pad channels zero.  Master weights,
weight gradients, BatchNorm parameters / statistics, losses and the optimiser stay fp32 (wav2lip_amd/autograd.py builds the
train graphs; this module only holds the layer handle and layout helpers).  No CPU path: a missing library or a CPU tensor
raises RuntimeError.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvGeom, check, current_stream, ptr


def round8(c):
    return (c + 7) // 8 * 8


class ActB:
    """channel slice [off, off+C) of an NHWC bf16 buffer [N, H, W, Ctot]"""

    def __init__(self, buf, off, C_):
        if buf.dtype != torch.bfloat16:
            raise RuntimeError("wav2lip_amd.bf16: buffer must be bfloat16, got %s" % buf.dtype)
        self.buf, self.off, self.C = buf, off, C_
        self.N, self.H, self.W, self.cs = buf.shape
        if off % 8 or self.cs % 8:
            raise RuntimeError("wav2lip_amd.bf16: channel offset %d / stride %d must be multiples of 8" % (off, self.cs))

    @property
    def ptr(self):
        return C.c_void_p(self.buf.data_ptr() + 2 * self.off)

    def view(self):
        """[N, C, H, W] strided torch view of the slice (zero-copy, bfloat16)"""
        return self.buf[..., self.off:self.off + self.C].permute(0, 3, 1, 2)


def new_buf(N, H, W, Cn, device):
    """zeroed NHWC bf16 buffer with the channel count rounded up to 8"""
    return torch.zeros((N, H, W, round8(Cn)), device=device, dtype=torch.bfloat16)


class ConvB:
    """one `w2l_convb` handle: y = act(conv(x, W) * scale + shift (+ res)) over bf16 NHWC tensors; `weight` is the fp32 master
    tensor in torch layout (re-packed to bf16 by update())"""

    def __init__(self, geom, weight):
        self._lib = _lib.load()
        self.geom = geom
        if not weight.is_cuda:
            raise RuntimeError("wav2lip_amd.bf16: weights must live on a HIP device; this engine has no CPU path")
        w = weight.detach().contiguous().float()
        h = C.c_void_p()
        check(self._lib.w2l_convb_create(C.byref(geom), ptr(w), current_stream(), C.byref(h)), "convb_create")
        self.handle = h
        self.cin, self.cout = geom.cin, geom.cout

    def update(self, weight):
        w = weight.detach().contiguous().float()
        check(self._lib.w2l_convb_update(self.handle, ptr(w), current_stream()), "convb_update")
        self._keep = w      # stream-ordered: alive until the next update replaces it

    @staticmethod
    def update_many(pairs):
        """re-pack every (ConvB, fp32 master weight) pair in ONE launch (w2l_convb_update_many: what an optimiser step
        invalidates).  A weight that is not already a contiguous fp32 device tensor goes through its own update()."""
        if not pairs:
            return
        many, keep = [], []
        for layer, weight in pairs:
            w = weight.detach()
            if w.dtype == torch.float32 and w.is_contiguous() and w.is_cuda:
                many.append((layer, w))
            else:
                layer.update(weight)
        if not many:
            return
        n = len(many)
        handles = (C.c_void_p * n)(*[layer.handle.value for layer, _ in many])
        weights = (C.c_void_p * n)(*[w.data_ptr() for _, w in many])
        check(many[0][0]._lib.w2l_convb_update_many(n, handles, weights, current_stream()), "convb_update_many")

    def out_hw(self, H, W):
        ho, wo = C.c_int(), C.c_int()
        check(self._lib.w2l_conv_out_hw(C.byref(self.geom), H, W, C.byref(ho), C.byref(wo)), "conv_out_hw")
        return ho.value, wo.value

    def set_tile(self, tile):
        check(self._lib.w2l_convb_set_tile(self.handle, tile), "convb_set_tile")

    def run(self, x, y, res=None, scale=None, shift=None, ksplit=0):
        check(self._lib.w2l_convb_forward(self.handle, current_stream(), x.N, x.H, x.W, x.ptr, x.cs, y.ptr, y.cs,
                                          res.ptr if res is not None else None, res.cs if res is not None else 0,
                                          ptr(scale), ptr(shift), ksplit), "convb_forward")

    def run_bn(self, x, z, bias, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift):
        """z = conv(x) + bias AND the batch statistics of z (w2l_convb_forward_bn: sums taken in the conv epilogue)"""
        check(self._lib.w2l_convb_forward_bn(self.handle, current_stream(), x.N, x.H, x.W, x.ptr, x.cs, z.ptr, z.cs, ptr(bias),
                                             ptr(gamma), ptr(beta), float(eps), float(momentum), ptr(running_mean), ptr(running_var),
                                             ptr(mean), ptr(rstd), ptr(scale), ptr(shift)), "convb_forward_bn")

    STORE_MASKED = 0x100      # W2L_BNBWD_STORE_MASKED

    def run_bnbwd(self, x, y, res, bz, by, bact, mean, rstd, bscale, bshift, dgamma, dbeta, store_masked=False):
        """y = conv(x) (+ res) AND the BatchNorm-backward column sums of the block whose dy this output is (w2l_convb_forward_bnbwd);
        returns True when the sums were written in the epilogue (False: split-K launch, the caller reduces).  `store_masked`
        (ReLU blocks): a fused launch stores the masked gradient g = dy * [block output > 0] instead of dy."""
        fused = C.c_int(0)
        if store_masked:
            bact = int(bact) | self.STORE_MASKED
        check(self._lib.w2l_convb_forward_bnbwd(self.handle, current_stream(), x.N, x.H, x.W, x.ptr, x.cs, y.ptr, y.cs,
                                                res.ptr if res is not None else None, res.cs if res is not None else 0,
                                                bz.ptr, bz.cs, by.ptr if by is not None else None, by.cs if by is not None else 0,
                                                int(bact), ptr(mean), ptr(rstd), ptr(bscale), ptr(bshift), ptr(dgamma), ptr(dbeta),
                                                C.byref(fused)), "convb_forward_bnbwd")
        return bool(fused.value)

    def run_actbwd(self, x, y, res, by, bact, dbias=None):
        """y = (conv(x) (+ res)) * act'(by): the data-gradient launch that completes the dy of an activation block WITHOUT BatchNorm
        stores that block's dz directly (w2l_convb_forward_actbwd) and, with `dbias` (fp32, round8(cout) entries), the column sums
        of that dz = the block's bias gradient; False when the launch could not carry the mask (split-K, a special-case kernel):
        plain dy was stored, dbias untouched"""
        fused = C.c_int(0)
        check(self._lib.w2l_convb_forward_actbwd(self.handle, current_stream(), x.N, x.H, x.W, x.ptr, x.cs, y.ptr, y.cs,
                                                 res.ptr if res is not None else None, res.cs if res is not None else 0,
                                                 by.ptr, by.cs, int(bact), ptr(dbias), C.byref(fused)), "convb_forward_actbwd")
        return bool(fused.value)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.w2l_convb_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
