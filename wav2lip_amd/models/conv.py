"""Building blocks with the reference's names and state-dict layout (models/conv.py:5-44):

  Conv2d           conv -> BatchNorm2d -> (+x if residual) -> ReLU          keys conv_block.0.*, conv_block.1.*
  nonorm_Conv2d    conv -> LeakyReLU(0.01)                                  keys conv_block.0.*
  Conv2dTranspose  ConvTranspose2d -> BatchNorm2d -> ReLU                   keys conv_block.0.*, conv_block.1.*

The torch modules inside `conv_block` are parameter containers only (checkpoint compatibility, optimiser
state); their torch forward is never called.  Compute is one fused HIP launch per block.
"""
import torch
from torch import nn

from .. import engine
from .._lib import ACT_LEAKY, ACT_RELU


class _FusedBlock(nn.Module):
    _transposed = False
    _act = ACT_RELU
    _norm = True

    def _setup(self, conv, cout, residual):
        parts = [conv] + ([nn.BatchNorm2d(cout)] if self._norm else [])
        self.conv_block = nn.Sequential(*parts)
        self.residual = bool(residual) and self._norm  # nonorm_Conv2d ignores `residual` (models/conv.py:21-31)
        self._fused = None
        self._fused_version = None

    def fused(self):
        """the BN-FOLDED HIP layer for the current parameters (inference plans); re-packed when they change.  A block whose
        BatchNorm is in batch-statistics mode has no folded form: train-mode calls run on the recorded path
        (wav2lip_amd/autograd.py: TrainGraph - whole networks through their GraphCache, a stand-alone block through
        `forward` below), never through this method."""
        if self._norm and self.conv_block[1].training:     # the BatchNorm's own flag decides (a .train() model whose BatchNorms
            raise RuntimeError(                            # were switched to .eval() folds like an eval model)
                "wav2lip_amd: a train-mode BatchNorm block cannot be BN-folded into an inference plan; call the network (or the "
                "block) - the call is routed to the train graph (autograd.TrainGraph) - or put the module in .eval() first")
        ver = engine.param_version(self)
        if self._fused is None or self._fused_version != ver:
            conv = self.conv_block[0]
            bn = self.conv_block[1] if self._norm else None
            self._fused = engine.FusedConv(conv, bn, self._act, transposed=self._transposed)
            self._fused_version = ver
        return self._fused

    def forward(self, x):
        """stand-alone use of one block on an NCHW tensor (whole models run through a Plan / TrainGraph instead)"""
        from .. import autograd
        engine.require_cuda(x, "input")
        with engine.on_device_of(x, self):
            if autograd.needs_graph(self, (x,)):
                # train mode (batch statistics) or a gradient-recording call: a one-node train graph, forward + backward on HIP
                if getattr(self, "_block_graphs", None) is None:
                    object.__setattr__(self, "_train_graphs", autograd.GraphCache(autograd.build_block))
                    object.__setattr__(self, "_block_graphs", True)
                N, Cin, H, W = x.shape
                return autograd.run_graph(self._train_graphs, self, (N, H, W, str(x.device)), (N, H, W, x.device),
                                          (x.contiguous().float(),))[0]
            layer = self.fused()
            N, Cin, H, W = x.shape
            lib = engine._lib.load()
            stream = engine._lib.current_stream()
            xin = engine.new_buf(N, H, W, layer.cin_p, x.device, zero=(layer.cin_p != Cin))
            engine.check(lib.w2l_nchw_to_nhwc(stream, N, Cin, H, W, engine.ptr(x.contiguous().float()),
                                              engine.ptr(xin), layer.cin_p, layer.cin_p), "nchw_to_nhwc")
            ho, wo = layer.out_hw(H, W)
            y = engine.new_buf(N, ho, wo, layer.cout, x.device)
            res = xin if self.residual else None
            layer.forward_raw(N, H, W, engine.ptr(xin), layer.cin_p, engine.ptr(y), layer.cout,
                              engine.ptr(res), layer.cin_p if res is not None else 0, stream)
            return y.permute(0, 3, 1, 2)


class Conv2d(_FusedBlock):
    def __init__(self, cin, cout, kernel_size, stride, padding, residual=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._setup(nn.Conv2d(cin, cout, kernel_size, stride, padding), cout, residual)
        self.act = nn.ReLU()


class nonorm_Conv2d(_FusedBlock):
    _act = ACT_LEAKY
    _norm = False

    def __init__(self, cin, cout, kernel_size, stride, padding, residual=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._setup(nn.Conv2d(cin, cout, kernel_size, stride, padding), cout, False)
        self.act = nn.LeakyReLU(0.01, inplace=True)


class Conv2dTranspose(_FusedBlock):
    _transposed = True

    def __init__(self, cin, cout, kernel_size, stride, padding, output_padding=0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._setup(nn.ConvTranspose2d(cin, cout, kernel_size, stride, padding, output_padding), cout, False)
        self.act = nn.ReLU()


class PlainConv(nn.Module):
    """Adapter giving a bare nn.Conv2d (+ following Sigmoid) the `fused()` protocol:
    the RGB head models/wav2lip.py:84-85 and the disc's binary_pred models/wav2lip.py:152."""

    residual = False

    def __init__(self, conv, act):
        super().__init__()
        object.__setattr__(self, "_conv", conv)  # not registered: the owner already holds it in its tree
        self._act = act
        self._fused = None
        self._fused_version = None

    def fused(self):
        conv = self._conv
        ver = engine.param_version(conv)
        if self._fused is None or self._fused_version != ver:
            self._fused = engine.FusedConv(conv, None, self._act)
            self._fused_version = ver
        return self._fused


class HeadFusedBlock(nn.Module):
    """`block` (Conv2d / nonorm_Conv2d) followed by a bare 1x1 nn.Conv2d + activation, executed as ONE launch: the
    1x1 contraction runs in the block's epilogue (output_block of the generator, models/wav2lip.py:83-85)."""

    residual = False

    def __init__(self, block, conv1x1, act):
        super().__init__()
        object.__setattr__(self, "_block", block)    # not registered: the owner already holds both in its tree
        object.__setattr__(self, "_conv", conv1x1)
        self._act = act
        self._fused = None
        self._fused_version = None

    def fused(self):
        blk = self._block
        if blk._norm and blk.conv_block[1].training:
            return blk.fused()   # raises: a train-mode BatchNorm block has no folded form
        ver = engine.param_version(blk) + engine.param_version(self._conv)
        if self._fused is None or self._fused_version != ver:
            self._fused = engine.FusedConv(blk.conv_block[0], blk.conv_block[1] if blk._norm else None, blk._act,
                                           transposed=blk._transposed, head=(self._conv, self._act))
            self._fused_version = ver
        return self._fused
