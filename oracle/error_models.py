"""Error models applied to the EXACT (fp64) oracle graphs - test infrastructure (only tests/ and tests/golden/ import this).

`bf16_storage_noise(seed)`: inside the context every F.conv2d / F.conv_transpose2d sees its input, its weight and its output -
and, on the way back, their gradients - jittered by eps = 2**-8 relative, the rounding error bound of bf16 (8 significant bits).
It is the bf16-storage training path's error model (wav2lip_amd/autograd.py NodeB: x, z, y, dy, dz in bf16) applied to the exact
graph: the spread of the results over seeds measures how far bf16 rounding ALONE can move a loss or a gradient of a case - the
yardstick the HIP path is held to instead of a fitted percentage.  (Same model as tests/test_train_gpu.py uses at small batches.)
"""
import torch
import torch.nn.functional as F


class _Jitter(torch.autograd.Function):
    """y = x * (1 + eps * u), u ~ U(-1, 1) per element, forward AND backward (independent draws)"""

    @staticmethod
    def forward(ctx, x, eps, gen):
        ctx.eps, ctx.gen = eps, gen
        return x * (1 + eps * (2 * torch.rand(x.shape, generator=gen, dtype=x.dtype) - 1))

    @staticmethod
    def backward(ctx, g):
        return g * (1 + ctx.eps * (2 * torch.rand(g.shape, generator=ctx.gen, dtype=g.dtype) - 1)), None, None


class bf16_storage_noise:
    def __init__(self, seed, eps=2.0 ** -8):
        self.eps, self.gen = eps, torch.Generator().manual_seed(seed)

    def __enter__(self):
        self.c, self.ct = F.conv2d, F.conv_transpose2d
        eps, gen = self.eps, self.gen

        def wrap(fn):
            def f(x, w, b=None, **kw):
                return _Jitter.apply(fn(_Jitter.apply(x, eps, gen), _Jitter.apply(w, eps, gen), b, **kw), eps, gen)
            return f
        F.conv2d, F.conv_transpose2d = wrap(self.c), wrap(self.ct)

    def __exit__(self, *a):
        F.conv2d, F.conv_transpose2d = self.c, self.ct
