"""`Adam` with the torch.optim.Adam interface the reference constructs (wav2lip_train.py:359-360,
color_syncnet_train.py:270-271, hq_wav2lip_train.py:418-421: `optim.Adam([p for p in model.parameters() if
p.requires_grad], lr=..., betas=...)`), executed as ONE fused multi-tensor HIP launch per step (w2l_adam_step).

Restriction against torch.optim.Adam: a group is updated as a whole - if only SOME of its parameters have `grad is None` the step
raises (torch would skip those); the mirrored networks always produce a gradient for every trainable parameter of a module
that took part in the loss (exact zeros where a parameter had no influence, e.g. a conv bias in front of a batch-statistics
BatchNorm), so the reference's loops never hit it.

State layout and `state_dict()` / `load_state_dict()` follow torch (per-parameter `step`, `exp_avg`, `exp_avg_sq`;
param_groups with lr / betas / eps / weight_decay), so the reference's checkpoints (`"optimizer"` entry,
wav2lip_train.py:289-297,311-316) round-trip.  Moments live in two flat fp32 arenas, one slice per parameter.
"""
import ctypes as C

import torch

from . import _lib, engine
from ._lib import AdamTensor, check, current_stream, ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._fused = {}   # group index -> (handle, params, table)

    def _group_state(self, gi, group):
        params = [p for p in group["params"]]
        ent = self._fused.get(gi)
        if ent is not None and [id(p) for p in ent[1]] == [id(p) for p in params]:
            return ent
        lib = _lib.load()
        for p in params:
            if not p.is_cuda:
                raise RuntimeError("wav2lip_amd.optim.Adam: parameters must live on a HIP device (no CPU path)")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("wav2lip_amd.optim.Adam: parameters must be contiguous fp32")
        sizes = (C.c_longlong * len(params))(*[p.numel() for p in params])
        h = C.c_void_p()
        check(lib.w2l_adam_create(len(params), sizes, C.byref(h)), "adam_create")
        total = sum(p.numel() for p in params)
        dev = params[0].device
        arena_m = torch.zeros(total, device=dev, dtype=torch.float32)
        arena_v = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in params:
            st = self.state[p]
            n = p.numel()
            m = arena_m[off:off + n].view_as(p)
            v = arena_v[off:off + n].view_as(p)
            if "exp_avg" in st:    # state restored by load_state_dict: move it into the arena
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = m, v
            if "step" not in st:
                st["step"] = torch.tensor(0.0)
            off += n
        table = (AdamTensor * len(params))()
        ent = (h, params, table, (arena_m, arena_v))
        self._fused[gi] = ent
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            if not group["params"]:
                continue
            if len({p.device for p in group["params"]}) != 1:
                raise RuntimeError("wav2lip_amd.optim.Adam: the parameters of one group must live on one device")
            with engine.on_device_of(group["params"][0]):
                self._step_group(lib, gi, group)
        return loss

    def _step_group(self, lib, gi, group):
        h, params, table, _ = self._group_state(gi, group)
        missing = [p for p in params if p.grad is None]
        if missing:
            if len(missing) == len(params):
                return
            raise RuntimeError("wav2lip_amd.optim.Adam: %d of %d parameters of a group have no gradient; the fused "
                               "step updates a whole group at once" % (len(missing), len(params)))
        steps = {int(self.state[p]["step"]) for p in params}
        if len(steps) != 1:
            raise RuntimeError("wav2lip_amd.optim.Adam: parameters of one group are at different step counts")
        step = steps.pop() + 1
        keep = []
        for i, p in enumerate(params):
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.contiguous().float()
                keep.append(g)
            st = self.state[p]
            table[i].param = p.data_ptr()
            table[i].grad = g.data_ptr()
            table[i].exp_avg = st["exp_avg"].data_ptr()
            table[i].exp_avg_sq = st["exp_avg_sq"].data_ptr()
            table[i].n = p.numel()
        b1, b2 = group["betas"]
        check(lib.w2l_adam_step(h, current_stream(), table, float(group["lr"]), float(b1), float(b2),
                                float(group["eps"]), float(group["weight_decay"]), step), "adam_step")
        step_t = torch.tensor(float(step))      # one host tensor per group and step, shared by its parameters
        for p in params:
            self.state[p]["step"] = step_t
        engine.PARAM_EPOCH[0] += 1   # parameters were written behind torch's version counters: invalidate packed weights
        self._keep = keep

    def _release(self):
        lib = _lib.load()
        for ent in self._fused.values():
            lib.w2l_adam_destroy(ent[0])
        self._fused = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._release()    # handles + arenas are rebuilt from the restored moments on the next step

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
