"""Host-side engine: turns the parameter containers of the mirrored modules (wav2lip_amd/models) into
fused HIP layers (libw2l_hip.so) and static NHWC buffer plans.

Data layout in HBM: every activation is NHWC fp32 `[N, H, W, Ctot]`; an `Act` is a channel slice
`[off, off+C)` of such a buffer, handed to the kernels as (pointer to channel `off`, channel stride Ctot).
Skip concats (reference models/wav2lip.py:104-114) are never materialised: the decoder block and the
encoder block that feed a concat write into disjoint channel slices of one buffer.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, ConvGeom, check, ptr


# Launch configurations (tile, split-K) come from the committed shape-keyed table (wav2lip_amd/tune_table.json, loaded by
# _lib.load) and, for shapes it does not hold, from the library's heuristic: both depend on the shape only, so a layer's
# summation order - and every bit of its result - is the same in every run and on every box.  W2L_AUTOTUNE=1 opts into
# stopwatch tuning (times the candidates once per plan; tools/make_tune_table.py uses it to regenerate the table): results
# then differ in the last bits from run to run.
AUTOTUNE = os.environ.get("W2L_AUTOTUNE", "0") == "1"


# Per-plan launch configurations for INFERENCE plans whose batch size has no entries in the shape-keyed table
# (wav2lip_amd/plan_configs.json, written by tools/batch_sweep.py --dump-configs on a GPU box).  The table only holds the batch
# sizes that were tuned (1, 8, 16, ..., 256), and the library's heuristic - which also serves the small training shapes whose
# golden gradients are anchored to its summation order, so it stays as it is - never splits K: a batch-2 step took longer than a
# batch-8 step.  The file holds one list [[launch name, configuration id, split-K], ...] per batch size: the batches under
# "table" as the table resolves them (never applied: those plans keep resolving through the table), the others tuned on their
# own.  A plan of any other batch size N borrows the list of the smallest listed batch >= N (the largest one beyond that):
# configuration ids depend on the layer geometry, not on N.  Only static inference plans ask (training graphs never do), the
# choice is a function of N alone, so outputs stay bit-reproducible.  W2L_PLAN_CONFIGS=0 switches it off (heuristic as before).
PLAN_CONFIGS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plan_configs.json")
_plan_configs = {}


def plan_configs_enabled():
    return (os.environ.get("W2L_PLAN_CONFIGS", "1") != "0" and not AUTOTUNE and not _lib.EXACT
            and not os.environ.get("W2L_TUNE_TABLE"))


def load_plan_configs(path=None):
    """{kind: {"table": [batch sizes resolved by the table], "plans": {batch: [[name, id, split-K], ...]}}} of the committed file
    (cached per path); a missing file is an empty dict"""
    import json
    path = path or PLAN_CONFIGS_PATH
    if path not in _plan_configs:
        doc = {}
        if os.path.exists(path):
            with open(path) as fh:
                doc = json.load(fh)
        for kind, d in doc.items():
            d["plans"] = {int(b): v for b, v in d["plans"].items()}
            d["table"] = sorted(int(b) for b in d.get("table", []))
        _plan_configs[path] = doc
    return _plan_configs[path]


def plan_config_source(kind, N, doc=None):
    """the batch size whose committed launch list a `kind` plan of batch N runs, or None (table / heuristic as before)"""
    d = (load_plan_configs() if doc is None else doc).get(kind)
    if not d or not d["plans"] or N in d["table"]:
        return None
    sizes = sorted(d["plans"])
    return next((m for m in sizes if m >= N), sizes[-1])


_PLAN_MISMATCH_WARNED = [False]


def apply_plan_configs(plan, kind, N, doc=None, strict=False):
    """set the explicit (configuration, split-K) of every launch of an inference plan from the committed per-plan lists; returns the
    batch size the list came from, or None when the plan is left to the table / heuristic.  A list that names other launches than
    the live plan (a stale plan_configs.json after a change of the launch list) is ignored with ONE warning - inference at an
    ordinary batch size must not fail over a tuning file; `strict=True` (tests, tools) raises instead."""
    if doc is None and not plan_configs_enabled():
        return None
    doc = load_plan_configs() if doc is None else doc
    src = plan_config_source(kind, N, doc)
    if src is None:
        return None
    cfg = doc[kind]["plans"][src]
    names = [r[0] for r in plan.records]
    if [c[0] for c in cfg] != names:
        msg = ("wav2lip_amd: plan_configs.json lists other launches than this %s plan (%d vs %d); regenerate it with "
               "tools/batch_sweep.py --dump-configs" % (kind, len(cfg), len(names)))
        if strict:
            raise RuntimeError(msg)
        if not _PLAN_MISMATCH_WARNED[0]:
            _PLAN_MISMATCH_WARNED[0] = True
            import warnings
            warnings.warn(msg + " - ignored, the launch table / heuristic is used")
        return None
    for i, (_, t, k) in enumerate(cfg):
        plan.set_config(i, int(t), int(k))
    return src


# Precision of the TRAINING path (wav2lip_amd/autograd.py):
#   "f32"   exact fp32 products, fp32 tensors (default; gradients pinned to the reference);
#   "bf16"  the bf16-STORAGE path BASELINE configs[3] / [4] name: activations, pre-BatchNorm conv outputs and their gradients are
#           NHWC bf16 in HBM, every contraction multiplies bf16 operands on the bf16 matrix cores with fp32 accumulation, master
#           weights / weight gradients / BatchNorm statistics / losses / Adam stay fp32 (csrc/conv_bf16.hip, wgrad_bf16.hip,
#           train_bf16.hip);
#   "bf16c" round 2's contraction-only variant (fp32 tensors, operands rounded inside the fp32-layout kernels), kept for A/B.
# Inference plans always run fp32 (the 1e-3 pixel parity path).
TRAIN_PRECISION = [os.environ.get("W2L_TRAIN_PRECISION", "f32")]


def set_train_precision(name):
    """"f32", "bf16" (bf16 storage) or "bf16c" (contractions only); applies to train graphs built afterwards"""
    if name not in ("f32", "bf16", "bf16c"):
        raise ValueError("train precision must be 'f32', 'bf16' or 'bf16c'")
    TRAIN_PRECISION[0] = name


# W2L_TWO_STREAMS=0 runs the generator's face and audio encoders back to back on one stream (default: concurrently on two)
TWO_STREAM_ENCODERS = os.environ.get("W2L_TWO_STREAMS", "1") != "0"


# W2L_HIP_GRAPHS=1 replays the generator's launch sequence from a captured HIP graph (pays at small batches, where the step is
# launch-bound; off by default)
HIP_GRAPHS = os.environ.get("W2L_HIP_GRAPHS", "0") == "1"


# W2L_WGRAD_OVERLAP=0 keeps the weight-gradient GEMMs of a backward pass on the main stream (default: on a side stream,
# overlapping the HBM-bound BatchNorm-backward passes and the data gradients of the critical path)
WGRAD_OVERLAP = os.environ.get("W2L_WGRAD_OVERLAP", "1") != "0"


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            "wav2lip_amd: %s must live on a HIP device (got %s); this engine has no CPU path" % (what, t.device))


class FusedConv:
    """One `w2l_conv` handle: act(conv(x, W) * scale + shift (+ res))."""

    def __init__(self, conv, bn, act, transposed=False, head=None):
        """head = (nn.Conv2d 1x1, act): fused into the epilogue (w2l_conv_attach_head); the layer then has
        `cout` = the head's output channels and `cout_inner` = the conv's"""
        lib = _lib.load()
        w = conv.weight.detach()
        require_cuda(w, "conv weight")
        w = w.contiguous().float()
        dev = w.device
        kh, kw = _pair(conv.kernel_size)
        sh, sw = _pair(conv.stride)
        ph, pw = _pair(conv.padding)
        oph, opw = _pair(conv.output_padding) if transposed else (0, 0)
        if _pair(conv.dilation) != (1, 1) or conv.groups != 1:
            raise RuntimeError("dilation/groups are not part of the Wav2Lip hot path")
        cin = conv.in_channels
        cout = conv.out_channels
        self.geom = ConvGeom(int(transposed), cin, cout, kh, kw, sh, sw, ph, pw, oph, opw, act)
        self.cin, self.cout = cin, cout
        self.cin_p = lib.w2l_conv_cin_padded(cin)
        self.device = dev
        scale = torch.empty(cout, device=dev, dtype=torch.float32)
        shift = torch.empty(cout, device=dev, dtype=torch.float32)
        bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
        stream = _lib.current_stream()
        if bn is not None:
            g = bn.weight.detach().float().contiguous() if bn.weight is not None else None
            b = bn.bias.detach().float().contiguous() if bn.bias is not None else None
            mu = bn.running_mean.detach().float().contiguous()
            var = bn.running_var.detach().float().contiguous()
            check(lib.w2l_bn_fold(stream, cout, ptr(bias), ptr(g), ptr(b), ptr(mu), ptr(var),
                                  float(bn.eps), ptr(scale), ptr(shift)), "bn_fold")
        else:
            check(lib.w2l_bn_fold(stream, cout, ptr(bias), None, None, None, None, 0.0,
                                  ptr(scale), ptr(shift)), "bn_fold")
        h = C.c_void_p()
        check(lib.w2l_conv_create(C.byref(self.geom), ptr(w), ptr(scale), ptr(shift), stream, C.byref(h)),
              "conv_create")
        self.handle = h
        self._lib = lib
        self.cout_inner = cout
        self.head_c = 0
        if head is not None:
            hconv, hact = head
            if _pair(hconv.kernel_size) != (1, 1) or _pair(hconv.stride) != (1, 1) or _pair(hconv.padding) != (0, 0) \
                    or hconv.in_channels != cout:
                raise RuntimeError("only a 1x1 stride-1 conv on the layer's output can be fused as a head")
            hw = hconv.weight.detach().float().contiguous().view(hconv.out_channels, cout)
            hb = hconv.bias.detach().float().contiguous() if hconv.bias is not None else None
            check(lib.w2l_conv_attach_head(h, ptr(hw), ptr(hb), hconv.out_channels, hact, stream), "conv_attach_head")
            self.head_c = hconv.out_channels
            self.cout = hconv.out_channels

    def tune_key(self, N, H, W, has_res=False, precision=0):
        """the shape key the library files a launch of this layer under (include/w2l_hip.h, "tune table"): transposed cin cout kh
        kw sh sw ph pw oph opw precision has_residual head_c N H W"""
        g = self.geom
        return (g.transposed, g.cin, g.cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.oph, g.opw, int(precision),
                int(bool(has_res)), int(self.head_c), int(N), int(H), int(W))

    def out_hw(self, H, W):
        ho, wo = C.c_int(), C.c_int()
        check(self._lib.w2l_conv_out_hw(C.byref(self.geom), H, W, C.byref(ho), C.byref(wo)), "conv_out_hw")
        return ho.value, wo.value

    def macs(self, N, H, W):
        m = int(self._lib.w2l_conv_macs(C.byref(self.geom), N, H, W))
        if self.head_c:
            ho, wo = self.out_hw(H, W)
            m += N * ho * wo * self.cout_inner * self.head_c
        return m

    def set_tile(self, tile_id):
        check(self._lib.w2l_conv_set_tile(self.handle, tile_id), "conv_set_tile")

    def forward_raw(self, N, H, W, x_ptr, x_cs, y_ptr, y_cs, res_ptr=None, res_cs=0, stream=None):
        check(self._lib.w2l_conv_forward(self.handle, stream or _lib.current_stream(), N, H, W,
                                         x_ptr, x_cs, y_ptr, y_cs, res_ptr, res_cs), "conv_forward")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.w2l_conv_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Act:
    """Channel slice [off, off+C) of an NHWC buffer `buf` of shape [N, H, W, Ctot]."""

    def __init__(self, buf, off, C_):
        self.buf, self.off, self.C = buf, off, C_
        self.N, self.H, self.W, self.cs = buf.shape

    @property
    def ptr(self):
        return C.c_void_p(self.buf.data_ptr() + 4 * self.off)

    def view(self):
        """[N, C, H, W] strided torch view of the slice (zero-copy)"""
        return self.buf[..., self.off:self.off + self.C].permute(0, 3, 1, 2)


def new_buf(N, H, W, Ctot, device, zero=False):
    f = torch.zeros if zero else torch.empty
    return f((N, H, W, Ctot), device=device, dtype=torch.float32)


class Plan:
    """A recorded sequence of fused-conv launches over fixed buffers (w2l_plan)."""

    def __init__(self):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.w2l_plan_create(C.byref(h)), "plan_create")
        self.handle = h
        self.tuned = False
        self.keep = []      # keeps FusedConv objects and buffers alive
        self.records = []   # (name, layer, N, H, W) for reporting
        self.has_res = []   # per launch: a residual tensor is added (part of the launch's tune-table key)

    def add(self, name, layer, src, dst, res=None):
        if src.C < layer.cin or src.cs - src.off < layer.cin_p:
            raise RuntimeError("plan %s: input slice has %d channels, layer wants %d" % (name, src.C, layer.cin))
        ho, wo = layer.out_hw(src.H, src.W)
        if (dst.H, dst.W) != (ho, wo) or dst.C != layer.cout or dst.N != src.N:
            raise RuntimeError("plan %s: output slice %s does not match %s" %
                               (name, (dst.N, dst.H, dst.W, dst.C), (src.N, ho, wo, layer.cout)))
        check(self._lib.w2l_plan_add_conv(self.handle, layer.handle, src.N, src.H, src.W, src.ptr, src.cs,
                                          dst.ptr, dst.cs, res.ptr if res is not None else None,
                                          res.cs if res is not None else 0), "plan_add_conv")
        self.keep += [layer, src.buf, dst.buf] + ([res.buf] if res is not None else [])
        self.records.append((name, layer, src.N, src.H, src.W))
        self.has_res.append(res is not None)

    def add_raw(self, other, index):
        """re-record launch `index` of plan `other` (same layer handle and buffers)"""
        check(self._lib.w2l_plan_copy_item(self.handle, other.handle, index), "plan_copy_item")
        self.records.append(other.records[index])
        self.has_res.append(other.has_res[index])
        self.keep.append(other)

    def run(self, stream=None):
        if not self.tuned and AUTOTUNE:
            self.autotune()
        check(self._lib.w2l_plan_run(self.handle, stream or _lib.current_stream()), "plan_run")

    def autotune(self, reps=2):
        """pick the fastest (tile, split-K) per launch by timing them on the device (w2l_plan_autotune)"""
        check(self._lib.w2l_plan_autotune(self.handle, _lib.current_stream(), reps), "plan_autotune")
        self.tuned = True

    def set_config(self, index, tile, ksplit=1):
        check(self._lib.w2l_plan_set_config(self.handle, index, tile, ksplit), "plan_set_config")

    def configs(self):
        out = []
        for i in range(self._lib.w2l_plan_size(self.handle)):
            t, k = C.c_int(), C.c_int()
            check(self._lib.w2l_plan_get_config(self.handle, i, C.byref(t), C.byref(k)), "plan_get_config")
            out.append((self.records[i][0], t.value, k.value))
        return out

    def save_configs(self, path):
        """write the tuned (tile, split-K) list as JSON (bench.py --tune-cache: profile runs skip the autotune)"""
        import json
        with open(path, "w") as fh:
            json.dump([[n, t, k] for n, t, k in self.configs()], fh)

    def load_configs(self, path):
        import json
        with open(path) as fh:
            cfg = json.load(fh)
        names = [r[0] for r in self.records]
        if [c[0] for c in cfg] != names:
            raise RuntimeError("tune cache %s was written for a different plan" % path)
        for i, (_, t, k) in enumerate(cfg):
            self.set_config(i, t, k)
        self.tuned = True

    def profile(self, reps=3):
        """per-launch milliseconds (HIP events on the current stream)"""
        n = self._lib.w2l_plan_size(self.handle)
        ms = (C.c_float * n)()
        check(self._lib.w2l_plan_profile(self.handle, _lib.current_stream(), reps, ms), "plan_profile")
        return [(self.records[i][0], float(ms[i]), self.records[i][1].macs(*self.records[i][2:]))
                for i in range(n)]

    def macs(self):
        return sum(r[1].macs(*r[2:]) for r in self.records)

    def executed_flops(self):
        """[(name, FLOPs the matrix cores execute for the launch with its current configuration)]: padded tiles and K,
        16 instead of 36 products per 2x2 tile on Winograd launches (w2l_plan_executed_flops)"""
        return [(n, f) for n, f, _, _ in self.resolved()]

    def resolved(self):
        """[(name, executed FLOPs, kernel family, (config id, split-K))] per launch as it would run now: explicit
        configuration, tune-table entry or heuristic; family = "igemm" | "wino" | "wino2" | "tp2" | "wino4" | "split" | "wino2s" | "tp2s" | "stem7s" | "k3s" (the
        kernel that runs it; "split" / "wino2s" = the implicit GEMM / F(2x2) Winograd with fp32 operands as three bf16 pieces, whose
        FLOPs are bf16 matrix-core FLOPs)"""
        n = self._lib.w2l_plan_size(self.handle)
        fl = (C.c_longlong * n)()
        cfg = (C.c_int * (2 * n))()
        check(self._lib.w2l_plan_executed_flops(self.handle, fl, cfg), "plan_executed_flops")
        fam = {0: "igemm", 1: "wino", 2: "wino2", 3: "tp2", 4: "wino4", 5: "split", 6: "wino2s", 7: "tp2s", 8: "stem7s", 9: "k3s"}
        return [(self.records[i][0], int(fl[i]), fam[self._lib.w2l_conv_config_family(int(cfg[2 * i]))],
                 (int(cfg[2 * i]), int(cfg[2 * i + 1]))) for i in range(n)]

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.w2l_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class BufPool:
    """Trivial free-list of NHWC scratch buffers keyed by shape: ping-pong reuse inside a plan."""

    def __init__(self, device):
        self.device = device
        self.free = {}
        self.total_bytes = 0

    def get(self, N, H, W, Ctot):
        key = (N, H, W, Ctot)
        lst = self.free.get(key)
        if lst:
            return lst.pop()
        self.total_bytes += 4 * N * H * W * Ctot
        return new_buf(N, H, W, Ctot, self.device)

    def put(self, buf):
        self.free.setdefault(tuple(buf.shape), []).append(buf)


# bumped by wav2lip_amd.optim.Adam.step(): the fused optimiser writes parameters without touching torch's per-tensor
# version counters, so every packed-weight cache also keys on this epoch
PARAM_EPOCH = [0]


def param_version(module):
    """fingerprint of a module's weights: re-pack when it changes.  A tuple of (storage address, torch version counter) per
    tensor plus the optimiser epoch - re-allocations (.to(), .data = ...) and in-place writes both show up, and two changes
    can never cancel as they could in a sum.  Writes that bypass torch's counters (`p.data.copy_()`) need `invalidate()`."""
    return (PARAM_EPOCH[0],) + tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


def invalidate():
    """force every packed-weight / BN-fold cache to rebuild on next use (after out-of-band writes to parameter storage)"""
    PARAM_EPOCH[0] += 1


class on_device_of:
    """`with on_device_of(tensor, module):` - makes the tensor's device current for the launches inside (the library enqueues on
    torch's CURRENT stream, which belongs to the current device) and refuses inputs that live on another device than the
    module's parameters."""

    def __init__(self, t, module=None):
        if module is not None:
            p = next(module.parameters(), None)
            if p is not None and p.device != t.device:
                raise RuntimeError("wav2lip_amd: input is on %s but the module's parameters are on %s" % (t.device, p.device))
        self._ctx = torch.cuda.device(t.device) if t.is_cuda else None

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)


def layers_of(seq):
    """fused layers of an nn.Sequential of mirrored blocks (builds lazily, cached on the block)"""
    return [blk.fused() for blk in seq]


def run_chain(plan, pool, name, blocks, src, final_dst=None):
    """Record a chain of blocks src -> ... -> final_dst (or a pooled scratch buffer); returns the last Act."""
    x = src
    owned = None  # scratch buffer currently holding x (to be released when dead)
    for j, blk in enumerate(blocks):
        layer = blk.fused()
        ho, wo = layer.out_hw(x.H, x.W)
        last = j == len(blocks) - 1
        if last and final_dst is not None:
            dst = final_dst
            new_owned = None
        else:
            buf = pool.get(x.N, ho, wo, layer.cout)
            dst = Act(buf, 0, layer.cout)
            new_owned = buf
        res = x if getattr(blk, "residual", False) else None
        plan.add("%s.%d" % (name, j), layer, x, dst, res)
        if owned is not None:
            pool.put(owned)
        owned = new_owned
        x = dst
    return x, owned
