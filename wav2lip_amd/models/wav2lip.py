"""`Wav2Lip` (generator) and `Wav2Lip_disc_qual` with the reference's API and state-dict keys
(models/wav2lip.py:8-125, :127-184), executed as a static plan of fused HIP launches.

Architecture tables: each row is (kind, cin, cout, kernel, stride, padding, extra) where kind is
'c' Conv2d, 'r' residual Conv2d (3x3 s1 p1, cin == cout), 't' Conv2dTranspose (extra = output_padding),
'n' nonorm_Conv2d.  The tables restate the layer stacks of the reference constructors.
"""
import torch
from torch import nn

from .. import autograd, engine
from .._lib import ACT_NONE, ACT_SIGMOID, check, current_stream, load, ptr
from .conv import Conv2d, Conv2dTranspose, HeadFusedBlock, PlainConv, nonorm_Conv2d


def _res(c, n=1):
    return [("r", c, c, 3, 1, 1, 0)] * n


def _down(cin, cout, nres, stride=2):
    return [("c", cin, cout, 3, stride, 1, 0)] + _res(cout, nres)


def _up(cin, cout, nres):
    return [("t", cin, cout, 3, 2, 1, 1)] + _res(cout, nres)


FACE_ENCODER = [                                   # models/wav2lip.py:12-36
    [("c", 6, 16, 7, 1, 3, 0)],                    # 96x96
    _down(16, 32, 2),                              # 48x48
    _down(32, 64, 3),                              # 24x24
    _down(64, 128, 2),                             # 12x12
    _down(128, 256, 2),                            # 6x6
    _down(256, 512, 1),                            # 3x3
    [("c", 512, 512, 3, 1, 0, 0), ("c", 512, 512, 1, 1, 0, 0)],  # 1x1
]


def audio_encoder_rows(n_res256):
    """models/wav2lip.py:38-55 (one res256 block) and models/syncnet.py:35-53 (two)"""
    return ([("c", 1, 32, 3, 1, 1, 0)] + _res(32, 2)
            + _down(32, 64, 2, (3, 1)) + _down(64, 128, 2, 3) + _down(128, 256, n_res256, (3, 2))
            + [("c", 256, 512, 3, 1, 0, 0), ("c", 512, 512, 1, 1, 0, 0)])


FACE_DECODER = [                                   # models/wav2lip.py:57-81
    [("c", 512, 512, 1, 1, 0, 0)],
    [("t", 1024, 512, 3, 1, 0, 0)] + _res(512, 1),  # 3x3
    _up(1024, 512, 2),                             # 6x6
    _up(768, 384, 2),                              # 12x12
    _up(512, 256, 2),                              # 24x24
    _up(320, 128, 2),                              # 48x48
    _up(160, 64, 2),                               # 96x96
]

DISC_ENCODER = [                                   # models/wav2lip.py:131-150
    [("n", 3, 32, 7, 1, 3, 0)],                                            # 48x96
    [("n", 32, 64, 5, (1, 2), 2, 0), ("n", 64, 64, 5, 1, 2, 0)],           # 48x48
    [("n", 64, 128, 5, 2, 2, 0), ("n", 128, 128, 5, 1, 2, 0)],             # 24x24
    [("n", 128, 256, 5, 2, 2, 0), ("n", 256, 256, 5, 1, 2, 0)],            # 12x12
    [("n", 256, 512, 3, 2, 1, 0), ("n", 512, 512, 3, 1, 1, 0)],            # 6x6
    [("n", 512, 512, 3, 2, 1, 0), ("n", 512, 512, 3, 1, 1, 0)],            # 3x3
    [("n", 512, 512, 3, 1, 0, 0), ("n", 512, 512, 1, 1, 0, 0)],            # 1x1
]


def make_block(row):
    kind, cin, cout, k, s, p, extra = row
    if kind == "c":
        return Conv2d(cin, cout, kernel_size=k, stride=s, padding=p)
    if kind == "r":
        return Conv2d(cin, cout, kernel_size=k, stride=s, padding=p, residual=True)
    if kind == "t":
        return Conv2dTranspose(cin, cout, kernel_size=k, stride=s, padding=p, output_padding=extra)
    if kind == "n":
        return nonorm_Conv2d(cin, cout, kernel_size=k, stride=s, padding=p)
    raise ValueError(kind)


def make_stack(rows):
    return nn.Sequential(*[make_block(r) for r in rows])


def fold_time(audio_sequences, face_sequences):
    """5-D inputs: fold T into the batch, t-major (models/wav2lip.py:91-94)"""
    a = audio_sequences.transpose(0, 1).reshape((-1,) + tuple(audio_sequences.shape[2:]))
    f = face_sequences.permute(2, 0, 1, 3, 4).reshape((-1, face_sequences.shape[1]) + tuple(face_sequences.shape[3:]))
    return a, f


class _GeneratorGraph:
    """Buffers + launch plan of the generator for one (batch, height, width, device)."""

    def __init__(self, model, N, H, W, device):
        self.lib = load()
        self.N, self.H, self.W = N, H, W
        pool = engine.BufPool(device)
        pool_audio = engine.BufPool(device)     # the audio branch runs on its own stream: no scratch shared with the face branch
        plan = engine.Plan()
        enc = model.face_encoder_blocks
        dec = model.face_decoder_blocks
        self.x_in = engine.new_buf(N, H, W, 8, device, zero=True)         # 6 image channels + 2 zero pad
        self.mel_in = engine.new_buf(N, 80, 16, 4, device, zero=True)     # 1 mel channel + 3 zero pad
        # shapes of the encoder pyramid / decoder outputs -> one concat buffer per decoder stage
        enc_hw, h, w = [], H, W
        for blk in enc:
            for b in blk:
                h, w = b.fused().out_hw(h, w)
            enc_hw.append((h, w, blk[-1].fused().cout))
        nb = len(dec)
        cats = []
        for i, blk in enumerate(dec):
            eh, ew, ec = enc_hw[nb - 1 - i]
            dc = blk[-1].fused().cout
            cats.append((engine.new_buf(N, eh, ew, dc + ec, device), dc, ec))
        # face encoder: last layer of block i writes the skip slice of concat buffer nb-1-i
        x = engine.Act(self.x_in, 0, 8)
        for i, blk in enumerate(enc):
            buf, dc, ec = cats[nb - 1 - i]
            x, _ = engine.run_chain(plan, pool, "face_encoder_blocks.%d" % i, list(blk), x,
                                    engine.Act(buf, dc, ec))
        self.n_face = len(plan.records)
        # audio encoder
        a, a_buf = engine.run_chain(plan, pool_audio, "audio_encoder", list(model.audio_encoder),
                                    engine.Act(self.mel_in, 0, 4))
        self.n_audio = len(plan.records) - self.n_face
        if (a.H, a.W) != (1, 1):
            raise RuntimeError("audio encoder must reduce the mel window to 1x1, got %dx%d" % (a.H, a.W))
        # decoder: block i writes channels [0, dc) of concat buffer i, the next block reads all of it
        x = a
        for i, blk in enumerate(dec):
            buf, dc, ec = cats[i]
            x, _ = engine.run_chain(plan, pool, "face_decoder_blocks.%d" % i, list(blk), x,
                                    engine.Act(buf, 0, dc))
            x = engine.Act(buf, 0, dc + ec)
        # output block: conv 80->32 + BN + ReLU with the 1x1 32->3 + sigmoid head fused into its epilogue (one launch)
        self.out = engine.Act(engine.new_buf(N, H, W, 4, device, zero=True), 0, 3)
        engine.run_chain(plan, pool, "output_block", [model._head], x, self.out)
        if (x.H, x.W) != (H, W):
            raise RuntimeError("generator output is %dx%d for a %dx%d input" % (x.H, x.W, H, W))
        self.plan = plan
        self.scratch_bytes = pool.total_bytes + pool_audio.total_bytes
        # execution: the face and audio encoders are independent until the decoder's first concat and neither fills the chip
        # in its deep layers, so they run concurrently on two streams; `plan` (all launches in one sequence) stays the
        # object that is autotuned, profiled and counted
        self.parts = None
        self.hip_graph = None
        self.side = torch.cuda.Stream(device=device) if device.type == "cuda" and engine.TWO_STREAM_ENCODERS else None

    def load_nchw(self, audio, face):
        s = current_stream()
        N = self.N
        check(self.lib.w2l_nchw_to_nhwc(s, N, 6, self.H, self.W, ptr(face), ptr(self.x_in), 8, 8), "nchw_to_nhwc")
        check(self.lib.w2l_nchw_to_nhwc(s, N, 1, 80, 16, ptr(audio), ptr(self.mel_in), 4, 4), "nchw_to_nhwc")

    def _split(self):
        """three sub-plans over the same launches (face encoder | audio encoder | decoder + output) carrying the tuned configs"""
        cfg = self.plan.configs()
        bounds = [(0, self.n_face), (self.n_face, self.n_face + self.n_audio), (self.n_face + self.n_audio, len(cfg))]
        parts = []
        for lo, hi in bounds:
            p = engine.Plan()
            for i in range(lo, hi):
                p.add_raw(self.plan, i)
                p.set_config(i - lo, cfg[i][1], cfg[i][2])
            p.tuned = True
            parts.append(p)
        return parts

    def _launch(self):
        if self.side is None:
            self.plan.run()
            return
        face, audio, tail = self.parts
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)                 # the inputs were written on the main stream
        with torch.cuda.stream(self.side):
            audio.run()
        face.run()
        main.wait_stream(self.side)
        tail.run()

    def run(self):
        if not self.plan.tuned and engine.AUTOTUNE:
            self.plan.autotune()
        if self.side is not None and self.parts is None:
            self.parts = self._split()
        if not engine.HIP_GRAPHS:
            self._launch()
            return
        # the whole launch sequence (both streams, fork and join included) is captured once into a HIP graph and replayed:
        # at small batches the step is launch-bound (53 launches of a few microseconds each)
        if self.hip_graph is None:
            cap = torch.cuda.Stream(device=self.out.buf.device)     # capture needs a non-default stream; the warm-up runs on
            cap.wait_stream(torch.cuda.current_stream())            # the SAME stream so that its split-K scratch exists
            with torch.cuda.stream(cap):
                self._launch()
            cap.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=cap):
                self._launch()
            self.hip_graph = gr
        self.hip_graph.replay()

    def output_nchw(self):
        y = torch.empty((self.N, 3, self.H, self.W), device=self.out.buf.device, dtype=torch.float32)
        check(self.lib.w2l_nhwc_to_nchw(current_stream(), self.N, 3, self.H, self.W, self.out.ptr, self.out.cs,
                                        ptr(y)), "nhwc_to_nchw")
        return y


class Wav2Lip(nn.Module):
    MAX_PLAN_BATCH = 512      # frames per static plan: larger inference batches are chunked (see forward)

    def __init__(self):
        super().__init__()
        self.face_encoder_blocks = nn.ModuleList([make_stack(r) for r in FACE_ENCODER])
        self.audio_encoder = make_stack(audio_encoder_rows(1))
        self.face_decoder_blocks = nn.ModuleList([make_stack(r) for r in FACE_DECODER])
        self.output_block = nn.Sequential(Conv2d(80, 32, kernel_size=3, stride=1, padding=1),
                                          nn.Conv2d(32, 3, kernel_size=1, stride=1, padding=0),
                                          nn.Sigmoid())
        object.__setattr__(self, "_head", HeadFusedBlock(self.output_block[0], self.output_block[1], ACT_SIGMOID))
        self._graphs = {}
        object.__setattr__(self, "_train_graphs", autograd.GraphCache(autograd.build_generator))

    def graph(self, N, H=96, W=96, device=None, lane=0):
        """the static launch plan for batch N (built on first use, rebuilt if the weights changed); `lane` selects one of
        several independent buffer sets for batches in flight on different streams (inference.PipelinedRunner)"""
        device = device or next(self.parameters()).device
        ver = engine.param_version(self)
        key = (N, H, W, str(device), lane)
        g = self._graphs.get(key)
        if g is None or g[0] != ver:
            if any(v[0] != ver for v in self._graphs.values()):
                self._graphs.clear()
            g = (ver, _GeneratorGraph(self, N, H, W, torch.device(device)))
            if (H, W) == (96, 96):    # batch sizes the launch table does not hold: committed per-plan configurations (engine.py)
                engine.apply_plan_configs(g[1].plan, "generator_96", N)
            twin = next((v[1] for k, v in self._graphs.items() if k[:4] == key[:4] and v[1].plan.tuned), None)
            if twin is not None:      # another lane of the same geometry is already tuned: same launches, same configurations
                for i, (_, t_, k_) in enumerate(twin.plan.configs()):
                    g[1].plan.set_config(i, t_, k_)
                g[1].plan.tuned = True
            self._graphs[key] = g
        return g[1]

    def forward(self, audio_sequences, face_sequences):
        engine.require_cuda(face_sequences, "face_sequences")
        with engine.on_device_of(face_sequences, self):      # launches go to the current stream of the INPUT's device; refuses a device mismatch
            return self._forward_impl(audio_sequences, face_sequences)

    def _forward_impl(self, audio_sequences, face_sequences):
        engine.require_cuda(face_sequences, "face_sequences")
        engine.require_cuda(audio_sequences, "audio_sequences")
        B = audio_sequences.size(0)
        five_d = face_sequences.dim() > 4
        if five_d:
            audio_sequences, face_sequences = fold_time(audio_sequences, face_sequences)
        face = face_sequences.contiguous().float()
        audio = audio_sequences.contiguous().float()
        N, _, H, W = face.shape
        if autograd.needs_graph(self, (audio, face)):
            # training / gradient-recording call (wav2lip_train.py:220): one autograd node, HIP forward and backward
            out = autograd.run_graph(self._train_graphs, self, (N, H, W, str(face.device)), (N, H, W, face.device),
                                     (audio, face))[0]
        else:
            # one NHWC buffer must stay below 2 GiB (32-bit buffer-descriptor offsets): the 80-channel concat buffer at the
            # input resolution is the largest, so huge batches run as equal chunks of at most MAX_PLAN_BATCH frames
            cap = max(1, min(self.MAX_PLAN_BATCH, ((1 << 31) - 1) // (H * W * 80 * 4)))
            outs = []
            for lo in range(0, N, cap):
                n = min(cap, N - lo)
                g = self.graph(n, H, W, face.device)
                g.load_nchw(audio[lo:lo + n].contiguous(), face[lo:lo + n].contiguous())
                g.run()
                outs.append(g.output_nchw())
            out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        if five_d:  # (T*B, 3, H, W) -> (B, 3, T, H, W), models/wav2lip.py:118-120
            out = out.view(-1, B, 3, H, W).permute(1, 2, 0, 3, 4).contiguous()
        return out


class _DiscGraph:
    def __init__(self, model, N, H, W, device):
        self.lib = load()
        self.N, self.H, self.W = N, H, W
        pool = engine.BufPool(device)
        plan = engine.Plan()
        self.x_in = engine.new_buf(N, H, W, 4, device, zero=True)
        x = engine.Act(self.x_in, 0, 4)
        for i, blk in enumerate(model.face_encoder_blocks):
            x, owned = engine.run_chain(plan, pool, "face_encoder_blocks.%d" % i, list(blk), x)
        self.pred = engine.Act(engine.new_buf(N, x.H, x.W, 1, device), 0, 1)
        engine.run_chain(plan, pool, "binary_pred", [model._head], x, self.pred)
        self.plan = plan


class Wav2Lip_disc_qual(nn.Module):
    def __init__(self):
        super().__init__()
        self.face_encoder_blocks = nn.ModuleList([make_stack(r) for r in DISC_ENCODER])
        self.binary_pred = nn.Sequential(nn.Conv2d(512, 1, kernel_size=1, stride=1, padding=0), nn.Sigmoid())
        self.label_noise = .0
        object.__setattr__(self, "_head", PlainConv(self.binary_pred[0], ACT_SIGMOID))
        self._graphs = {}
        object.__setattr__(self, "_train_graphs", autograd.GraphCache(autograd.build_disc))

    def get_lower_half(self, face_sequences):
        return face_sequences[:, :, face_sequences.size(2) // 2:]

    def to_2d(self, face_sequences):
        # (B, C, T, H, W) -> (T*B, C, H, W), t-major (models/wav2lip.py:158-161)
        return face_sequences.permute(2, 0, 1, 3, 4).reshape(
            (-1, face_sequences.shape[1]) + tuple(face_sequences.shape[3:]))

    def _predict(self, face_sequences):
        engine.require_cuda(face_sequences, "face_sequences")
        with engine.on_device_of(face_sequences, self):      # launches go to the current stream of the INPUT's device; refuses a device mismatch
            return self._predict_impl(face_sequences)

    def _predict_impl(self, face_sequences):
        engine.require_cuda(face_sequences, "face_sequences")
        x = self.get_lower_half(self.to_2d(face_sequences)).contiguous().float()
        N, C_, H, W = x.shape
        if autograd.needs_graph(self, (x,)):
            out = autograd.run_graph(self._train_graphs, self, (N, H, W, str(x.device)), (N, H, W, x.device), (x,))[0]
            return out.reshape(N, -1)
        ver = engine.param_version(self)
        key = (N, H, W, str(x.device))
        g = self._graphs.get(key)
        if g is None or g[0] != ver:
            self._graphs.clear()
            g = (ver, _DiscGraph(self, N, H, W, x.device))
            self._graphs[key] = g
        g = g[1]
        check(g.lib.w2l_nchw_to_nhwc(current_stream(), N, C_, H, W, ptr(x), ptr(g.x_in), 4, 4), "nchw_to_nhwc")
        g.plan.run()
        return g.pred.buf.reshape(N, -1).clone()

    def perceptual_forward(self, false_face_sequences):
        """BCE(D(fake), 1), models/wav2lip.py:163-174"""
        from ..losses import bce_mean
        pred = self._predict(false_face_sequences)
        return bce_mean(pred, torch.ones_like(pred))

    def forward(self, face_sequences):
        return self._predict(face_sequences)
