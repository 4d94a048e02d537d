"""`SyncNet_color`, the expert lip-sync discriminator, with the reference's API and state-dict keys
(models/syncnet.py:7-66): face (B,15,48,96) and mel (B,1,80,16) -> two L2-normalised 512-d embeddings."""
import torch
from torch import nn

from .. import autograd, engine
from .._lib import check, current_stream, load, ptr
from .wav2lip import _down, _res, audio_encoder_rows, make_stack

SYNC_FACE_ENCODER = (                              # models/syncnet.py:11-33
    [("c", 15, 32, 7, 1, 3, 0),
     ("c", 32, 64, 5, (1, 2), 1, 0)] + _res(64, 2)
    + _down(64, 128, 3) + _down(128, 256, 2) + _down(256, 512, 2)
    + [("c", 512, 512, 3, 2, 1, 0), ("c", 512, 512, 3, 1, 0, 0), ("c", 512, 512, 1, 1, 0, 0)])


class _SyncGraph:
    def __init__(self, model, N, H, W, device):
        self.lib = load()
        self.N, self.H, self.W = N, H, W
        pool = engine.BufPool(device)
        plan = engine.Plan()
        self.face_in = engine.new_buf(N, H, W, 16, device, zero=True)   # 15 channels + 1 zero pad
        self.mel_in = engine.new_buf(N, 80, 16, 4, device, zero=True)
        self.face_out, _ = engine.run_chain(plan, pool, "face_encoder", list(model.face_encoder),
                                            engine.Act(self.face_in, 0, 16))
        self.audio_out, _ = engine.run_chain(plan, pool, "audio_encoder", list(model.audio_encoder),
                                             engine.Act(self.mel_in, 0, 4))
        for o in (self.face_out, self.audio_out):
            if (o.H, o.W) != (1, 1):
                raise RuntimeError("SyncNet encoders must end at 1x1, got %dx%d" % (o.H, o.W))
        self.plan = plan


class SyncNet_color(nn.Module):
    def __init__(self):
        super().__init__()
        self.face_encoder = make_stack(SYNC_FACE_ENCODER)
        self.audio_encoder = make_stack(audio_encoder_rows(2))
        self._graphs = {}
        object.__setattr__(self, "_train_graphs", autograd.GraphCache(autograd.build_syncnet))

    def forward(self, audio_sequences, face_sequences):
        engine.require_cuda(face_sequences, "face_sequences")
        with engine.on_device_of(face_sequences, self):      # launches go to the current stream of the INPUT's device; refuses a device mismatch
            return self._forward_impl(audio_sequences, face_sequences)

    def _forward_impl(self, audio_sequences, face_sequences):
        engine.require_cuda(face_sequences, "face_sequences")
        engine.require_cuda(audio_sequences, "audio_sequences")
        face = face_sequences.contiguous().float()
        audio = audio_sequences.contiguous().float()
        N, C_, H, W = face.shape
        if autograd.needs_graph(self, (audio, face)):
            # train mode (color_syncnet_train.py:150-158) or the frozen expert inside the generator's loss
            # (wav2lip_train.py:187-198, which never calls .eval(): BN runs on batch statistics there too)
            a, v = autograd.run_graph(self._train_graphs, self, (N, H, W, str(face.device)), (N, H, W, face.device),
                                      (audio, face))
            return autograd.L2NormRows.apply(a.reshape(N, -1)), autograd.L2NormRows.apply(v.reshape(N, -1))
        ver = engine.param_version(self)
        key = (N, H, W, str(face.device))
        g = self._graphs.get(key)
        if g is None or g[0] != ver:
            self._graphs.clear()
            g = (ver, _SyncGraph(self, N, H, W, face.device))
            self._graphs[key] = g
        g = g[1]
        s = current_stream()
        check(g.lib.w2l_nchw_to_nhwc(s, N, C_, H, W, ptr(face), ptr(g.face_in), 16, 16), "nchw_to_nhwc")
        check(g.lib.w2l_nchw_to_nhwc(s, N, 1, 80, 16, ptr(audio), ptr(g.mel_in), 4, 4), "nchw_to_nhwc")
        g.plan.run()
        a = torch.empty((N, 512), device=face.device, dtype=torch.float32)
        v = torch.empty((N, 512), device=face.device, dtype=torch.float32)
        check(g.lib.w2l_l2norm_rows(s, N, 512, g.audio_out.ptr, g.audio_out.cs, ptr(a)), "l2norm_rows")
        check(g.lib.w2l_l2norm_rows(s, N, 512, g.face_out.ptr, g.face_out.cs, ptr(v)), "l2norm_rows")
        return a, v
