"""Loss glue of the training scripts on the HIP path, forward and backward (each is one autograd node whose two
directions are HIP launches, wav2lip_amd/autograd.py):

  cosine_loss(a, v, y)   BCELoss(cosine_similarity(a, v).unsqueeze(1), y)   wav2lip_train.py:179-184,
                                                                            color_syncnet_train.py:133-138
  BCELoss / bce_mean     nn.BCELoss / F.binary_cross_entropy (mean)         models/wav2lip.py:171, hq_wav2lip_train.py:249,253
  L1Loss / l1_loss       nn.L1Loss (mean)                                   wav2lip_train.py:191,227
  get_sync_loss          lower half of the generated window -> frozen SyncNet -> cosine_loss vs ones
                                                                            wav2lip_train.py:192-198
"""
import torch

from . import autograd
from ._lib import check, current_stream, load, ptr
from .engine import require_cuda

syncnet_T = 5              # wav2lip_train.py:40
syncnet_mel_step_size = 16  # wav2lip_train.py:41


def bce_mean(p, y):
    return autograd.BCEMean.apply(p, y)


def l1_loss(a, b):
    return autograd.L1Mean.apply(a, b)


class BCELoss(torch.nn.Module):
    """drop-in for nn.BCELoss() (mean reduction, no weights)"""

    def forward(self, p, y):
        return bce_mean(p, y)


class L1Loss(torch.nn.Module):
    """drop-in for nn.L1Loss() (mean reduction)"""

    def forward(self, a, b):
        return l1_loss(a, b)


def cosine_similarity(a, v):
    require_cuda(a, "a")
    a = a.detach().contiguous().float()
    v = v.detach().contiguous().float()
    N, C = a.shape
    cos = torch.empty(N, device=a.device, dtype=torch.float32)
    check(load().w2l_cosine_bce(current_stream(), N, C, ptr(a), ptr(v), None, ptr(cos), None), "cosine")
    return cos


def cosine_loss(a, v, y):
    return autograd.CosineBCE.apply(a, v, y)


def get_sync_loss(syncnet, mel, g):
    """wav2lip_train.py:192-198: g (B,3,T,H,W) -> lower half, frames stacked on channels t-major -> SyncNet -> cosine
    loss against all-ones.  The slicing/cat are torch view ops (plumbing); the networks and the loss are HIP."""
    g = g[:, :, :, g.size(3) // 2:]
    g = torch.cat([g[:, :, i] for i in range(syncnet_T)], dim=1)
    a, v = syncnet(mel, g)
    y = torch.ones(g.size(0), 1, device=g.device, dtype=torch.float32)
    return cosine_loss(a, v, y)
