"""Loss glue of the training scripts on the HIP path (forward values):
cosine_loss = BCELoss(cosine_similarity(a, v).unsqueeze(1), y)  (wav2lip_train.py:179-184,
color_syncnet_train.py:133-138) and F.binary_cross_entropy (models/wav2lip.py:171)."""
import torch

from ._lib import check, current_stream, load, ptr
from .engine import require_cuda


def bce_mean(p, y):
    require_cuda(p, "p")
    p = p.contiguous().float().view(-1)
    y = y.contiguous().float().view(-1).to(p.device)
    out = torch.empty(1, device=p.device, dtype=torch.float32)
    check(load().w2l_bce_mean(current_stream(), p.numel(), ptr(p), ptr(y), ptr(out)), "bce_mean")
    return out[0]


def cosine_similarity(a, v):
    require_cuda(a, "a")
    a = a.contiguous().float()
    v = v.contiguous().float()
    N, C = a.shape
    cos = torch.empty(N, device=a.device, dtype=torch.float32)
    check(load().w2l_cosine_bce(current_stream(), N, C, ptr(a), ptr(v), None, ptr(cos), None), "cosine")
    return cos


def cosine_loss(a, v, y):
    require_cuda(a, "a")
    a = a.contiguous().float()
    v = v.contiguous().float()
    y = y.contiguous().float().view(-1).to(a.device)
    N, C = a.shape
    cos = torch.empty(N, device=a.device, dtype=torch.float32)
    loss = torch.empty(1, device=a.device, dtype=torch.float32)
    check(load().w2l_cosine_bce(current_stream(), N, C, ptr(a), ptr(v), ptr(y), ptr(cos), ptr(loss)), "cosine_bce")
    return loss[0]
