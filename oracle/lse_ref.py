"""Oracle for the LSE-style scoring arithmetic (evaluation/scores_LSE/SyncNetInstance_calc_scores.py:19-31,129-137) on the
in-tree SyncNet_color embeddings.  torch CPU, the reference's own expressions.  TEST INFRASTRUCTURE.
The scorer NETWORK of the published metric (joonson/syncnet_python) is un-vendored: parity of the metric itself is unpinned;
this pins the scoring arithmetic, which IS in the reference."""
import torch

from . import models_ref


def calc_pdist(feat1, feat2, vshift=10):
    """:19-31 verbatim in behaviour"""
    win_size = vshift * 2 + 1
    feat2p = torch.nn.functional.pad(feat2, (0, 0, vshift, vshift))
    dists = []
    for i in range(0, len(feat1)):
        dists.append(torch.nn.functional.pairwise_distance(feat1[[i], :].repeat(win_size, 1), feat2p[i:i + win_size, :]))
    return dists


def scores(im_feat, cc_feat, vshift=15):
    """:129-137 -> (offset, conf, minval, mdist)"""
    dists = calc_pdist(im_feat, cc_feat, vshift=vshift)
    mdist = torch.mean(torch.stack(dists, 1), 1)
    minval, minidx = torch.min(mdist, 0)
    offset = vshift - minidx
    conf = torch.median(mdist) - minval
    return int(offset), float(conf), float(minval), mdist


def lse_like(sd_sync, frames_u8, mel, fps=25., vshift=15):
    """frames_u8 numpy [T,96,96,3]; mel numpy [80,Tm]; windows as wav2lip_train.py:80,192-195"""
    import numpy as np
    T, H = frames_u8.shape[0], frames_u8.shape[1]
    x = torch.from_numpy(frames_u8[:, H // 2:].transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255.))
    faces, mels = [], []
    for v in range(0, T - 5 + 1):
        s = int(80. * (v / float(fps)))
        if s + 16 > mel.shape[1]:
            break
        faces.append(x[v:v + 5].reshape(15, x.shape[2], x.shape[3]))
        mels.append(torch.from_numpy(mel[:, s:s + 16].copy()).unsqueeze(0))
    a, v = models_ref.syncnet_forward(sd_sync, torch.stack(mels), torch.stack(faces))
    return scores(v, a, vshift)
