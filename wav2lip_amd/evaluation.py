"""LSE-D / LSE-C style lip-sync scoring (BASELINE.json's parity metric) on the HIP path.

The published numbers come from evaluation/scores_LSE/, which wraps the un-vendored `joonson/syncnet_python` model and
weights (`SyncNetModel.S`, data/syncnet_v2.model): not computable offline.  The scoring ARITHMETIC is in the
reference though (SyncNetInstance_calc_scores.py:19-31,95-150), and it only needs two embedding streams.  This module runs
that arithmetic on the embeddings of the in-tree expert `SyncNet_color` (the network the generator is trained against):

    per video frame v: face window = frames v..v+4, lower halves, stacked on channels (wav2lip_train.py:192-195 layout)
                       mel window  = 16 columns from int(80 * v / fps)          (wav2lip_train.py:80)
    dists[i][j] = || face_emb[i] - pad(audio_emb)[i+j] + 1e-6 ||, j over 2*vshift+1 offsets     (calc_pdist :19-31)
    mdist = mean_i dists;  LSE-D = min_j mdist;  LSE-C = median(mdist) - min;  offset = vshift - argmin   (:131-137)

With identical weights on both sides, equal scores for the engine's frames and the CPU path's frames is the stand-in for
"LSE-D / LSE-C parity"; the absolute values are not comparable to the paper's (different scorer network).
"""
import numpy as np
import torch

from ._lib import check, current_stream, load, ptr

syncnet_T = 5
mel_step = 16


def sync_windows(frames_u8, mel, fps=25.):
    """frames_u8: torch uint8 [T,96,96,3] generated crops (device); mel: torch float32 [80,Tm] (device).
    Returns (faces [n,15,48,96] float32 in [0,1], mels [n,1,80,16]) for every frame v with a full 5-frame and 16-column
    window."""
    T = frames_u8.shape[0]
    H = frames_u8.shape[1]
    x = frames_u8[:, H // 2:].permute(0, 3, 1, 2).float() / 255.           # [T,3,48,96]
    faces, mels = [], []
    for v in range(0, T - syncnet_T + 1):
        s = int(80. * (v / float(fps)))
        if s + mel_step > mel.shape[1]:
            break
        faces.append(x[v:v + syncnet_T].reshape(3 * syncnet_T, x.shape[2], x.shape[3]))
        mels.append(mel[:, s:s + mel_step].unsqueeze(0))
    if not faces:
        raise ValueError("clip too short for one 5-frame / 16-column window")
    return torch.stack(faces).contiguous(), torch.stack(mels).contiguous()


def lse_from_embeddings(face_emb, audio_emb, vshift=15):
    """SyncNetInstance_calc_scores.py:129-137 on two [n,C] embedding streams -> (offset, LSE-C, LSE-D, mdist [2*vshift+1])"""
    f1 = face_emb.contiguous().float()
    f2 = audio_emb.contiguous().float()
    n, C = f1.shape
    win = 2 * vshift + 1
    d = torch.empty((n, win), device=f1.device, dtype=torch.float32)
    check(load().w2l_shifted_pdist(current_stream(), n, C, vshift, ptr(f1), ptr(f2), ptr(d)), "shifted_pdist")
    mdist = d.mean(dim=0)
    minval, minidx = torch.min(mdist, 0)
    conf = torch.median(mdist) - minval
    return int(vshift - int(minidx)), float(conf), float(minval), mdist


@torch.no_grad()
def lse_like(syncnet, frames_u8, mel, fps=25., vshift=15, batch_size=64):
    """scores of a generated clip under the in-tree SyncNet_color (eval mode): dict(offset, lse_c, lse_d, n)"""
    was_training = syncnet.training
    syncnet.eval()
    try:
        faces, mels = sync_windows(frames_u8, mel, fps)
        a_all, v_all = [], []
        for lo in range(0, faces.shape[0], batch_size):
            a, v = syncnet(mels[lo:lo + batch_size], faces[lo:lo + batch_size])
            a_all.append(a)
            v_all.append(v)
        offset, conf, minval, mdist = lse_from_embeddings(torch.cat(v_all), torch.cat(a_all), vshift)
    finally:
        syncnet.train(was_training)
    return dict(offset=offset, lse_c=conf, lse_d=minval, n=int(faces.shape[0]), mdist=mdist.cpu().numpy())
