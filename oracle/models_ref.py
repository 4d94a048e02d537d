"""Oracle for the three networks: plain torch CPU fp32, functional, driven directly by a reference-format
state_dict.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows the reference: block semantics models/conv.py:14-19,28-31,41-44; generator models/wav2lip.py:87-125;
discriminator models/wav2lip.py:155-184; SyncNet models/syncnet.py:55-66.  Layer geometry (stride / padding /
output_padding / residual) is not stored in a state_dict, so it is restated here as compact strings:
    "k<K>[s<S>|s<Sh>x<Sw>]p<P>[r][T[o<OP>]]"    r = residual, T = transposed conv
"""
import re

import torch
import torch.nn.functional as F

GEN_FACE_ENC = [["k7p3"], ["k3s2p1", "k3p1r", "k3p1r"], ["k3s2p1", "k3p1r", "k3p1r", "k3p1r"],
                ["k3s2p1", "k3p1r", "k3p1r"], ["k3s2p1", "k3p1r", "k3p1r"], ["k3s2p1", "k3p1r"], ["k3p0", "k1p0"]]
GEN_AUDIO_ENC = ["k3p1", "k3p1r", "k3p1r", "k3s3x1p1", "k3p1r", "k3p1r", "k3s3p1", "k3p1r", "k3p1r",
                 "k3s3x2p1", "k3p1r", "k3p0", "k1p0"]
GEN_FACE_DEC = [["k1p0"], ["k3p0T", "k3p1r"], ["k3s2p1To1", "k3p1r", "k3p1r"], ["k3s2p1To1", "k3p1r", "k3p1r"],
                ["k3s2p1To1", "k3p1r", "k3p1r"], ["k3s2p1To1", "k3p1r", "k3p1r"], ["k3s2p1To1", "k3p1r", "k3p1r"]]
SYNC_FACE_ENC = ["k7p3", "k5s1x2p1", "k3p1r", "k3p1r", "k3s2p1", "k3p1r", "k3p1r", "k3p1r", "k3s2p1", "k3p1r",
                 "k3p1r", "k3s2p1", "k3p1r", "k3p1r", "k3s2p1", "k3p0", "k1p0"]
SYNC_AUDIO_ENC = ["k3p1", "k3p1r", "k3p1r", "k3s3x1p1", "k3p1r", "k3p1r", "k3s3p1", "k3p1r", "k3p1r",
                  "k3s3x2p1", "k3p1r", "k3p1r", "k3p0", "k1p0"]
DISC_ENC = [["k7p3"], ["k5s1x2p2", "k5p2"], ["k5s2p2", "k5p2"], ["k5s2p2", "k5p2"], ["k3s2p1", "k3p1"],
            ["k3s2p1", "k3p1"], ["k3p0", "k1p0"]]

_GEOM = re.compile(r"k(\d+)(?:s(\d+)(?:x(\d+))?)?p(\d+)(r)?(T)?(?:o(\d+))?$")


def parse_geom(s):
    m = _GEOM.match(s)
    if not m:
        raise ValueError(s)
    sh = int(m.group(2) or 1)
    sw = int(m.group(3) or sh)
    return dict(stride=(sh, sw), padding=int(m.group(4)), residual=bool(m.group(5)), transposed=bool(m.group(6)),
                output_padding=int(m.group(7) or 0))


def block(x, sd, prefix, geom, norm=True, training=False):
    """one Conv2d / Conv2dTranspose / nonorm_Conv2d block of models/conv.py; training=True = BatchNorm on batch
    statistics, updating sd's running_mean / running_var / num_batches_tracked in place as nn.BatchNorm2d does"""
    g = parse_geom(geom)
    w, b = sd[prefix + ".conv_block.0.weight"], sd[prefix + ".conv_block.0.bias"]
    if g["transposed"]:
        y = F.conv_transpose2d(x, w, b, stride=g["stride"], padding=g["padding"], output_padding=g["output_padding"])
    else:
        y = F.conv2d(x, w, b, stride=g["stride"], padding=g["padding"])
    if not norm:
        return F.leaky_relu(y, 0.01)
    p = prefix + ".conv_block.1."
    y = F.batch_norm(y, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                     training=training, momentum=0.1, eps=1e-5)
    if training and (p + "num_batches_tracked") in sd:
        sd[p + "num_batches_tracked"] += 1
    if g["residual"]:
        y = y + x
    return F.relu(y)


def _seq(x, sd, prefix, geoms, norm=True, training=False):
    for j, g in enumerate(geoms):
        x = block(x, sd, "%s.%d" % (prefix, j), g, norm, training)
    return x


@torch.no_grad()
def wav2lip_forward(sd, audio_sequences, face_sequences):
    return wav2lip_graph(sd, audio_sequences, face_sequences, training=False)


@torch.no_grad()
def syncnet_forward(sd, audio_sequences, face_sequences):
    return syncnet_graph(sd, audio_sequences, face_sequences, training=False)


@torch.no_grad()
def disc_forward(sd, face_sequences):
    return disc_graph(sd, face_sequences)


# The *_graph functions below build a differentiable torch graph (autograd on the sd tensors that require grad): they are
# the gradient oracle for the training path.  training=True puts every BatchNorm in batch-statistics mode.
def wav2lip_graph(sd, audio_sequences, face_sequences, training=False):
    B = audio_sequences.size(0)
    five_d = face_sequences.dim() > 4
    if five_d:
        audio_sequences = torch.cat([audio_sequences[:, i] for i in range(audio_sequences.size(1))], dim=0)
        face_sequences = torch.cat([face_sequences[:, :, i] for i in range(face_sequences.size(2))], dim=0)
    emb = _seq(audio_sequences, sd, "audio_encoder", GEN_AUDIO_ENC, True, training)
    feats, x = [], face_sequences
    for i, geoms in enumerate(GEN_FACE_ENC):
        x = _seq(x, sd, "face_encoder_blocks.%d" % i, geoms, True, training)
        feats.append(x)
    x = emb
    for i, geoms in enumerate(GEN_FACE_DEC):
        x = _seq(x, sd, "face_decoder_blocks.%d" % i, geoms, True, training)
        x = torch.cat((x, feats.pop()), dim=1)
    x = block(x, sd, "output_block.0", "k3p1", True, training)
    x = torch.sigmoid(F.conv2d(x, sd["output_block.1.weight"], sd["output_block.1.bias"]))
    if five_d:
        x = torch.stack(torch.split(x, B, dim=0), dim=2)
    return x


def syncnet_graph(sd, audio_sequences, face_sequences, training=False):
    f = _seq(face_sequences, sd, "face_encoder", SYNC_FACE_ENC, True, training)
    a = _seq(audio_sequences, sd, "audio_encoder", SYNC_AUDIO_ENC, True, training)
    a = F.normalize(a.view(a.size(0), -1), p=2, dim=1)
    f = F.normalize(f.view(f.size(0), -1), p=2, dim=1)
    return a, f


def disc_graph(sd, face_sequences):
    x = torch.cat([face_sequences[:, :, i] for i in range(face_sequences.size(2))], dim=0)
    x = x[:, :, x.size(2) // 2:]
    for i, geoms in enumerate(DISC_ENC):
        x = _seq(x, sd, "face_encoder_blocks.%d" % i, geoms, norm=False)
    x = torch.sigmoid(F.conv2d(x, sd["binary_pred.0.weight"], sd["binary_pred.0.bias"]))
    return x.view(len(x), -1)


def cosine_loss(a, v, y):
    """wav2lip_train.py:179-184"""
    d = F.cosine_similarity(a, v)
    return F.binary_cross_entropy(d.unsqueeze(1), y)


def get_sync_loss(sd_sync, mel, g, training=True):
    """wav2lip_train.py:192-198 (the reference never puts its frozen SyncNet in eval mode: training=True)"""
    g = g[:, :, :, g.size(3) // 2:]
    g = torch.cat([g[:, :, i] for i in range(5)], dim=1)
    a, v = syncnet_graph(sd_sync, mel, g, training)
    return cosine_loss(a, v, torch.ones(g.size(0), 1))


def perceptual_loss(sd_disc, g):
    """models/wav2lip.py:163-174"""
    p = disc_graph(sd_disc, g)
    return F.binary_cross_entropy(p, torch.ones(len(p), 1))
