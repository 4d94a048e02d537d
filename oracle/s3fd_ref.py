"""Oracle for the S3FD face detector path: torch CPU fp32 network driven by a reference-format
state_dict + numpy restatement of the box decode / NMS / batch post-processing.  TEST INFRASTRUCTURE.

Follows face_detection/detection/sfd/net_s3fd.py:6-129 (network), detect.py:55-91 (`batch_detect`), bbox.py:44-64 (`nms`),
:91-129 (`decode`), sfd_detector.py:39-45 (`detect_from_batch`), api.py:61-77 (`get_detections_for_batch`),
inference.py:59-104 (`get_smoothened_boxes`, padding).  Pinned by tests/golden/make_golden_s3fd.py, which runs the REAL
reference modules (with a stub `cv2` module: none of these functions touches it) on seeded weights and inputs.
"""
import numpy as np
import torch
import torch.nn.functional as F

# (name, cin, cout, kernel, stride, padding) in state-dict order (net_s3fd.py:25-66)
CONVS = [("conv1_1", 3, 64, 3, 1, 1), ("conv1_2", 64, 64, 3, 1, 1), ("conv2_1", 64, 128, 3, 1, 1), ("conv2_2", 128, 128, 3, 1, 1),
         ("conv3_1", 128, 256, 3, 1, 1), ("conv3_2", 256, 256, 3, 1, 1), ("conv3_3", 256, 256, 3, 1, 1),
         ("conv4_1", 256, 512, 3, 1, 1), ("conv4_2", 512, 512, 3, 1, 1), ("conv4_3", 512, 512, 3, 1, 1),
         ("conv5_1", 512, 512, 3, 1, 1), ("conv5_2", 512, 512, 3, 1, 1), ("conv5_3", 512, 512, 3, 1, 1),
         ("fc6", 512, 1024, 3, 1, 3), ("fc7", 1024, 1024, 1, 1, 0), ("conv6_1", 1024, 256, 1, 1, 0), ("conv6_2", 256, 512, 3, 2, 1),
         ("conv7_1", 512, 128, 1, 1, 0), ("conv7_2", 128, 256, 3, 2, 1)]
NORMS = [("conv3_3_norm", 256, 10.), ("conv4_3_norm", 512, 8.), ("conv5_3_norm", 512, 5.)]
HEADS = [("conv3_3_norm", 256, 4), ("conv4_3_norm", 512, 2), ("conv5_3_norm", 512, 2), ("fc7", 1024, 2), ("conv6_2", 512, 2),
         ("conv7_2", 256, 2)]      # (source, cin, conf channels); loc always 4 channels


def _c(x, sd, name, stride=1, pad=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=pad)


def l2norm(x, w):
    norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10
    return x / norm * w.view(1, -1, 1, 1)


@torch.no_grad()
def s3fd_forward(sd, x):
    """net_s3fd.py:68-129 -> [cls1, reg1, ..., cls6, reg6]"""
    geo = {n: (s, p) for n, _, _, _, s, p in CONVS}
    r = lambda h, n: F.relu(_c(h, sd, n, *geo[n]))
    h = r(r(x, "conv1_1"), "conv1_2")
    h = F.max_pool2d(h, 2, 2)
    h = r(r(h, "conv2_1"), "conv2_2")
    h = F.max_pool2d(h, 2, 2)
    h = r(r(r(h, "conv3_1"), "conv3_2"), "conv3_3")
    f3 = h
    h = F.max_pool2d(h, 2, 2)
    h = r(r(r(h, "conv4_1"), "conv4_2"), "conv4_3")
    f4 = h
    h = F.max_pool2d(h, 2, 2)
    h = r(r(r(h, "conv5_1"), "conv5_2"), "conv5_3")
    f5 = h
    h = F.max_pool2d(h, 2, 2)
    h = r(r(h, "fc6"), "fc7")
    ffc7 = h
    h = r(r(h, "conv6_1"), "conv6_2")
    f6 = h
    h = r(r(h, "conv7_1"), "conv7_2")
    f7 = h
    feats = {"conv3_3_norm": l2norm(f3, sd["conv3_3_norm.weight"]), "conv4_3_norm": l2norm(f4, sd["conv4_3_norm.weight"]),
             "conv5_3_norm": l2norm(f5, sd["conv5_3_norm.weight"]), "fc7": ffc7, "conv6_2": f6, "conv7_2": f7}
    out = []
    for src, _, _ in HEADS:
        out.append(_c(feats[src], sd, src + "_mbox_conf"))
        out.append(_c(feats[src], sd, src + "_mbox_loc"))
    chunk = torch.chunk(out[0], 4, 1)
    bmax = torch.max(torch.max(chunk[0], chunk[1]), chunk[2])
    out[0] = torch.cat([bmax, chunk[3]], dim=1)
    return out


def preprocess(images_bgr):
    """api.py:62 + detect.py:57-63: uint8 [B,H,W,3] BGR -> float32 [B,3,H,W] RGB minus (104,117,123)"""
    imgs = images_bgr[..., ::-1] - np.array([104, 117, 123])
    return torch.from_numpy(imgs.transpose(0, 3, 1, 2).copy()).float()


def dense_boxes(olist):
    """per level: [B, FH*FW, 5] = (x1, y1, x2, y2, score) for EVERY position (detect.py:66-84 without the 0.05 gate)"""
    levels = []
    for i in range(len(olist) // 2):
        ocls = F.softmax(olist[2 * i], dim=1)
        oreg = olist[2 * i + 1]
        B, _, FH, FW = ocls.shape
        stride = 2 ** (i + 2)
        ys, xs = torch.meshgrid(torch.arange(FH), torch.arange(FW), indexing="ij")
        axc = stride / 2 + xs.float() * stride
        ayc = stride / 2 + ys.float() * stride
        priors = torch.stack([axc, ayc, torch.full_like(axc, stride * 4.), torch.full_like(axc, stride * 4.)], -1).view(1, -1, 4)
        loc = oreg.permute(0, 2, 3, 1).reshape(B, -1, 4)
        boxes = torch.cat((priors[:, :, :2] + loc[:, :, :2] * 0.1 * priors[:, :, 2:],
                           priors[:, :, 2:] * torch.exp(loc[:, :, 2:] * 0.2)), 2)
        boxes[:, :, :2] -= boxes[:, :, 2:] / 2
        boxes[:, :, 2:] += boxes[:, :, :2]
        score = ocls[:, 1].reshape(B, -1, 1)
        levels.append(torch.cat([boxes, score], 2).numpy())
    return levels


def nms(dets, thresh):
    """bbox.py:44-64"""
    if 0 == len(dets):
        return []
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1, yy1 = np.maximum(x1[i], x1[order[1:]]), np.maximum(y1[i], y1[order[1:]])
        xx2, yy2 = np.minimum(x2[i], x2[order[1:]]), np.minimum(y2[i], y2[order[1:]])
        w, h = np.maximum(0.0, xx2 - xx1 + 1), np.maximum(0.0, yy2 - yy1 + 1)
        ovr = w * h / (areas[i] + areas[order[1:]] - w * h)
        inds = np.where(ovr <= thresh)[0]
        order = order[inds + 1]
    return keep


def detections(levels, gate=0.05, nms_thresh=0.3, keep_thresh=0.5):
    """sfd_detector.py:39-45 on the dense boxes: per image, candidates above `gate`, NMS, then score > keep_thresh.
    (batch_detect gates a position when ANY image of the batch exceeds 0.05 there and may list it more than once; boxes that
    end up above 0.5 after NMS are the same either way, see DESIGN.md.)"""
    B = levels[0].shape[0]
    out = []
    for b in range(B):
        d = np.concatenate([lv[b] for lv in levels], 0)
        d = d[d[:, 4] > gate]
        keep = nms(d, nms_thresh)
        d = d[keep]
        out.append([x for x in d if x[-1] > keep_thresh])
    return out


def rects(det_lists):
    """api.py:66-77: first (highest-score) box, clipped at 0, truncated to int; None when there is no face"""
    res = []
    for d in det_lists:
        if len(d) == 0:
            res.append(None)
            continue
        d0 = np.clip(d[0], 0, None)
        x1, y1, x2, y2 = map(int, d0[:-1])
        res.append((x1, y1, x2, y2))
    return res


def get_smoothened_boxes(boxes, T):
    """inference.py:59-66, in place on the caller's array: face_detect passes an INTEGER array (np.array of int lists,
    inference.py:101), so every mean is truncated toward zero on assignment and later windows see the truncated values"""
    for i in range(len(boxes)):
        if i + T > len(boxes):
            window = boxes[len(boxes) - T:]
        else:
            window = boxes[i: i + T]
        boxes[i] = np.mean(window, axis=0)
    return boxes
