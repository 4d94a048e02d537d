"""Generates the committed S3FD golden fixture from the REAL reference face detector (network + batch_detect + nms +
detect_from_batch + get_detections_for_batch arithmetic), CPU fp32.  Runs only in the build container.

    python tests/golden/make_golden_s3fd.py

The reference modules import cv2 at the top but the functions used here never call it: a stub module stands in.
Weights are seeded He-style (no s3fd.pth offline); the conf heads get a positive foreground bias on a few channels so that
the fixture contains real detections (scores above 0.5) as well as the dense maps.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import s3fd_ref  # noqa: E402


from wav2lip_amd.synthetic import s3fd_frames as images  # noqa: E402,F401
from wav2lip_amd.synthetic import s3fd_state_dict as seeded_state_dict  # noqa: E402,F401


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    from face_detection.detection.sfd import bbox as ref_bbox
    from face_detection.detection.sfd import detect as ref_detect
    from face_detection.detection.sfd.net_s3fd import s3fd
    assert ref_detect.__file__.startswith(REF)
    torch.set_num_threads(8)
    net = s3fd().eval()
    sd = seeded_state_dict()
    assert sorted(sd) == sorted(net.state_dict()), set(sd) ^ set(net.state_dict())
    net.load_state_dict(sd)
    img = images()
    x = s3fd_ref.preprocess(img)
    with torch.no_grad():
        olist = net(x)
    oo = s3fd_ref.s3fd_forward(sd, x)
    for a, b in zip(olist, oo):
        assert torch.equal(a, b), "oracle/s3fd_ref.py does not reproduce the reference network bit-for-bit"
    out = {"out%d" % i: o.numpy() for i, o in enumerate(olist)}
    # the reference's own post-processing: batch_detect (api.py flips BGR->RGB before it) + nms + thresholds
    bboxlists = ref_detect.batch_detect(net, img[..., ::-1].copy(), device="cpu")
    keeps = [ref_bbox.nms(bboxlists[:, i, :], 0.3) for i in range(bboxlists.shape[1])]
    lists = [bboxlists[keep, i, :] for i, keep in enumerate(keeps)]
    lists = [[x for x in bl if x[-1] > 0.5] for bl in lists]
    ref_rects = []
    for d in lists:
        if len(d) == 0:
            ref_rects.append((-1, -1, -1, -1))
            continue
        d0 = np.clip(d[0], 0, None)
        ref_rects.append(tuple(map(int, d0[:-1])))
    # oracle restatement must agree
    levels = s3fd_ref.dense_boxes(oo)
    dets = s3fd_ref.detections(levels)
    orects = [r if r is not None else (-1, -1, -1, -1) for r in s3fd_ref.rects(dets)]
    print("reference rects", ref_rects, " kept per image", [len(l) for l in lists])
    assert orects == ref_rects, (orects, ref_rects)
    for a, b in zip(lists, dets):
        assert len(a) == len(b) and all(np.allclose(x, y, atol=1e-4) for x, y in zip(a, b))
    out["rects"] = np.asarray(ref_rects, dtype=np.int64)
    out["n_kept"] = np.asarray([len(l) for l in lists], dtype=np.int64)
    out["kept0"] = np.asarray(lists[0][:8], dtype=np.float32) if lists[0] else np.zeros((0, 5), np.float32)
    for i, lv in enumerate(levels):
        out["dense%d" % i] = lv[:, ::max(1, lv.shape[1] // 64)]        # subsample of the dense decoded maps
    path = os.path.join(HERE, "golden_s3fd_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
