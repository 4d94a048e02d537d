"""S3FD face detector on the HIP path against the golden frozen from the REAL reference network +
batch_detect + nms (tests/golden/make_golden_s3fd.py) and against the oracle (oracle/s3fd_ref.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT
from oracle import s3fd_ref
from wav2lip_amd import _lib
from wav2lip_amd._lib import check, ptr

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_s3fd import images, seeded_state_dict  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_s3fd_v1.npz"))


def test_glue_kernels(cuda):
    lib = _lib.load()
    s = _lib.current_stream()
    torch.manual_seed(0)
    x = torch.randn(2, 64, 11, 14)
    xg = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    y = torch.empty(2, 5, 7, 64, device=cuda)
    check(lib.w2l_maxpool2x2(s, 2, 11, 14, 64, ptr(xg), 64, ptr(y), 64))
    assert torch.equal(y.permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 2, 2))
    w = torch.rand(64) * 10
    yn = torch.empty_like(xg)
    check(lib.w2l_l2norm_scale(s, 2 * 11 * 14, 64, ptr(xg), 64, ptr(w.to(cuda)), ptr(yn), 64))
    ref = s3fd_ref.l2norm(x, w)
    assert (yn.permute(0, 3, 1, 2).cpu() - ref).abs().max() <= 1e-5
    img = np.random.default_rng(0).integers(0, 256, (2, 9, 7, 3), dtype=np.uint8)
    out = torch.full((2, 9, 7, 4), 5.0, device=cuda)
    check(lib.w2l_s3fd_pack(s, 2 * 9 * 7, ptr(torch.from_numpy(img).to(cuda)), ptr(out), 4))
    refp = s3fd_ref.preprocess(img).permute(0, 2, 3, 1)
    assert torch.equal(out[..., :3].cpu(), refp) and bool((out[..., 3] == 0).all())


def _model(cuda):
    from wav2lip_amd import face_detection as fd
    net = fd.s3fd()
    net.load_state_dict(seeded_state_dict())
    return net.to(cuda).eval()


def test_network_outputs_match_reference_golden(gold, cuda):
    net = _model(cuda)
    x = s3fd_ref.preprocess(images()).to(cuda)
    outs = net(x)
    assert len(outs) == 12
    for i, o in enumerate(outs):
        ref = torch.from_numpy(gold["out%d" % i])
        assert o.shape == ref.shape
        scale = ref.abs().max().item()
        assert (o.cpu() - ref).abs().max().item() <= 2e-4 * scale + 1e-5, (i, (o.cpu() - ref).abs().max().item(), scale)


def test_dense_boxes_and_detections_match_reference_golden(gold, cuda):
    from wav2lip_amd import face_detection as fd
    img = images()
    fa = fd.FaceAlignment(fd.LandmarksType._2D, device="cuda", state_dict=seeded_state_dict())
    levels = fa.face_detector.dense_boxes(torch.from_numpy(img).to(cuda))
    for i, lv in enumerate(levels):
        got = lv.cpu().numpy()
        ref = gold["dense%d" % i]
        sub = got[:, ::max(1, got.shape[1] // 64)]
        assert sub.shape == ref.shape and np.abs(sub - ref).max() <= 2e-3, (i, np.abs(sub - ref).max())
    dets = fa.detect_from_batch(img)
    assert [len(d) for d in dets] == gold["n_kept"].tolist()
    assert np.abs(np.asarray(dets[0][:8]) - gold["kept0"]).max() <= 2e-3
    rects = fa.get_detections_for_batch(img)
    assert [tuple(r) for r in rects] == [tuple(r) for r in gold["rects"].tolist()]


def test_face_detect_front_end(cuda):
    """inference.py:68-104: rects -> pads -> integer smoothing -> crops; a black frame has no face -> ValueError"""
    from wav2lip_amd import face_detection as fd
    from wav2lip_amd.inference import face_detect
    fa = fd.FaceAlignment(fd.LandmarksType._2D, device="cuda", state_dict=seeded_state_dict())
    img = images()
    frames = [img[0], img[1], img[0], img[1], img[0], img[1], img[0]]
    res = face_detect(frames, fa, pads=(0, 10, 0, 0), nosmooth=False, batch_size=2)
    rects = s3fd_ref.rects(s3fd_ref.detections(s3fd_ref.dense_boxes(s3fd_ref.s3fd_forward(seeded_state_dict(), s3fd_ref.preprocess(img)))))
    boxes = []
    for k in range(len(frames)):
        x1, y1, x2, y2 = rects[k % 2]
        boxes.append([max(0, x1), max(0, y1), min(128, x2), min(96, y2 + 10)])
    boxes = s3fd_ref.get_smoothened_boxes(np.array(boxes), T=5)
    for (crop, (y1, y2, x1, x2)), b, f in zip(res, boxes, frames):
        assert (x1, y1, x2, y2) == tuple(int(v) for v in b)
        assert np.array_equal(crop, f[y1:y2, x1:x2])
