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


def test_device_nms_keep_lists_are_bit_exact(cuda):
    """`w2l_s3fd_nms` against bbox.py:44-64 as restated in the oracle: random overlapping boxes (a few clusters, so that many
    boxes are suppressed), a gate, degenerate boxes (empty union: NaN overlap -> removed, as `np.where(ovr <= thresh)` does),
    an empty candidate set, and the reference's golden keep counts through `detect_from_batch` (test above)."""
    from wav2lip_amd.face_detection.s3fd import nms, nms_batch
    rng = np.random.default_rng(7)
    for n, thresh in ((1, 0.3), (37, 0.3), (700, 0.3), (3000, 0.5), (1500, 0.0)):
        cx = rng.choice([40.0, 90.0, 200.0, 210.0], n) + rng.normal(0, 6, n)
        cy = rng.choice([30.0, 120.0], n) + rng.normal(0, 6, n)
        w, h = rng.uniform(8, 60, n), rng.uniform(8, 60, n)
        score = (rng.permutation(n) + 0.5) / n                                       # distinct scores
        d = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, score], 1).astype(np.float32)
        assert len(np.unique(d[:, 4])) == n
        assert nms(d, thresh) == [int(i) for i in s3fd_ref.nms(d, thresh)], (n, thresh)
        # the gated batch form: two images, the second one reversed
        tb = torch.from_numpy(np.stack([d, d[::-1].copy()])).to(cuda)
        keep, counts = nms_batch(tb, 0.25, thresh)
        for b, db in enumerate((d, d[::-1].copy())):
            rows = np.nonzero(db[:, 4] > 0.25)[0]
            ref = [int(rows[i]) for i in s3fd_ref.nms(db[rows], thresh)] if len(rows) else []
            assert keep[b, :int(counts[b])].cpu().tolist() == ref, (n, thresh, b)
    # degenerate: zero-area boxes far apart have an empty intersection AND (x2 - x1 + 1) = 0 -> 0 / 0
    z = np.array([[5, 5, 4, 4, 0.9], [50, 50, 49, 49, 0.8], [5, 5, 30, 30, 0.7]], np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = [int(i) for i in s3fd_ref.nms(z, 0.3)]
    assert nms(z, 0.3) == ref
    # nothing above the gate
    keep, counts = nms_batch(torch.zeros(2, 9, 5, device=cuda), 0.05, 0.3)
    assert counts.cpu().tolist() == [0, 0]
    assert nms(np.zeros((0, 5), np.float32), 0.3) == []
    # equal scores: the later row wins (documented order; the reference's own order for ties is numpy-build dependent)
    e = np.array([[0, 0, 10, 10, 0.5], [100, 100, 110, 110, 0.5], [0, 0, 10, 10, 0.5]], np.float32)
    assert nms(e, 0.3) == [2, 1]


def test_nms_orders_negative_scores_and_takes_tables_larger_than_the_device_bitmap(cuda):
    """(a) `nms(dets, thresh)` has no gate: negative scores (and -0.0) must rank BELOW the positive ones, as bbox.py:50
    `scores.argsort()[::-1]` ranks them; (b) a table of more than 262 144 rows is accepted (the device bound is on the rows ABOVE THE
    GATE); (c) more than 262 144 rows above the gate: that image's suppression runs on the host, the other image's on the device,
    both equal to the oracle's keep list."""
    from wav2lip_amd.face_detection.s3fd import _nms_host, nms, nms_batch
    d = np.array([[0, 0, 10, 10, -0.5], [100, 0, 110, 10, 0.25], [200, 0, 210, 10, -0.0], [300, 0, 310, 10, 0.75],
                  [400, 0, 410, 10, -2.0], [1, 1, 11, 11, 0.5]], np.float32)
    assert nms(d, 0.3) == [int(i) for i in s3fd_ref.nms(d, 0.3)] == [3, 5, 1, 2, 4]
    rng = np.random.default_rng(11)
    P = 300000
    big = np.zeros((P, 5), np.float32)
    big[:, 0] = rng.uniform(0, 4000, P)
    big[:, 1] = rng.uniform(0, 2000, P)
    big[:, 2] = big[:, 0] + rng.uniform(5, 40, P)
    big[:, 3] = big[:, 1] + rng.uniform(5, 40, P)
    big[:, 4] = rng.permutation(P).astype(np.float32) / P * 0.04          # all below the 0.05 gate ...
    hot = rng.choice(P, 500, replace=False)
    big[hot, 4] = 0.5 + rng.permutation(500).astype(np.float32) / 1000      # ... except 500 rows
    keep, counts = nms_batch(torch.from_numpy(big[None]).to(cuda), 0.05, 0.3)
    rows = np.nonzero(big[:, 4] > 0.05)[0]
    ref = [int(rows[i]) for i in s3fd_ref.nms(big[rows], 0.3)]
    assert keep[0, :int(counts[0])].cpu().tolist() == ref
    # every row above the gate: image 0 overflows the device pass, image 1 (a short list padded with gated-out rows) does not.
    # One box 262 208 times with distinct scores: the best one suppresses all others, so the host pass is a single round.
    P2 = 262144 + 64
    wide = np.zeros((P2, 5), np.float32)
    wide[:, 2:4] = 10
    wide[:, 4] = 0.1 + rng.permutation(P2).astype(np.float32) / P2 * 0.8
    small = np.zeros((P2, 5), np.float32)
    small[:len(hot)] = big[hot]                  # 500 clustered-enough rows above the gate, the rest gated out (score 0)
    keep, counts = nms_batch(torch.from_numpy(np.stack([wide, small])).to(cuda), 0.05, 0.3)
    assert int(counts[0]) == 1 and int(keep[0, 0]) == int(np.argmax(wide[:, 4]))
    rows = np.nonzero(small[:, 4] > 0.05)[0]
    assert keep[1, :int(counts[1])].cpu().tolist() == [int(rows[i]) for i in s3fd_ref.nms(small[rows], 0.3)]
    # the host pass itself against the oracle on a clustered set
    c = np.stack([rng.choice([40.0, 90.0], 400) + rng.normal(0, 6, 400), rng.choice([30.0, 120.0], 400) + rng.normal(0, 6, 400)], 1)
    wh = rng.uniform(8, 60, (400, 2))
    dd = np.concatenate([c - wh / 2, c + wh / 2, ((rng.permutation(400) + 0.5) / 400)[:, None]], 1).astype(np.float32)
    assert _nms_host(dd, 0.3) == [int(i) for i in s3fd_ref.nms(dd, 0.3)]
