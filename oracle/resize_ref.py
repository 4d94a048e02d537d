"""Oracle for the two `cv2.resize` calls either side of the hot path (inference.py:126 crop -> 96x96, :270 96x96 -> box
size) and the paste-back (:271).  TEST INFRASTRUCTURE.

PARITY UNPINNED at the OpenCV boundary: opencv-python==4.1.0.25 (requirements.txt:4) is neither under /root/reference
nor importable in this image, and the reference has no test for it.  This restates the published algorithm of
`cv::resize(..., INTER_LINEAR)` for CV_8UC3 (modules/imgproc/src/resize.cpp, 4.1.0): the fixed-point bilinear path
  * dsize == ssize                     -> copy
  * exact 2x2 down-scale              -> the INTER_AREA fast path (s00+s01+s10+s11+2)>>2  (resize() substitutes it)
  * otherwise: fx = (float)((dx+0.5)*scale_x - 0.5), sx = floor(fx), fx -= sx, edge clamps (sx<0 -> sx=0,fx=0;
    sx>=w-1 -> sx=w-1,fx=0); coefficients cvRound((1-fx)*2048), cvRound(fx*2048) as int16 (same vertically, where the
    rows are clipped instead of the coefficient being zeroed); horizontal pass in int32 (S0*a0 + S1*a1), vertical pass
    uchar(( ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2).
Plain numpy, vectorised over the destination grid.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _axis_tables(ssize, dsize):
    """per destination index: (s0, s1 source indices, a0, a1 int coefficients) along one axis, horizontal convention"""
    scale = 1.0 / (float(dsize) / float(ssize))          # resize(): inv_scale = dsize/ssize (double); scale = 1./inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coef(f):
    a1 = np.rint(f.astype(np.float32) * np.float32(COEF_SCALE)).astype(np.int64)
    a0 = np.rint((np.float32(1.0) - f.astype(np.float32)) * np.float32(COEF_SCALE)).astype(np.int64)
    return a0, a1


def resize_linear_u8(src, dsize_wh):
    """src uint8 [H,W,C]; dsize_wh = (width, height) as cv2.resize takes it -> uint8 [h,w,C]"""
    src = np.asarray(src)
    H, W = src.shape[:2]
    w, h = int(dsize_wh[0]), int(dsize_wh[1])
    if (w, h) == (W, H):
        return src.copy()
    if W == 2 * w and H == 2 * h:
        s = src.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, fx = _axis_tables(W, w)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= W - 1
    fx = np.where(hi, np.float32(0), fx)
    sx = np.where(hi, W - 1, sx)
    ax0, ax1 = _coef(fx)
    sx1 = np.minimum(sx + 1, W - 1)                      # only read with coefficient 0 at the right edge
    sy, fy = _axis_tables(H, h)
    by0, by1 = _coef(fy)
    y0 = np.clip(sy, 0, H - 1)
    y1 = np.clip(sy + 1, 0, H - 1)
    s = src.astype(np.int64)
    rows = s[:, sx] * ax0[None, :, None] + s[:, sx1] * ax1[None, :, None]       # [H, w, C] int32 range
    r0, r1 = rows[y0], rows[y1]                                                 # [h, w, C]
    out = (((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_resize(frame, box, size=96):
    """inference.py:121-126: face = frame[y1:y2, x1:x2]; cv2.resize(face, (size, size))"""
    y1, y2, x1, x2 = box
    return resize_linear_u8(frame[y1:y2, x1:x2], (size, size))


def resize_paste(frame, pred_u8, box):
    """inference.py:270-271: p = cv2.resize(p.astype(np.uint8), (x2 - x1, y2 - y1)); f[y1:y2, x1:x2] = p (in place)"""
    y1, y2, x1, x2 = box
    frame[y1:y2, x1:x2] = resize_linear_u8(pred_u8, (x2 - x1, y2 - y1))
    return frame
