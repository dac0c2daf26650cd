"""Base anchor enumeration (host side, cached per configuration).
Follows pyutils/mask-faster-rcnn/lib/layer_utils/generate_anchors.py:41-103: ratio-major,
scale-minor windows around the (0,0,15,15) reference box, np.round half-to-even.
The per-pixel shifts of layer_utils/snippets.py:13-29 are applied inside the device kernels
(anchor(y,x,a) = base[a] + 16*(x,y,x,y))."""
import numpy as np


def _whc(a):
    w = a[2] - a[0] + 1.0
    h = a[3] - a[1] + 1.0
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _boxes(ws, hs, xc, yc):
    ws = np.asarray(ws, np.float64).reshape(-1, 1)
    hs = np.asarray(hs, np.float64).reshape(-1, 1)
    return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def base_anchors(scales, ratios, base_size=16):
    ratios = np.asarray(ratios, np.float64)
    scales = np.asarray(scales, np.float64)
    w, h, xc, yc = _whc(np.array([0, 0, base_size - 1, base_size - 1], np.float64))
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    per_ratio = _boxes(ws, hs, xc, yc)
    rows = []
    for r in per_ratio:
        w, h, xc, yc = _whc(r)
        rows.append(_boxes(w * scales, h * scales, xc, yc))
    return np.vstack(rows).astype(np.float32)


def all_anchors(H, W, scales, ratios, feat_stride=16):
    """(H*W*A, 4) in (y, x, a) order — what generate_anchors_pre returns; used by tests / test_image."""
    b = base_anchors(scales, ratios)
    sx, sy = np.meshgrid(np.arange(W) * feat_stride, np.arange(H) * feat_stride)
    sh = np.stack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()), 1).astype(np.float32)
    return (b[None, :, :] + sh[:, None, :]).reshape(-1, 4).astype(np.float32)
