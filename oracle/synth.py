"""Seeded synthetic blobs: one image x one expression, the blobs
contract of lib/loaders/cycle_loader.py:327-357."""
import numpy as np


def make_blob(H=600, W=1000, T=20, V=3349, seed=1234, scale=1.6):
    rs = np.random.RandomState(seed)
    data = rs.normal(0, 50.0, (1, H, W, 3)).astype(np.float32)
    x1 = rs.uniform(0, 0.6 * W); y1 = rs.uniform(0, 0.5 * H)
    w = rs.uniform(0.08 * W, 0.4 * W); h = rs.uniform(0.13 * H, 0.5 * H)
    x2 = min(x1 + w, W - 1); y2 = min(y1 + h, H - 1)
    cls = rs.randint(1, 81)
    gt_boxes = np.array([[x1, y1, x2, y2, cls]], np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
    gt_masks = ((((xx - cx) / max((x2 - x1) / 2, 1)) ** 2 + ((yy - cy) / max((y2 - y1) / 2, 1)) ** 2) <= 1.0).astype(np.uint8)[None]
    labels = rs.randint(1, V, (1, T)).astype(np.int64)
    cap_labels = np.zeros((1, T + 2), np.int64); cap_labels[0, 1:T + 1] = labels[0]
    cap_masks = np.ones((1, T + 2), np.float32)
    return dict(data=data, im_info=np.array([[H, W, scale]], np.float32), gt_boxes=gt_boxes, gt_masks=gt_masks,
                labels=labels, cap_labels=cap_labels, cap_masks=cap_masks, file_name='synthetic_%d' % seed)
