"""Mask post-processing of the evaluation path — host side, numpy + PIL
(reference: pyutils/mask-faster-rcnn/lib/utils/mask_utils.py:31-77 `clip_np_boxes`, `recover_masks`).
`imresize` restates scipy.misc.imresize of scipy <= 1.2 (pilutil.py: bytescale -> PIL resize), which the
reference's environment provides and which modern scipy removed."""
import numpy as np
from PIL import Image

_RESAMPLE = {'nearest': 0, 'lanczos': 1, 'bilinear': 2, 'bicubic': 3, 'cubic': 3}


def bytescale(data, cmin=None, cmax=None, high=255, low=0):
    """scipy.misc.pilutil.bytescale: uint8 passes through; everything else is stretched from [min, max] to [low, high]."""
    data = np.asarray(data)
    if data.dtype == np.uint8:
        return data
    if cmin is None:
        cmin = data.min()
    if cmax is None:
        cmax = data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = float(high - low) / cscale
    bytedata = (data - cmin) * scale + low
    return (bytedata.clip(low, high) + 0.5).astype(np.uint8)


def imresize(arr, size, interp='bilinear'):
    """size: (rows, cols) tuple, int percentage or float fraction (scipy.misc.imresize)."""
    im = Image.fromarray(bytescale(arr))
    if isinstance(size, (int, np.integer)):
        size = tuple((np.array(im.size) * (size / 100.0)).astype(int))
    elif isinstance(size, float):
        size = tuple((np.array(im.size) * size).astype(int))
    else:
        size = (int(size[1]), int(size[0]))
    return np.array(im.resize(size, resample=_RESAMPLE[interp]))


def clip_np_boxes(boxes, im_shape):
    """mask_utils.py:31-44 (in place)."""
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def recover_masks(masks, rois, ih, iw, interp='bilinear'):
    """mask_utils.py:46-77: (N,14,14) float [0,1] masks + (N,4) boxes -> (N, ih, iw) uint8 [0,255] (each mask is stretched to
    its own [min,max] by imresize's bytescale, as in the reference; `masks` and `rois` are modified in place like there)."""
    assert rois.shape[0] == masks.shape[0], '%s rois vs %d masks' % (rois.shape[0], masks.shape[0])
    num_rois = rois.shape[0]
    recovered = np.zeros((num_rois, ih, iw), dtype=np.uint8)
    rois = clip_np_boxes(rois, (ih, iw))
    for i in np.arange(num_rois):
        mask = masks[i, :, :]
        mask *= 255.
        h, w = int(rois[i, 3] - rois[i, 1] + 1), int(rois[i, 2] - rois[i, 0] + 1)
        x, y = int(rois[i, 0]), int(rois[i, 1])
        mask = imresize(mask, (h, w), interp=interp)
        recovered[i, y:y + h, x:x + w] = mask
    return recovered
