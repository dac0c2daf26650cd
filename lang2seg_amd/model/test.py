"""Evaluation path (reference: pyutils/mask-faster-rcnn/lib/model/test.py:97-360): `im_detect`, box / segmentation IoU and
`eval_split` — pick the highest-scoring (roi, class), box accuracy @0.5, run the mask head on that box, recover the mask at
the original image size, cumulative IoU and precision@{.5,.6,.7,.8,.9}.  The network side is `Network.test_image` /
`_predict_masks_from_boxes_and_labels` (HIP kernels); everything here is per-sentence host post-processing, as in the reference."""
import numpy as np

from ..utils.mask_utils import recover_masks, imresize
from .config import cfg


def bbox_transform_inv_np(boxes, deltas):
    """model/bbox_transform.py:36-62 in float32 numpy ((n,4) boxes, (n,4C) deltas)."""
    boxes = boxes.astype(np.float32); deltas = deltas.astype(np.float32)
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), np.float32)
    widths = boxes[:, 2] - boxes[:, 0] + np.float32(1.0)
    heights = boxes[:, 3] - boxes[:, 1] + np.float32(1.0)
    ctr_x = boxes[:, 0] + np.float32(0.5) * widths
    ctr_y = boxes[:, 1] + np.float32(0.5) * heights
    dx, dy, dw, dh = deltas[:, 0::4], deltas[:, 1::4], deltas[:, 2::4], deltas[:, 3::4]
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = np.exp(dw) * widths[:, None]
    ph = np.exp(dh) * heights[:, None]
    out = np.zeros_like(deltas)
    out[:, 0::4] = pcx - np.float32(0.5) * pw
    out[:, 1::4] = pcy - np.float32(0.5) * ph
    out[:, 2::4] = pcx + np.float32(0.5) * pw
    out[:, 3::4] = pcy + np.float32(0.5) * ph
    return out


def _clip_boxes(boxes, im_shape):
    """test.py:77-87."""
    boxes[:, 0::4] = np.maximum(boxes[:, 0::4], 0)
    boxes[:, 1::4] = np.maximum(boxes[:, 1::4], 0)
    boxes[:, 2::4] = np.minimum(boxes[:, 2::4], im_shape[1] - 1)
    boxes[:, 3::4] = np.minimum(boxes[:, 3::4], im_shape[0] - 1)
    return boxes


def detect_from_outputs(scores, bbox_pred, rois, im_info):
    """the numpy half of im_detect (test.py:113-128): class-wise boxes in the ORIGINAL image."""
    scale = im_info[0][2]
    boxes = rois[:, 1:5] / scale
    scores = np.reshape(scores, [scores.shape[0], -1])
    bbox_pred = np.reshape(bbox_pred, [bbox_pred.shape[0], -1])
    if cfg.TEST.BBOX_REG:
        pred_boxes = bbox_transform_inv_np(boxes, bbox_pred)
        im_shape = (round(im_info[0][0] / scale), round(im_info[0][1] / scale), 3)
        pred_boxes = _clip_boxes(pred_boxes, im_shape)
    else:
        pred_boxes = np.tile(boxes, (1, scores.shape[1]))
    return scores, pred_boxes


def im_detect(net, blobs):
    """test.py:97-130 -> (scores (n,C), pred_boxes (n,4C) original image, net_conv, im_scale)."""
    _, scores, bbox_pred, rois, net_conv = net.test_image(blobs)
    scores, pred_boxes = detect_from_outputs(scores, bbox_pred, rois, blobs['im_info'])
    return scores, pred_boxes, net_conv, blobs['im_info'][0][2]


def computeIoU_box(box1, box2):
    """test.py:163-176."""
    inter_x1 = max(box1[0], box2[0]); inter_y1 = max(box1[1], box2[1])
    inter_x2 = min(box1[2], box2[2]); inter_y2 = min(box1[3], box2[3])
    if inter_x1 < inter_x2 and inter_y1 < inter_y2:
        inter = (inter_x2 - inter_x1 + 1) * (inter_y2 - inter_y1 + 1)
    else:
        inter = 0
    union = (box1[2] - box1[0] + 1) * (box1[3] - box1[1] + 1) + (box2[2] - box2[0] + 1) * (box2[3] - box2[1] + 1) - inter
    return float(inter) / union


def computeIoU_seg(pred_seg, gt_seg):
    """test.py:179-183."""
    I = np.sum(np.logical_and(pred_seg, gt_seg))
    U = np.sum(np.logical_or(pred_seg, gt_seg))
    return I, U


def best_detection(scores, boxes):
    """test.py:257-260: the (roi, class) with the highest foreground score, first occurrence in row-major order."""
    pred = np.where(scores == np.max(scores[:, 1:]))
    pred_roi, pred_class = pred[0][0], pred[1][0]
    return pred_roi, pred_class, boxes[pred_roi, pred_class * 4:(pred_class + 1) * 4]


def segment_from_mask_prob(mask_prob, pred_box, im_info):
    """test.py:331-334: 14x14 probabilities of the chosen box -> binary mask at the original image size."""
    scale = im_info[0][2]
    ih, iw = int(round(im_info[0][0] / scale)), int(round(im_info[0][1] / scale))
    pred_mask = recover_masks(mask_prob, np.array([pred_box]), ih, iw)
    return np.squeeze((pred_mask > 122.).astype(np.uint8), axis=0)


def eval_split(loader, model, crit, split, opt, max_per_image=100, thresh=0.):
    """test.py:185-450.  Returns the reference's tuple
    (acc, eval_seg_iou_list, seg_correct, seg_total, cum_I, cum_U, num_sent): box accuracy @0.5, the precision thresholds, the
    number of sentences whose mask IoU reaches each threshold, the number of sentences, cumulative intersection / union pixels."""
    num_sents = opt.get('num_sents', -1)
    verbose = opt.get('verbose', True)
    model.eval()
    loss_evals, acc, num_sent = 0, 0, 0
    cum_I, cum_U = 0, 0
    eval_seg_iou_list = [.5, .6, .7, .8, .9]
    seg_correct = np.zeros(len(eval_seg_iou_list), dtype=np.int32)
    seg_total = 0
    finish = False
    while True:
        data = loader.getTestBatch(split)
        labels = np.asarray(data['labels'])
        for i in range(labels.shape[0]):
            label = labels[i:i + 1, :]
            max_len = int((label != 0).sum())
            blobs = dict(im_info=data['im_info'], file_name=data.get('file_name'), bounds=data.get('bounds'),
                         gt_boxes=data['gt_boxes'][i:i + 1, :], gt_masks=data['gt_masks'][i:i + 1, :, :], labels=label[:, :max_len], sent_id=i)
            dev = dict.get(data, '_device', None)
            if dev is not None and 'data' in dev:
                blobs['_device'] = {'data': dev['data']}          # loaders/cycle_loader.py keeps the image blob in HBM: no host round trip
            else:
                blobs['data'] = data['data']
            scores, boxes, net_conv, im_scale = im_detect(model, blobs)
            pred_roi, pred_class, pred_box = best_detection(scores, boxes)
            gt_box = blobs['gt_boxes'][0, :4] / im_scale
            if computeIoU_box(pred_box, gt_box) >= 0.5:
                acc += 1
            loss_evals += 1
            mask_prob = model._predict_masks_from_boxes_and_labels(net_conv, np.array([pred_box]) * im_scale, np.array([pred_class]))
            mask_prob = mask_prob.cpu().numpy().copy()
            pred_mask = segment_from_mask_prob(mask_prob, pred_box.copy(), blobs['im_info'])
            gt_mask = imresize(np.squeeze(blobs['gt_masks'], axis=0), size=pred_mask.shape, interp='nearest')
            I, U = computeIoU_seg(pred_mask, gt_mask)
            cum_I += I; cum_U += U
            for k, t in enumerate(eval_seg_iou_list):
                seg_correct[k] += (I * 1.0 / U >= t)
            seg_total += 1
            num_sent += 1
            if num_sents > 0 and loss_evals >= num_sents:
                finish = True
                break
        if verbose:
            b = data['bounds']
            print('evaluating [%s] ... image[%d/%d]\'s sents, det acc=%.2f%%, seg acc=%.2f%%, seg IoU=%.2f%%' % (
                split, b['it_pos_now'], b['it_max'], acc * 100.0 / max(loss_evals, 1), seg_correct[0] * 100.0 / max(seg_total, 1),
                cum_I * 100.0 / max(cum_U, 1)))
        if finish or data['bounds']['wrapped']:
            break
    return acc / loss_evals, eval_seg_iou_list, seg_correct, seg_total, cum_I, cum_U, num_sent


def summarize(eval_seg_iou_list, seg_correct, seg_total, cum_I, cum_U):
    """the text block of tools/eval_spatial.py:104-110 and ([prec@X], overall IoU) as fractions"""
    s = ''
    for k, t in enumerate(eval_seg_iou_list):
        s += '    precision@%s = %.2f\n' % (str(t), seg_correct[k] * 100. / seg_total)
    s += '    overall IoU = %.2f\n' % (cum_I * 100. / cum_U)
    return s, [seg_correct[k] * 1.0 / seg_total for k in range(len(eval_seg_iou_list))], cum_I * 1.0 / cum_U
