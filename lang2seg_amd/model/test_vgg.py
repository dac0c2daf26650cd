"""Evaluation of the VGG16 / Faster R-CNN network (reference: pyutils/mask-faster-rcnn/lib/model/test_vgg.py:187-276): the same
loop as model/test.py with the segmentation half commented out — per sentence im_detect, the best (roi, class) by foreground
score, box IoU >= 0.5 against the referred box.  Returns (box accuracy, number of sentences) like the reference."""
from .test import im_detect, best_detection, computeIoU_box  # noqa: F401

import numpy as np


def eval_split(loader, model, crit, split, opt, max_per_image=100, thresh=0.):
    num_sents = opt.get('num_sents', -1)
    verbose = opt.get('verbose', True)
    model.eval()
    loss_evals, acc = 0, 0
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
                blobs['_device'] = {'data': dev['data']}
            else:
                blobs['data'] = data['data']
            scores, boxes, net_conv, im_scale = im_detect(model, blobs)
            pred_roi, pred_class, pred_box = best_detection(scores, boxes)
            gt_box = blobs['gt_boxes'][0, :4] / im_scale
            if computeIoU_box(pred_box, gt_box) >= 0.5:
                acc += 1
            loss_evals += 1
            if num_sents > 0 and loss_evals >= num_sents:
                finish = True
                break
        if verbose:
            print('evaluating [%s] ... sent %d, det acc=%.2f%%' % (split, loss_evals, acc * 100.0 / max(loss_evals, 1)))
        if finish or data['bounds']['wrapped']:
            break
    return acc * 1.0 / max(loss_evals, 1), loss_evals
