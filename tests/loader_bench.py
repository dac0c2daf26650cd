#!/usr/bin/env python
"""(Lives under tests/: it times the oracle's numpy restatement next to the loader, and only tests/ may use oracle/.)
Input-side throughput: CycleLoader.getBatch on a generated COCO-sized dataset (480x640 JPEGs, 2-3 referred
objects per image, polygon-like masks) against the oracle's numpy restatement of the reference's per-image CPU work
(prep_im_for_blob + RLE decode + union + nearest resize, one thread).  Prints per-batch times and the device kernel time."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch


def main():
    from PIL import Image
    from data_util import write_tiny_dataset
    from lang2seg_amd.loaders.cycle_loader import CycleLoader, imread_bgr
    from lang2seg_amd.model.config import cfg
    from oracle import data as OD
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    root = tempfile.mkdtemp()
    info, labels, imgs = write_tiny_dataset(root, sizes=tuple((480, 640) if i % 3 else (640, 427) for i in range(n_img)), label_length=10, vocab=1999)
    for im in info['images']:                       # JPEG files like the dataset's
        p = os.path.join(root, 'images', im['file_name'])
        Image.open(p).save(p.replace('.png', '.jpg'), quality=90)
    for r in info['refs']:
        r['split'] = 'train'
    import json
    json.dump(info, open(os.path.join(root, 'data.json'), 'w'))
    for prefetch in (False, True):
        ld = CycleLoader(os.path.join(root, 'data.json'), os.path.join(root, 'data.h5'), image_root=os.path.join(root, 'images'),
                         image_pattern='img_{:0>12d}.jpg', prefetch=prefetch, verbose=False)
        ts = []
        for _ in range(2 * n_img):
            t0 = time.time()
            b = ld.getBatch('train')
            ts.append(time.time() - t0)
            if prefetch:
                time.sleep(0.008)                   # the train step the worker thread overlaps with
        torch.cuda.synchronize()
        ts = np.array(ts[n_img:]) * 1e3             # second pass over the images: allocator pools are warm
        print('CycleLoader.getBatch prefetch=%d: median %.2f ms, max %.2f ms per image on the training thread' % (prefetch, np.median(ts), ts.max()))
    # device part alone
    ld = CycleLoader(os.path.join(root, 'data.json'), os.path.join(root, 'data.h5'), image_root=os.path.join(root, 'images'),
                     image_pattern='img_{:0>12d}.jpg', prefetch=False, verbose=False)
    hs = ld._host_stage(ld.split_ix['train'][1])
    ld._device_stage(hs, 600, 1000); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ld._device_stage(hs, 600, 1000)
    b.record(); torch.cuda.synchronize()
    print('device stage (H2D of raw bytes + prep_image + rle_to_mask x refs): %.1f us per image' % (a.elapsed_time(b) * 1e3 / 20))
    # the reference's per-image CPU work, restated (one thread)
    t0 = time.time()
    for im in info['images'][:8]:
        bgr = imread_bgr(os.path.join(root, 'images', im['file_name'].replace('.png', '.jpg')))
        OD.get_batch(info, labels, im['image_id'], bgr, cfg.PIXEL_MEANS, 600, 1000)
    print('oracle (numpy restatement of cycle_loader.py getBatch, 1 thread): %.1f ms per image' % ((time.time() - t0) / 8 * 1e3))


if __name__ == '__main__':
    main()
