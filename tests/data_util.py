"""Helpers of the loader tests: the RLE fixture and a tiny on-disk dataset in the reference's format."""
import json
import os

import numpy as np

from golden_util import GOLD


def load_rle_fixture():
    g = dict(np.load(os.path.join(GOLD, 'ref_rle.npz'), allow_pickle=False))
    cases = []
    for i in range(len(g['h'])):
        h, w = int(g['h'][i]), int(g['w'][i])
        cnt = g['counts'][g['counts_off'][i]:g['counts_off'][i + 1]]
        bits = np.unpackbits(g['masks_packed'][g['masks_off'][i]:g['masks_off'][i + 1]])[:h * w].reshape(h, w)
        cases.append(dict(h=h, w=w, s=g['strings'][i].decode('ascii'), counts=cnt, mask=bits.astype(np.uint8)))
    groups = [[int(x) for x in row if x >= 0] for row in g['groups']]
    return cases, groups


def write_tiny_dataset(root, seed=0, sizes=((120, 160), (90, 150), (200, 140)), label_length=8, vocab=30):
    """data.json + data.h5.npy + PNG images (lossless, so the decoded pixels are the written ones) under `root`.
    Layout of lib/loaders/cycle_loader.py's docstring; refs hold 'rle' lists as the reference's prepro writes them."""
    from PIL import Image
    from oracle import data as OD
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'images'), exist_ok=True)
    images, refs, sents, anns = [], [], [], []
    imgs = {}
    labels = []
    ref_id = sent_id = 0
    for k, (h, w) in enumerate(sizes):
        image_id = 1000 + k
        bgr = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(os.path.join(root, 'images', 'img_%012d.png' % image_id))
        imgs[image_id] = bgr
        rids = []
        for r in range(1 + k % 2 + 1):                      # 2, 3, 2 referred objects
            x0, y0 = rs.randint(0, w // 2), rs.randint(0, h // 2)
            bw, bh = rs.randint(8, w // 2), rs.randint(8, h // 2)
            segs = []
            for sgi in range(1 + (r % 2)):                  # some objects have two segments
                m = np.zeros((h, w), np.uint8)
                yy, xx = np.mgrid[0:h, 0:w]
                cx, cy = x0 + bw * (0.3 + 0.4 * sgi), y0 + bh * 0.5
                m[((xx - cx) / (0.3 * bw)) ** 2 + ((yy - cy) / (0.5 * bh)) ** 2 <= 1.0] = 1
                segs.append({'size': [h, w], 'counts': OD.rle_to_string(OD.rle_encode(m))})
            sids = []
            for _ in range(1 + rs.randint(0, 3)):
                n = rs.randint(2, label_length + 1)
                row = np.zeros(label_length, np.uint32); row[:n] = rs.randint(1, vocab, n)
                labels.append(row)
                sents.append({'sent_id': sent_id, 'tokens': [], 'h5_id': len(labels) - 1}); sids.append(sent_id); sent_id += 1
            refs.append({'ref_id': ref_id, 'ann_id': ref_id, 'box': [float(x0), float(y0), float(bw), float(bh)], 'image_id': image_id,
                         'split': 'train' if k < 2 else 'val', 'category_id': int(rs.randint(1, 81)), 'sent_ids': sids, 'att_wds': [],
                         'rle': segs})
            anns.append({'ann_id': ref_id, 'category_id': refs[-1]['category_id'], 'image_id': image_id, 'box': refs[-1]['box'], 'h5_id': ref_id})
            rids.append(ref_id); ref_id += 1
        # every image of a split must agree on it: the split is read from the first ref (cycle_loader.py:62)
        images.append({'image_id': image_id, 'ref_ids': rids, 'file_name': 'img_%012d.png' % image_id, 'width': w, 'height': h, 'h5_id': k})
    w2i = {'w%d' % i: i for i in range(1, vocab)}; w2i['<UNK>'] = vocab
    info = {'refs': refs, 'images': images, 'anns': anns, 'sentences': sents, 'word_to_ix': w2i,
            'cat_to_ix': {'c%d' % i: i for i in range(1, 81)}, 'label_length': label_length}
    json.dump(info, open(os.path.join(root, 'data.json'), 'w'))
    np.save(os.path.join(root, 'data.h5.npy'), np.stack(labels))
    return info, np.stack(labels), imgs
