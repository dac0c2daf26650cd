"""SyntheticLoader — the blobs contract of the reference loaders without a dataset.

`getBatch(split, batch_size=1)` returns the dict that lib/loaders/cycle_loader.py:327-357 builds
(`data` float32 (1,H,W,3) BGR mean-subtracted NHWC, `im_info` (1,3) = [H, W, scale], `gt_boxes`
(S,5) [x1,y1,x2,y2,cls], `gt_masks` uint8 (S,H,W), `labels` int64 (S,Lmax) zero padded,
`cap_labels` int64 (S,Lmax+2) with BOS/EOS = 0, `cap_masks` float32 (S,Lmax+2), `file_name`,
`ref_ids`, `bounds`) and keeps the cursor state the solver snapshots (`iterators`, `perm`,
`split_ix`; lib/loaders/loader.py:72-106, train_val_cycle.py:75-78,153-158).
Images are seeded noise of pixel scale, one random box + inscribed ellipse mask per sentence
(SURVEY.md §8d)."""
import numpy as np


class SyntheticLoader(object):
    def __init__(self, num_images=8, sents_per_image=1, H=600, W=1000, T=20, vocab_size=3349, seed=1234, scale=1.6, rank=0):
        self.H, self.W, self.T, self.scale = H, W, T, scale
        self.vocab_size = vocab_size
        self.label_length = T
        self.sents_per_image = sents_per_image
        self.seed = seed + rank * 10007
        self.split_ix = {'train': list(range(num_images)), 'val': list(range(num_images)), 'test': list(range(num_images))}
        self.iterators = {'train': 0, 'val': 0, 'test': 0}
        self.perm = {k: np.arange(len(v)) for k, v in self.split_ix.items()}
        self._cache = {}

    def _image(self, ix):
        if ix in self._cache:
            return self._cache[ix]
        H, W, T, V, S = self.H, self.W, self.T, self.vocab_size, self.sents_per_image
        rs = np.random.RandomState(self.seed + ix)
        data = rs.normal(0, 50.0, (1, H, W, 3)).astype(np.float32)
        gt_boxes = np.zeros((S, 5), np.float32)
        gt_masks = np.zeros((S, H, W), np.uint8)
        yy, xx = np.mgrid[0:H, 0:W]
        for s in range(S):
            x1 = rs.uniform(0, 0.6 * W); y1 = rs.uniform(0, 0.5 * H)
            w = rs.uniform(0.08 * W, 0.4 * W); h = rs.uniform(0.13 * H, 0.5 * H)
            x2 = min(x1 + w, W - 1); y2 = min(y1 + h, H - 1)
            gt_boxes[s] = [x1, y1, x2, y2, rs.randint(1, 81)]
            cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
            gt_masks[s] = ((((xx - cx) / max((x2 - x1) / 2, 1)) ** 2 + ((yy - cy) / max((y2 - y1) / 2, 1)) ** 2) <= 1.0)
        labels = rs.randint(1, V, (S, T)).astype(np.int64)
        cap_labels = np.zeros((S, T + 2), np.int64); cap_labels[:, 1:T + 1] = labels
        cap_masks = np.ones((S, T + 2), np.float32)
        blob = dict(data=data, im_info=np.array([[H, W, self.scale]], np.float32), gt_boxes=gt_boxes, gt_masks=gt_masks,
                    labels=labels, cap_labels=cap_labels, cap_masks=cap_masks, file_name='synthetic_%06d.jpg' % ix,
                    ref_ids=list(range(ix * S, ix * S + S)), sent_id=0)
        self._cache[ix] = blob
        return blob

    def getBatch(self, split, batch_size=1):
        assert batch_size == 1
        si = self.iterators[split]
        n = len(self.split_ix[split])
        wrapped = False
        ix = self.split_ix[split][self.perm[split][si]]
        si += 1
        if si >= n:
            si = 0
            wrapped = True
            if split == 'train':
                self.perm[split] = np.random.permutation(n)
        self.iterators[split] = si
        blob = dict(self._image(ix))
        blob['bounds'] = {'it_pos_now': si, 'it_max': n, 'wrapped': wrapped}
        if '_device' in self._cache.get(('dev', ix), {}):
            blob['_device'] = self._cache[('dev', ix)]['_device']
        else:
            self._cache[('dev', ix)] = blob          # device copies made by Network.upload_blob stay with the image
        return self._cache[('dev', ix)]

    getTestBatch = getBatch
