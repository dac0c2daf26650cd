"""Input side on the GPU, through the C ABI: run-length masks and the image blob bit-exact against the
oracle (oracle/data.py, pinned to the reference's maskApi.c by tests/golden/ref_rle.npz), and the loaders' blobs dict on a tiny
on-disk dataset in the reference's format."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from oracle import data as OD
from data_util import load_rle_fixture, write_tiny_dataset

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _to_mask(O, cnt_list, h, w, oh, ow):
    cnts = np.concatenate(cnt_list).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum([len(c) for c in cnt_list])]).astype(np.int32)
    dc = torch.from_numpy(cnts.view(np.int32)).to(DEV); do = torch.from_numpy(offs).to(DEV)
    ws = torch.empty(O.rle_ws_words(cnts.size, oh, ow), dtype=torch.int32, device=DEV)
    out = torch.full((oh, ow), 7, dtype=torch.uint8, device=DEV)
    O.rle_to_mask(dc, do, len(cnt_list), cnts.size, h, w, ws, out)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_rle_to_mask_matches_reference_fixture():
    from lang2seg_amd import ops as O
    cases, groups = load_rle_fixture()
    for c in cases:                        # plain decode: exactly the mask the reference's rleDecode produced
        got = _to_mask(O, [O.rle_from_string(c['s'])], c['h'], c['w'], c['h'], c['w'])
        assert np.array_equal(got, c['mask']), (c['h'], c['w'])
    rs = np.random.RandomState(0)
    for c in cases:                        # decode + PIL nearest resize (cycle_loader.py:209), up and down
        for _ in range(2):
            oh, ow = int(rs.randint(1, 700)), int(rs.randint(1, 1100))
            ref = OD.ref_mask([{'size': [c['h'], c['w']], 'counts': c['s']}], oh, ow)
            got = _to_mask(O, [O.rle_from_string(c['s'])], c['h'], c['w'], oh, ow)
            assert np.array_equal(got, ref), (c['h'], c['w'], oh, ow)
    for g in groups:                       # objects of several segments: union (cycle_loader.py:205-206)
        h, w = cases[g[0]]['h'], cases[g[0]]['w']
        rles = [{'size': [h, w], 'counts': cases[i]['s']} for i in g]
        for oh, ow in ((h, w), (600, 800), (97, 131)):
            got = _to_mask(O, [O.rle_from_string(cases[i]['s']) for i in g], h, w, oh, ow)
            assert np.array_equal(got, OD.ref_mask(rles, oh, ow))


def test_rle_to_mask_many_runs():
    """an object with more runs than one scan chunk (1024) and runs up to the full image"""
    from lang2seg_amd import ops as O
    rs = np.random.RandomState(3)
    h, w = 480, 640
    m = (rs.rand(h, w) > 0.5).astype(np.uint8)            # ~150k runs
    cnt = OD.rle_encode(m)
    assert cnt.size > 100000
    assert np.array_equal(_to_mask(O, [cnt], h, w, h, w), m)
    assert np.array_equal(_to_mask(O, [cnt], h, w, 600, 800), OD.ref_mask([{'size': [h, w], 'counts': OD.rle_to_string(cnt)}], 600, 800))
    one = np.array([0, h * w], np.uint32)                 # all ones: an empty first run
    assert np.array_equal(_to_mask(O, [one], h, w, 50, 60), np.ones((50, 60), np.uint8))


@pytest.mark.parametrize('hw', [(375, 500), (480, 640), (300, 900), (1200, 900), (64, 48), (600, 1000)])
def test_prep_image_matches_oracle(hw):
    """blob.py:32-47; bit-exact against the oracle's written-out INTER_LINEAR (cv2 itself is not available: parity unpinned, see oracle/data.py)"""
    from lang2seg_amd import ops as O
    from lang2seg_amd.model.config import cfg
    h, w = hw
    rs = np.random.RandomState(h + w)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    ref, rsc = OD.prep_im_for_blob(img, cfg.PIXEL_MEANS, 600, 1000)
    sc, oh, ow = O.prep_geometry(h, w, 600, 1000)
    assert sc == rsc and (oh, ow) == ref.shape[:2]
    out = torch.empty((oh, ow, 3), dtype=torch.float32, device=DEV)
    O.prep_image(torch.from_numpy(img).to(DEV), cfg.PIXEL_MEANS.reshape(-1), sc, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    # sanity of the restatement itself: a constant image stays constant, corners keep the corner pixels
    assert np.allclose(got[0, 0], img[0, 0].astype(np.float64) - cfg.PIXEL_MEANS.reshape(-1), atol=1e-4) or sc < 1


def test_cycle_loader_blobs_match_oracle(tmp_path):
    """CycleLoader.getBatch / getTestBatch on a tiny dataset in the reference's on-disk format: every key of the blobs dict equals
    the oracle's restatement of cycle_loader.py:143-357 (device-resident entries read back), the cursor walks and wraps like the
    reference's, and Network.upload_blob consumes the device entries without a host round trip."""
    from lang2seg_amd.loaders.cycle_loader import CycleLoader, GtMRCNLoader
    from lang2seg_amd.model.config import cfg
    root = str(tmp_path)
    info, labels, imgs = write_tiny_dataset(root)
    kw = dict(image_root=os.path.join(root, 'images'), image_pattern='img_{:0>12d}.png', verbose=False)
    ld = CycleLoader(os.path.join(root, 'data.json'), os.path.join(root, 'data.h5'), **kw)
    assert sorted(ld.split_ix) == ['train', 'val'] and len(ld.split_ix['train']) == 2
    np.random.seed(11)
    seen = []
    for step in range(5):                                   # 2 train images: wraps twice
        it_before = ld.iterators['train']
        b = ld.getBatch('train')
        image_id = [im['image_id'] for im in info['images'] if im['file_name'] == b['file_name']][0]
        seen.append(image_id)
        ref = OD.get_batch(info, labels, image_id, imgs[image_id], cfg.PIXEL_MEANS, cfg.TRAIN.SCALES[0], cfg.TRAIN.MAX_SIZE)
        assert b['bounds']['wrapped'] == (it_before == 1) and b['bounds']['it_max'] == 1
        for k in ('im_info', 'gt_boxes', 'labels', 'cap_labels', 'cap_masks'):
            assert np.array_equal(np.asarray(b[k]), ref[k]), k
        assert b['ref_ids'] == ref['ref_ids']
        assert np.array_equal(b['data'], ref['data'])                      # host access copies the device blob back
        assert np.array_equal(b['gt_masks'], ref['gt_masks'])
        assert b['gt_masks'].dtype == np.uint8 and b['data'].dtype == np.float32
    assert set(seen) == set(ld.split_ix['train'])
    # test batches: sequential cursor, no caption fields
    gl = GtMRCNLoader(os.path.join(root, 'data.json'), os.path.join(root, 'data.h5'), **kw)
    t = gl.getTestBatch('val')
    image_id = ld.split_ix['val'][0]
    ref = OD.get_batch(info, labels, image_id, imgs[image_id], cfg.PIXEL_MEANS, cfg.TRAIN.SCALES[0], cfg.TRAIN.MAX_SIZE, test=True)
    assert 'cap_labels' not in t and t['bounds'] == {'it_pos_now': 0, 'it_max': 0, 'wrapped': True}
    for k in ('im_info', 'gt_boxes', 'labels'):
        assert np.array_equal(np.asarray(t[k]), ref[k]), k
    assert np.array_equal(t['gt_masks'], ref['gt_masks']) and np.array_equal(t['data'], ref['data'])
    g2 = gl.getBatch('train')
    assert 'cap_labels' not in g2 and 'bounds' not in g2


def test_train_step_on_loader_blobs(tmp_path):
    """the train step fed from the loader's device-resident blobs gives the losses of the same step fed from host arrays"""
    from lang2seg_amd import selftest
    from lang2seg_amd.loaders.cycle_loader import CycleLoader
    from lang2seg_amd.optim import SGD
    from oracle import weights as OW
    root = str(tmp_path)
    info, labels, imgs = write_tiny_dataset(root, sizes=((60, 80), (48, 72)))
    from lang2seg_amd.model.config import cfg
    old = (cfg.TRAIN.SCALES, cfg.TRAIN.MAX_SIZE)
    cfg.TRAIN.SCALES, cfg.TRAIN.MAX_SIZE = (160,), 256
    try:
        ld = CycleLoader(os.path.join(root, 'data.json'), os.path.join(root, 'data.h5'), image_root=os.path.join(root, 'images'),
                         image_pattern='img_{:0>12d}.png', verbose=False)
        opt = OW.default_opt(vocab_size=ld.vocab_size, seq_length=ld.label_length)
        sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
        over = dict(BATCH_SIZE=16, RPN_PRE_NMS_TOP_N=600, RPN_POST_NMS_TOP_N=100, RPN_BATCHSIZE=64)
        b = ld.getBatch('train')
        host = {k: (np.array(b[k]) if isinstance(b[k], np.ndarray) else b[k]) for k in ('data', 'gt_masks', 'im_info', 'gt_boxes', 'labels', 'cap_labels', 'cap_masks')}
        res = []
        for blobs in (b, host):
            net = selftest.build_net(opt, over, 'f32', sd)
            res.append([net.train_step(blobs, i, SGD(net, 0.0)) for i in range(len(b['labels']))])
        assert len(res[0]) >= 2
        for a, c in zip(res[0], res[1]):
            assert np.allclose(a, c, rtol=1e-5, atol=1e-6), (a, c)
    finally:
        cfg.TRAIN.SCALES, cfg.TRAIN.MAX_SIZE = old


def test_eval_split_on_loader(tmp_path):
    """model/test.py eval_split driven by GtMRCNLoader.getTestBatch on the tiny on-disk dataset: same metrics whether the loop reads the
    loader's device-resident blobs or host arrays of the same batch (the reference's contract)."""
    from lang2seg_amd import selftest
    from lang2seg_amd.loaders.cycle_loader import GtMRCNLoader
    from lang2seg_amd.model.config import cfg
    from lang2seg_amd.model.test import eval_split
    from oracle import weights as OW
    root = str(tmp_path)
    write_tiny_dataset(root, sizes=((120, 160), (90, 150), (100, 140), (128, 128)))
    old = (cfg.TRAIN.SCALES, cfg.TRAIN.MAX_SIZE)
    cfg.TRAIN.SCALES, cfg.TRAIN.MAX_SIZE = (240,), 400
    try:
        mk = lambda: GtMRCNLoader(os.path.join(root, 'data.json'), os.path.join(root, 'data.h5'), image_root=os.path.join(root, 'images'),
                                  image_pattern='img_{:0>12d}.png', verbose=False)
        ld = mk()
        opt = OW.default_opt(vocab_size=ld.vocab_size, seq_length=ld.label_length)
        net = selftest.build_net(opt, {}, 'f32', OW.make_state_dict(opt, seed=3, head_gain=4.0))
        r_dev = eval_split(ld, net, None, 'val', dict(verbose=False))

        class HostView(object):            # the same batches as plain host dicts
            def __init__(self, inner):
                self.inner, self.split_ix = inner, inner.split_ix
            def getTestBatch(self, split):
                b = self.inner.getTestBatch(split)
                return {k: b[k] for k in ('data', 'gt_masks', 'im_info', 'gt_boxes', 'labels', 'file_name', 'bounds')}
        r_host = eval_split(HostView(mk()), net, None, 'val', dict(verbose=False))
        # (acc, thresholds, seg_correct, seg_total, cum_I, cum_U, num_sent): integer counts, identical either way
        assert r_dev[0] == r_host[0] and list(r_dev[2]) == list(r_host[2]) and r_dev[3:] == r_host[3:]
        assert r_dev[6] == r_dev[3] > 0 and 0 <= r_dev[4] <= r_dev[5]
    finally:
        cfg.TRAIN.SCALES, cfg.TRAIN.MAX_SIZE = old
