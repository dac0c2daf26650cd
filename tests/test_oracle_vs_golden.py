"""Pin the oracle (oracle/*.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import boxes as OB, weights as OW, net as ON
from golden_util import load, check_digest, setup_from_fixture, setup_from_fixture_test, variant_of


def test_anchor_known_answer():
    """generate_anchors.py:14-39 docstring (Matlab, 1-based) minus 1."""
    doc = np.array([[-83, -39, 100, 56], [-175, -87, 192, 104], [-359, -183, 376, 200],
                    [-55, -55, 72, 72], [-119, -119, 136, 136], [-247, -247, 264, 264],
                    [-35, -79, 52, 96], [-79, -167, 96, 184], [-167, -343, 184, 360]], np.float64)
    assert np.array_equal(OB.generate_anchors(), doc - 1)


def test_leaf_boxes():
    g = load('leaf')
    assert np.array_equal(OB.generate_anchors(), g['anchors.base'])
    a, n = OB.generate_anchors_pre(5, 7, 16, (4, 8, 16, 32), (0.5, 1, 2))
    assert np.array_equal(a, g['anchors.pre_5x7'])
    assert np.allclose(OB.bbox_transform(g['bt.ex'], g['bt.gt']), g['bt.targets'], rtol=1e-6, atol=1e-6)
    inv = OB.bbox_transform_inv(g['bt.ex'], g['bt.deltas'])
    assert np.allclose(inv, g['bt.inv'], rtol=1e-6, atol=1e-4)
    assert np.allclose(OB.clip_boxes(inv, (200, 320)), g['bt.clip'], rtol=1e-6, atol=1e-4)
    assert np.allclose(OB.bbox_overlaps(g['bt.ex'], g['bt.gt'][:7]), g['bt.iou'], rtol=1e-6, atol=1e-7)


def test_leaf_language():
    g = load('leaf')
    opt = OW.default_opt(vocab_size=37, seq_length=6)
    sd = OW.make_state_dict(opt, seed=11)
    net = ON.OracleNet(sd, opt)
    with torch.no_grad():
        hid = net.rnn_encoder(torch.from_numpy(g['enc.labels']))
        assert np.allclose(hid.numpy(), g['enc.hidden'], atol=1e-5)
        att = torch.from_numpy(g['cap.att'].astype(np.float32)).view(1, 196, 4096)
        lp = net.caption(att, torch.from_numpy(g['cap.seq']))
        assert np.allclose(lp.numpy(), g['cap.logprobs'], atol=1e-4)


def _run_e2e(tag):
    g = load(tag)
    opt, sd, blob, cfg, samp = setup_from_fixture(g)
    net = ON.OracleNet(sd, opt, cfg, variant=variant_of(g))
    # proposal order / NMS keeps are discontinuous in fp32 scores (near-ties swap rows): compare the
    # oracle's own proposals as a SET with tolerance, then teacher-force the reference's list so that
    # everything downstream (sampling, targets, heads, losses, grads) can be compared tightly.
    samp['forced_proposals'] = (g['int.proposal_rois'], g['int.proposal_scores'])
    T, L = net.forward_train(blob, samp)
    assert T['proposal_rois'].shape == g['int.proposal_rois'].shape
    key = lambda r: r[np.lexsort(np.round(r[:, ::-1] * 8).T)]
    assert np.allclose(key(T['proposal_rois']), key(g['int.proposal_rois']), atol=2e-3)
    assert np.array_equal(T['rpn_labels'].astype(np.int8), g['int.rpn_labels'])
    # column 0 (batch index) of a GT row appended by PTL:159-167 is uninitialised memory in the reference (`.new()`); nothing reads it
    assert np.allclose(T['rois'][:, 1:], g['int.rois'][:, 1:], atol=2e-3)
    assert np.array_equal(T['labels'].reshape(-1).astype(np.int64), g['int.labels'])
    assert np.array_equal(T['mask_targets'].astype(np.uint8), g['int.mask_targets'])
    for k in ['net_conv', 'rpn_cls_prob', 'rpn_bbox_pred', 'cls_score', 'bbox_pred'] + [k for k in ('mask_score', 'response') if 't.%s.sum' % k in g]:
        check_digest(g, 't.' + k, T[k].detach().numpy())
    check_digest(g, 't.rpn_bbox_targets', T['rpn_bbox_targets'])
    check_digest(g, 't.rpn_bbox_outside', T['rpn_bbox_outside'])
    check_digest(g, 't.bbox_targets', T['bbox_targets'])
    for k, v in L.items():
        assert abs(float(v) - float(g['loss.' + k])) < 1e-4 * max(1, abs(float(g['loss.' + k]))), (k, float(v), g['loss.' + k])
    grads = net.backward()
    w0 = {n: net.p[n].detach().clone() for n in net.trainable}
    net.sgd_step()
    names = sorted({k[2:].rsplit('.', 1)[0] for k in g if k.startswith('g.')})
    for n in names:
        check_digest(g, 'g.' + n, grads[n].numpy(), rtol=2e-4, atol=1e-7)
        check_digest(g, 'w1.' + n, net.p[n].detach().numpy(), rtol=1e-5, atol=1e-7)
    # the solver rule spelled out on the reference's own post-step weights (w1.* comes from the optimiser the reference's
    # SolverWrapper.construct_graph() of this variant built): a language-side tensor moves by 10 x lr outside the two cycle solvers
    lr, wd = float(g['solver.LEARNING_RATE']), float(g['solver.WEIGHT_DECAY'])
    mult = OW.SOLVERS[variant_of(g)]['lang_lr_mult']
    for n in names:
        if not any(t in n for t in OW.LANG_KEYS) or 'bias' in n:
            continue
        stride = int(g['w1.' + n + '.stride'])
        step_ref = w0[n].numpy().ravel()[::stride][:2048].astype(np.float64) - g['w1.' + n + '.sample'].astype(np.float64)
        unit = (grads[n] + wd * w0[n]).numpy().ravel()[::stride][:2048].astype(np.float64)          # first step: momentum buffer = d
        big = np.abs(unit) > 0.05 * np.abs(unit).max()
        ratio = np.median(step_ref[big] / unit[big]) / lr
        assert abs(ratio - mult) < 0.15 * mult, (n, ratio, mult)        # (float32 weights: the step is a few ulps of the weight)


def test_train_step_tiny():
    _run_e2e('tiny')


def test_train_step_tiny_pooling_align():
    """cfg.POOLING_ALIGN = True in the reference: _crop_pool_layer_align (NET:151-182) — affine grid from the RoI in image pixels over
    im_info's size, 14x14 crop + 2x2 max pool — in the RoI head, forward and backward."""
    _run_e2e('tiny_align')


def test_train_step_tiny_fixed_blocks_0():
    """cfg.RESNET.FIXED_BLOCKS = 0 in the reference (RES:290-299): layer1 trains as well (conv1 / bn1 stay frozen)."""
    _run_e2e('tiny_fb0')


@pytest.mark.parametrize('variant', ['baseline', 'spatial', 'response', 'cycle_response', 'vgg'])
def test_train_step_tiny_variants(variant):
    """the reference's other ResNet network variants (network.py, network_7f.py, network_7f_response.py,
    network_cycle_response.py): losses (incl. the response BCE), targets, gradients, post-SGD weights."""
    _run_e2e('tiny_' + variant)


@pytest.mark.parametrize('tag', ['full_spatial', 'full_cycle_response', 'full_vgg'])
def test_forward_full_size_variants(tag):
    """BASELINE.json configs 2, 4 and 5 at their stated size (600x1000, 12000 -> 2000 proposals, 256 RoIs; tests/golden/make_golden.py full_variants):
    the oracle's forward pass against the reference's own run - every loss 1e-4, integer targets exact (forward only: the CPU suite stays in minutes;
    the backward pass of every variant is pinned at the tiny size above and, on the device, against these fixtures' gradients)."""
    g = load(tag)
    opt, sd, blob, cfg, samp = setup_from_fixture(g)
    net = ON.OracleNet(sd, opt, cfg, variant=variant_of(g))
    samp['forced_proposals'] = (g['int.proposal_rois'], g['int.proposal_scores'])
    with torch.no_grad():
        T, L = net.forward_train(blob, samp)
    assert np.array_equal(T['rpn_labels'].astype(np.int8), g['int.rpn_labels'])
    assert np.array_equal(T['labels'].reshape(-1).astype(np.int64), g['int.labels'])
    if 'int.mask_targets' in g:
        assert np.array_equal(T['mask_targets'].astype(np.uint8), g['int.mask_targets'])
    for k in ['net_conv', 'cls_score', 'bbox_pred'] + [k for k in ('mask_score', 'response') if 't.%s.sum' % k in g]:
        check_digest(g, 't.' + k, T[k].detach().numpy())
    for k, v in L.items():
        assert abs(float(v) - float(g['loss.' + k])) < 1e-4 * max(1, abs(float(g['loss.' + k]))), (k, float(v), g['loss.' + k])


@pytest.mark.parametrize('tag', ['test_tiny', 'test_tiny_cycle_response', 'test_tiny_top', 'test_tiny_vgg', 'test_tiny_baseline', 'test_tiny_spatial', 'test_tiny_response'])
def test_test_mode(tag):
    """TEST mode of the reference (test_image NET:684-699 + _predict_masks_from_boxes_and_labels NET:595-626):
    300 TEST proposals, class scores / probabilities, de-normalised box deltas, mask probabilities."""
    g = load(tag)
    opt, sd, blob, cfg, _ = setup_from_fixture_test(g)
    net = ON.OracleNet(sd, opt, cfg, variant=variant_of(g))
    out = net.forward_test(blob, forced_proposals=g['int.rois'])
    D = np.abs(out['own_rois'][:, None, 1:] - g['int.rois'][None, :, 1:]).max(-1)
    assert out['own_rois'].shape == g['int.rois'].shape and D.min(1).max() < 5e-3 and D.min(0).max() < 5e-3
    assert np.allclose(out['cls_score'].numpy(), g['x.cls_score'], atol=1e-4)
    assert np.allclose(out['cls_prob'].numpy(), g['x.cls_prob'], atol=1e-6)
    assert np.allclose(out['bbox_pred'].numpy()[:, :24], g['x.bbox_pred'], atol=1e-5)
    check_digest(g, 't.bbox_pred', out['bbox_pred'].numpy())
    check_digest(g, 't.net_conv', out['net_conv'].numpy())
    if 't.mask_prob.sum' not in g:          # VGG16 / Faster R-CNN network: no mask branch
        assert 'mask_prob' not in out
        return
    check_digest(g, 't.mask_prob', out['mask_prob'].numpy())
    pm = net.predict_masks_from_boxes_and_labels(out['net_conv'], g['pm.boxes'], g['pm.labels'])
    assert np.allclose(pm.numpy(), g['pm.masks'], atol=1e-5)


TRAIN_TAGS = ['tiny', 'tiny_align', 'tiny_fb0', 'tiny_baseline', 'tiny_spatial', 'tiny_response', 'tiny_cycle_response', 'tiny_vgg',
              'full', 'full_spatial', 'full_cycle_response', 'full_vgg', 'full_baseline', 'full_response']


@pytest.mark.parametrize('tag', TRAIN_TAGS)
def test_solver_param_groups_vs_reference(tag):
    """every fixture carries the param-group table (key, lr, weight decay) of the optimiser the reference's own SolverWrapper of that
    variant built (make_golden.reference_solver -> construct_graph): the oracle's rule must reproduce it key by key."""
    g = load(tag)
    v = variant_of(g)
    assert str(g['solver.module']) == OW.SOLVERS[v]['module']
    ct = dict(ON.DEFAULT_CFG['TRAIN'], **OW.SOLVERS[v]['cfg'])
    for k in ('LEARNING_RATE', 'MOMENTUM', 'WEIGHT_DECAY', 'GAMMA'):
        assert float(g['solver.' + k]) == float(ct[k]), (k, g['solver.' + k], ct[k])
    assert bool(g['solver.DOUBLE_BIAS']) == bool(ct['DOUBLE_BIAS']) and bool(g['solver.BIAS_DECAY']) == bool(ct['BIAS_DECAY'])
    keys = [str(k) for k in g['solver.keys']]
    assert len(keys) > 50 and len(set(keys)) == len(keys)
    seen_lang = 0
    for k, lr, wd in zip(keys, g['solver.lr'], g['solver.wd']):
        mult, w = OW.param_group(v, k, ON.DEFAULT_CFG['TRAIN'])
        assert abs(ct['LEARNING_RATE'] * mult - float(lr)) <= 1e-12 and abs(w - float(wd)) <= 1e-15, (k, mult, w, lr, wd)
        seen_lang += any(t in k for t in OW.LANG_KEYS)
    assert seen_lang >= 13
    # the oracle trains what the reference trains
    opt, sd, blob, cfg, samp = setup_from_fixture(g)
    onet = ON.OracleNet.__new__(ON.OracleNet); onet.cfg = cfg
    # (resnet.fc: a parameter of the reference's module that no loss reaches - torch.optim.SGD skips it, the synthetic state dict omits it)
    assert sorted(k for k in sd if onet._is_trainable(k)) == sorted(k for k in keys if not k.startswith('resnet.fc.'))
