"""End-to-end parity of the HIP train step (forward, losses, backward, SGD) against the CPU oracle
and the reference-generated golden fixture, exact-f32 verification mode (tolerance 1e-4 as stated in
BASELINE.json north_star) plus a bf16 run with a loose tolerance."""
import copy
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from golden_util import load, check_digest, setup_from_fixture, variant_of

NAMES = ['rpn_cross_entropy', 'rpn_loss_box', 'cross_entropy', 'loss_box', 'loss_mask', 'loss_caption', 'total_loss']


def _setup(dtype, tag='tiny'):
    from lang2seg_amd import selftest
    g = load(tag)
    opt, sd, blob, ocfg, samp = setup_from_fixture(g)
    samp['forced_proposals'] = (g['int.proposal_rois'], g['int.proposal_scores'])
    over = {k[4:]: int(g[k]) for k in g if k.startswith('cfg.')}
    from lang2seg_amd.model.config import cfg
    for k in g:                                  # (before the network is built: FIXED_BLOCKS decides what is trainable; reset by conftest)
        if k.startswith('resnet.'):
            cfg.RESNET[k[7:]] = int(g[k])
    net = selftest.build_net(opt, over, dtype, sd, variant=variant_of(g))
    for k in g:
        if k.startswith('top.'):                 # e.g. POOLING_ALIGN (reset by conftest after the test)
            cfg[k[4:]] = bool(int(g[k]))
    net.parity = selftest.parity_from_samp(samp)
    return g, opt, sd, blob, ocfg, samp, net


# Tolerances.  f32 ("verification mode", exact-f32 MFMA): north_star's 1e-4 on losses and seg-logits, integer outputs bit-exact.
# bf16 (the benchmarked mode: bf16 activations / weight shadows, fp32 accumulation, fp32 master weights): losses within 1e-2
# relative; EVERY trainable tensor's gradient has cosine >= 0.99 with the f32 step's gradient of the same inputs on the device (which
# the f32 leg pins to the reference at 5e-4) and a norm within 5 % of it at the BASELINE size, 25 % on the tiny fixtures.  Measured
# (profiles/r02_bf16_grad_agreement.jsonl, 150 tensors per run): full size cosine >= 0.9978, norm within 2.7 %; tiny fixtures (head_gain 4)
# cosine >= 0.9903, norms within 3 % except the dynamic-filter FCs of the cycle fixture (6 %, one at 20 %: their gradient goes through
# d(response)[p] = <dy[p], x[p]>, a 1024-term dot product of bf16 values that nearly cancels); integer outputs still bit-exact (they depend on the boxes, not on the activations, once the
# proposals are teacher-forced).
BF16_LOSS_RTOL, BF16_COS, BF16_NORM = 1e-2, 0.99, 0.10
# Round 6: the tiny-fixture gates are the general ones (cosine >= 0.99, norm within 10 %: measured over the eight tiny fixtures, 59-160 tensors each,
# profiles/r06_bf16_grad_agreement.jsonl: cosine >= 0.9947, norms within 7.6 %) with TWO named exceptions instead of a wide gate for a whole family:
#   dynamic_fc_5 (weight and bias) of the CYCLE network's tiny fixtures (tiny, tiny_align, tiny_fb0): cosine 0.985-0.998, norm 0.76-0.89 of the f32 step's.  Its
#     gradient is sum_p dresp[p] x[:, p] with dresp[p] = <dy[p], x[p]>, a 1024-term dot product of bf16 values that nearly cancels; on 520 pixels a handful of them
#     carries the sum and its length re-rolls with every rounding pattern upstream (0.89 before layer1 was fused, 0.76 after).  The same tensor at the BASELINE size:
#     within 2.3 %; every other dynamic-filter tensor of every fixture: within 7.6 %.
#   resnet.layer1.* at FIXED_BLOCKS = 0: cosine 0.987 on layer1.0.conv1 - the deepest gradient of the step, behind 33 blocks of bf16 activations.
BF16_EXCEPTIONS = {('cycle', 'dynamic_fc_5.weight'): (0.97, 0.30), ('cycle', 'dynamic_fc_5.bias'): (0.97, 0.30)}
BF16_NORM_DYN_FULL = 0.10      # dynamic-filter FCs at the BASELINE size (measured <= 0.03)
VARIANT_TAGS = ['tiny', 'tiny_baseline', 'tiny_spatial', 'tiny_response', 'tiny_cycle_response', 'tiny_vgg', 'tiny_align', 'tiny_fb0']


def _grad_of(net, nme):
    from lang2seg_amd.nets.params import from_internal
    P = net.P
    gr = from_internal(nme, P.view(nme, P.grad).clone(), P.shapes[nme])
    if nme in P.rowscale_off:                       # stored gradient is w.r.t. the BN-folded weight
        gr = gr * P.bn_scale[nme].view(-1, *([1] * (gr.dim() - 1)))
    return gr.cpu().numpy()


def _check_grads(g, net, dtype, rtol_f32, ref_net=None, norm_tol=None):
    """f32: gradients of the fixture's tensors (reference layout) against the reference run: digest within rtol + relative L2 error (which,
    unlike the max-normalised digest, weighs the small entries too).
    bf16: EVERY trainable tensor, whole, against the f32 step of the same inputs on the device (`ref_net`, itself pinned to the reference by
    the f32 leg): cosine and norm ratio.  (The fixtures hold strided samples; a sample that happens to sit on small entries of a
    heavy-tailed tensor measures bf16's noise floor, not the tensor.)"""
    from golden_util import digest_metrics
    names = sorted({k[2:].rsplit('.', 1)[0] for k in g if k.startswith('g.')})
    bad, table = [], {}
    if dtype == 'f32':
        for nme in names:
            gr = _grad_of(net, nme)
            cos, emax, el2, p99 = digest_metrics(g, 'g.' + nme, gr)
            table[nme] = [round(cos, 6), round(emax, 6), round(el2, 6)]
            check_digest(g, 'g.' + nme, gr, rtol=rtol_f32, atol=1e-7)
            if not (el2 <= 4 * rtol_f32 and cos >= 1 - 1e-5):
                bad.append((nme, cos, emax, el2))
    else:
        P, Pr = net.P, ref_net.P
        gmax = max(float(Pr.view(k, Pr.grad).double().norm()) for k in Pr.trainable)
        for k in P.trainable:
            a, b = Pr.view(k, Pr.grad).double(), P.view(k, P.grad).double()
            na, nb = float(a.norm()), float(b.norm())
            if na <= 1e-7 * gmax:                      # e.g. alpha_net.bias: softmax is shift-invariant, its gradient is rounding noise
                continue
            cos = float((a * b).sum() / (na * nb + 1e-300))
            table[k] = [round(cos, 6), round(nb / na, 6)]
            # The dynamic-filter FCs: their gradient is sum_p dresp[p] x[:, p] with dresp[p] = <dy[p], x[p]>, a 1024-term dot product of bf16
            # values that nearly cancels, and a handful of pixels dominates the sum - the DIRECTION is theirs (cosine 0.9999), the LENGTH
            # re-rolls with the rounding pattern upstream.  On the tiny fixtures (520 pixels) that is worth up to 20 % (BF16_NORM); at the
            # BASELINE size (2394 pixels) the committed runs measure <= 2 % (profiles/r03_bf16_grad_agreement.jsonl, r04: <= 3 %), and the
            # gate there is 10 % for these tensors, `norm_tol` (5 %) for every other one.
            dyn = k.startswith(('dynamic_fc_', 'response_fc'))
            # BASELINE size (norm_tol given): norm_tol for every tensor, 10 % for the dynamic-filter FCs (measured <= 2.8 %).  Tiny fixtures: BF16_NORM for
            # every tensor but the named exceptions above
            ntol = BF16_NORM if norm_tol is None else (max(norm_tol, BF16_NORM_DYN_FULL) if dyn else norm_tol)
            ctol = BF16_COS
            if norm_tol is None:
                ex = BF16_EXCEPTIONS.get((variant_of(g), k))
                if ex is not None:
                    ctol, ntol = ex
                elif k.startswith('resnet.layer1.'):
                    ctol = 0.98
            if not (cos >= ctol and abs(nb / na - 1.0) <= ntol):
                bad.append((k, cos, nb / na))
    _log_grad_table(g, dtype, table)
    assert not bad, bad
    return names


def _log_grad_table(g, dtype, table):
    """append the per-tensor gradient agreement of this run to gpurun_out/grad_agreement.jsonl (evidence for the stated tolerances)"""
    import json, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'grad_agreement.jsonl'), 'a') as f:
            f.write(json.dumps({'variant': variant_of(g), 'H': int(g['meta_H']), 'dtype': dtype,
                                ('cos_maxerr_l2err_vs_reference_sample' if dtype == 'f32' else 'cos_normratio_vs_f32_device'): table}) + '\n')
    except OSError:
        pass


# bf16 proposal lists against the reference's: IoU-matched recall of the reference's boxes.  Measured on the MI355X (round 6,
# gpurun_out/proposal_agreement.jsonl -> profiles/r06_proposal_agreement.jsonl), recall at IoU >= 0.7 / >= 0.9:
#   tiny .985/.858  tiny_spatial .983/.950  tiny_response .970/.923  tiny_cycle_response .980/.857  tiny_vgg .977/.780  tiny_align .985/.858  tiny_fb0 .985/.855
#   full .991/.958  full_spatial .994/.941  full_cycle_response .993/.842
# and the outliers: tiny_baseline .577/.300 (the baseline network's RPN scores all sit within 1e-3 of each other on these weights: the
# top-1500 cut and the greedy scan pick a different, equally valid subset once bf16 reorders them) and full_vgg .536/.312 (the un-normalised
# 13-convolution VGG trunk lets bf16 activations drift furthest; 61 % of its boxes are still within 4 px of a reference box).
# Gates = those measurements less a margin; everything downstream of the list is teacher-forced in these tests, and the f32 legs compare the lists box by box.
#   full_baseline .649/.206 (the same regime as tiny_baseline, at the BASELINE size), full_response: inside the default gate
PROPOSAL_GATES = {'tiny_baseline': (0.50, 0.25), 'full_baseline': (0.55, 0.15), 'full_vgg': (0.45, 0.25)}
PROPOSAL_GATE_DEFAULT = (0.95, 0.75)


def _proposal_gate(tag, pa, n, n_ref):
    g70, g90 = PROPOSAL_GATES.get(tag, PROPOSAL_GATE_DEFAULT)
    assert abs(n - n_ref) <= 0.1 * n_ref and pa['recall_iou70'] >= g70 and pa['recall_iou90'] >= g90, (tag, pa)


def _proposal_agreement(tag, dtype, mine, ref):
    """how well the device's own proposal list (bf16: scores from bf16 activations reorder near-ties at the 12 000 cut and inside the greedy
    scan) describes the same boxes as the reference's: IoU-matched recall of the reference's boxes at IoU >= 0.9 / 0.7 (+1 areas, as
    utils/bbox.py:21-29) and the share of device boxes within 4 px of a reference box.  Logged per fixture (gpurun_out/proposal_agreement.jsonl)."""
    import json, os
    a, b = mine[:, 1:5].astype(np.float64), ref[:, 1:5].astype(np.float64)
    iw = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1, 0, None)
    ih = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1, 0, None)
    aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1); ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    iou = iw * ih / (aa[:, None] + ab[None, :] - iw * ih)
    best_for_ref = iou.max(0)
    D = np.abs(a[:, None] - b[None, :]).max(-1)
    out = dict(tag=tag, dtype=dtype, n_device=int(a.shape[0]), n_reference=int(b.shape[0]), recall_iou90=float((best_for_ref >= 0.9).mean()),
               recall_iou70=float((best_for_ref >= 0.7).mean()), precision_iou90=float((iou.max(1) >= 0.9).mean()), within_4px=float((D.min(1) < 4.0).mean()))
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'proposal_agreement.jsonl'), 'a') as f:
            f.write(json.dumps(out) + '\n')
    except OSError:
        pass
    return out


def _f32_reference_step(tag):
    """the f32 step of the same fixture inputs on the device (gradients left in net.P.grad)"""
    g, opt, sd, blob, ocfg, samp, net = _setup('f32', tag)
    net.forward_backward(net.upload_blob(blob, 0))
    torch.cuda.synchronize()
    return net


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('tag', VARIANT_TAGS)
def test_train_step_vs_fixture_and_oracle(tag, dtype):
    """every network variant of the reference (cycle = the benchmarked one; baseline / spatial / response / cycle_response / vgg /
    POOLING_ALIGN are BASELINE.json configs 0-4 + train_response.sh) against a fixture produced by the reference itself, in the
    exact-f32 verification mode AND in bf16, the mode bench.py measures."""
    from oracle import net as ON
    from lang2seg_amd.optim import SGD
    from lang2seg_amd.nets.variants import loss_names, SLOT
    g, opt, sd, blob, ocfg, samp, net = _setup(dtype, tag)
    f32 = dtype == 'f32'
    dev = net.upload_blob(blob, 0)
    loss = net.forward_backward(dev)
    torch.cuda.synchronize()
    lv = loss.cpu().numpy()
    t = net.t
    # the device's own proposals as a set vs the reference's
    n = int(t['proposal_n'].item())
    mine = t['proposal_rois'].cpu().numpy()[:n]
    ref = g['int.proposal_rois']
    if f32:
        assert mine.shape == ref.shape
        # set comparison by nearest neighbour (sorting rows is not stable under 1e-4 px perturbations)
        D = np.abs(mine[:, None, 1:] - ref[None, :, 1:]).max(-1)
        assert D.min(1).max() < 2e-2 and D.min(0).max() < 2e-2, (D.min(1).max(), D.min(0).max())
    else:
        # bf16 scores reorder near-ties, so a few keeps differ; the lists must still describe the same boxes
        _proposal_gate(tag, _proposal_agreement(tag, dtype, mine, ref), n, ref.shape[0])
    # integer outputs: bit-exact (proposals teacher-forced)
    assert np.array_equal(t['rpn_labels'].cpu().numpy().astype(np.int8), g['int.rpn_labels'].reshape(-1))
    assert np.array_equal(t['labels'].cpu().numpy().astype(np.int64), g['int.labels'])
    nfg = int(t['counts'][0].item())
    assert nfg == int(g['int.num_fg'])
    assert np.array_equal(t['mask_targets'].cpu().numpy()[:nfg].reshape(nfg, 14, 14).astype(np.uint8), g['int.mask_targets'])
    # (column 0 of a GT row appended by PTL:159-167 is uninitialised memory in the reference)
    assert np.allclose(t['rois'].cpu().numpy()[:, 1:], g['int.rois'][:, 1:], atol=1e-4)
    # losses vs the reference run (fixture)
    ltol = 1e-4 if f32 else BF16_LOSS_RTOL
    for k in loss_names(variant_of(g)):
        i = SLOT[k]
        assert abs(lv[i] - float(g['loss.' + k])) < ltol * max(1.0, abs(float(g['loss.' + k]))), (dtype, k, lv[i], g['loss.' + k])
    assert len(net._loss_slots()) == len(loss_names(variant_of(g)))
    Hc, Wc = 20, 26
    atol = 1e-4 if f32 else 3e-2
    nc = t['net_conv'].float().cpu().view(1, Hc, Wc, -1).permute(0, 3, 1, 2)
    check_digest(g, 't.net_conv', nc.numpy(), rtol=atol, atol=atol)
    heads = t['rcnn_heads'].cpu().numpy()
    # class scores / box deltas (NET:277-290): 1e-4 of the tensor's scale in f32
    hs = max(1.0, float(np.abs(g['x.cls_score']).max()))
    assert np.abs(heads[:, :8] - g['x.cls_score']).max() <= atol * hs, (np.abs(heads[:, :8] - g['x.cls_score']).max(), hs)
    assert np.abs(heads[:8, 81:97] - g['x.bbox_pred']).max() <= atol * max(1.0, float(np.abs(g['x.bbox_pred']).max()))
    check_digest(g, 't.cls_score', heads[:, :81], rtol=atol, atol=atol)
    check_digest(g, 't.bbox_pred', heads[:, 81:81 + 324], rtol=atol, atol=atol)
    if 't.mask_score.sum' in g and t.get('mask_score') is not None:
        # seg-logits (NET:292-307) on the first num_fg RoIs, reference layout (nfg, 81, 14, 14)
        ms = t['mask_score'].cpu().numpy().reshape(-1, 14, 14, 81)[:nfg].transpose(0, 3, 1, 2)
        check_digest(g, 't.mask_score', ms, rtol=atol, atol=atol)
    # gradients (reference layout) and post-SGD weights vs the fixture
    # (the VGG trunk's gradients cross 9 un-normalised 3x3 convolutions and two max-pools: 1e-3 like the full-size test)
    names = _check_grads(g, net, dtype, 1e-3 if tag == 'tiny_vgg' else 5e-4, ref_net=None if f32 else _f32_reference_step(tag))
    # the optimiser of this variant's solver (model.train_val.make_optimizer: param groups of the reference's construct_graph(), lr x 10 on
    # the language side outside the cycle solvers, config_vgg's WEIGHT_DECAY / DOUBLE_BIAS); w1.* in the fixture is what the reference's own
    # SolverWrapper of the variant left after optimizer.step()
    from lang2seg_amd.model.train_val import make_optimizer
    w0 = {nme: net.state_dict()[nme].numpy().copy() for nme in names if any(s_ in nme for s_ in ('rnn_encoder', 'dynamic_fc', 'response'))}
    g0 = {nme: np.asarray(_grad_of(net, nme)).copy() for nme in w0}              # (the update clears the gradients it consumes)
    sgd = make_optimizer(net)
    assert sgd.lr == float(g['solver.LEARNING_RATE']) and sgd.weight_decay == float(g['solver.WEIGHT_DECAY']) and sgd.momentum == float(g['solver.MOMENTUM'])
    sgd.step()
    torch.cuda.synchronize()
    sd1 = net.state_dict()
    if f32:
        # spelled out: a language-side weight moves by lang_lr_mult x lr x (g + wd w) in the first step
        from lang2seg_amd.nets.variants import SOLVERS
        mult = SOLVERS[variant_of(g)]['lang_lr_mult']
        P = net.P
        for nme, w_old in w0.items():
            if 'bias' in nme:
                continue
            gr = g0[nme]
            unit = (gr + sgd.weight_decay * w_old).ravel().astype(np.float64)
            step = (w_old - sd1[nme].numpy()).ravel().astype(np.float64)
            big = np.abs(unit) > 0.05 * np.abs(unit).max()
            ratio = float(np.median(step[big] / unit[big])) / sgd.lr
            assert abs(ratio - mult) < 0.15 * mult, (nme, ratio, mult)
    for nme in names:
        # f32: 1e-5 of the weight scale.  bf16: the update is lr x gradient, and the gradient of a tensor may deviate from the f32 step's by
        # what _check_grads allows (cosine >= BF16_COS, norm within BF16_NORM): that deviation times the learning rate is allowed on top - it
        # matters for the tensors with large gradients (the dynamic-filter FC bias of the baseline network: the sum over its 1024 entries moved
        # past 1e-5 of the weight scale when the round-3 tiles changed the rounding pattern upstream)
        lr_k = sgd.lr * net.P.param_group(nme, *net.P.seg_rule[:2])[0]        # this tensor's own learning rate (10 x lr on the language side of four solvers)
        dev_ = 0.0 if f32 else lr_k * (BF16_NORM + (2 * (1 - BF16_COS)) ** 0.5)
        check_digest(g, 'w1.' + nme, sd1[nme].numpy(), rtol=1e-5, atol=1e-7,
                     extra_sample=dev_ * float(np.abs(g['g.' + nme + '.sample']).max()), extra_sum=dev_ * float(g['g.' + nme + '.abssum']))


def test_train_step_per_token_captioner_fallback():
    """The three-launches-per-token form of the captioner recurrence (lang.hip) is what a network whose shapes the resident launches do not take
    (rnn_size / att_hid_size != 512, > 224 attention locations) falls back to; every fixture has the resident shapes, so the fallback is run
    here explicitly (Network.cap_persistent = False) on the tiny cycle fixture: same losses and caption-side gradients as the reference run, and
    within fp32 rounding of the resident form."""
    from lang2seg_amd.nets.network import Network
    was = Network.cap_persistent
    try:
        out = {}
        for resident in (False, True):
            Network.cap_persistent = resident
            g, opt, sd, blob, ocfg, samp, net = _setup('f32', 'tiny')
            lv = net.forward_backward(net.upload_blob(blob, 0)).cpu().numpy()
            torch.cuda.synchronize()
            assert bool(net.t.get('cap.resident')) == resident
            for i, k in enumerate(NAMES):
                assert abs(lv[i] - float(g['loss.' + k])) < 1e-4 * max(1.0, abs(float(g['loss.' + k]))), (resident, k, lv[i], g['loss.' + k])
            out[resident] = {k: _grad_of(net, k) for k in ('caption_model.core.h2h.weight', 'caption_model.core.attention.h2att.weight', 'caption_model.att_embed.0.weight')}
        for k in out[True]:
            a_, b_ = out[True][k], out[False][k]
            assert np.abs(a_ - b_).max() <= 1e-4 * max(1e-12, np.abs(b_).max()), k
    finally:
        Network.cap_persistent = was


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('tag', ['full', 'full_spatial', 'full_cycle_response', 'full_vgg', 'full_baseline', 'full_response'])
def test_train_step_full_size(tag, dtype):
    """Every BASELINE.json GPU config at its stated size (600x1000, 12000->2000 proposals, 256 RoIs; config 3 `full` = the headline with 20 tokens,
    V=3349; config 2 `full_spatial` and config 5 `full_vgg` with 10 tokens, V=1999; config 4 `full_cycle_response` with 20 tokens, V=3349) against the
    reference's own run of that variant (tests/golden/make_golden.py full | full_variants), f32 and bf16."""
    from lang2seg_amd import selftest
    from lang2seg_amd.nets.variants import loss_names, SLOT
    g = load(tag)
    opt, sd, blob, ocfg, samp = setup_from_fixture(g)
    samp['forced_proposals'] = (g['int.proposal_rois'], g['int.proposal_scores'])
    over = {k[4:]: int(g[k]) for k in g if k.startswith('cfg.')}
    net = selftest.build_net(opt, over, dtype, sd, variant=variant_of(g))
    net.parity = selftest.parity_from_samp(samp)
    lv = net.forward_backward(net.upload_blob(blob, 0)).cpu().numpy()
    t = net.t
    f32 = dtype == 'f32'
    n = int(t['proposal_n'].item())
    # the device's own 28 728 -> 12 000 -> NMS -> 2000 list against the reference's, as a set (two independent fp32 conv stacks
    # order near-tied scores differently; the bit-exact index check on identical inputs is test_kernels_gpu.py::test_sort_nms_full_size)
    mine = t['proposal_rois'].cpu().numpy()[:n]
    ref = g['int.proposal_rois']
    D = np.abs(mine[:, None, 1:] - ref[None, :, 1:]).max(-1)
    if f32:
        assert n == ref.shape[0]
        far_m, far_r = (D.min(1) > 2e-2), (D.min(0) > 2e-2)
        # boxes present in one list only: a swap of two near-tied scores at the 12 000 cut or in the greedy scan changes a few keeps
        assert far_m.sum() <= 0.01 * n and far_r.sum() <= 0.01 * n, (int(far_m.sum()), int(far_r.sum()))
    else:
        # (bf16 scores reorder near-ties: measured 0.78 ... 0.9 of the device's boxes within 4 px of a reference box over the four fixtures)
        # and 0.61 on the VGG trunk, whose un-normalised 3x3 stack lets bf16 activations drift furthest; everything downstream is teacher-forced)
        _proposal_gate(tag, _proposal_agreement(tag, dtype, mine, ref), n, ref.shape[0])
    assert np.array_equal(t['rpn_labels'].cpu().numpy().astype(np.int8), g['int.rpn_labels'].reshape(-1))
    assert np.array_equal(t['labels'].cpu().numpy().astype(np.int64), g['int.labels'])
    nfg = int(t['counts'][0].item())
    assert nfg == int(g['int.num_fg'])
    if 'int.mask_targets' in g and t.get('mask_targets') is not None:
        assert np.array_equal(t['mask_targets'].cpu().numpy()[:nfg].reshape(nfg, 14, 14).astype(np.uint8), g['int.mask_targets'])
    ltol = 1e-4 if f32 else BF16_LOSS_RTOL
    for k in loss_names(variant_of(g)):
        i = SLOT[k]
        assert abs(lv[i] - float(g['loss.' + k])) < ltol * max(1.0, abs(float(g['loss.' + k]))), (tag, dtype, k, lv[i], g['loss.' + k])
    atol = 1e-4 if f32 else 3e-2
    heads = t['rcnn_heads'].cpu().numpy()
    check_digest(g, 't.cls_score', heads[:, :81], rtol=atol, atol=atol)
    check_digest(g, 't.bbox_pred', heads[:, 81:81 + 324], rtol=atol, atol=atol)
    if 't.mask_score.sum' in g and t.get('mask_score') is not None:
        ms = t['mask_score'].cpu().numpy().reshape(-1, 14, 14, 81)[:nfg].transpose(0, 3, 1, 2)
        check_digest(g, 't.mask_score', ms, rtol=atol, atol=atol)
    assert np.abs(heads[:, :8] - g['x.cls_score']).max() <= atol * max(1.0, float(np.abs(g['x.cls_score']).max()))
    _check_grads(g, net, dtype, 1e-3, ref_net=None if f32 else _f32_reference_step(tag), norm_tol=0.05)


def test_gradients_are_bit_reproducible():
    """two steps from the same weights and inputs give bit-identical gradient buffers and losses' inputs: no floating-point atomics are
    left on the gradient path (grouped weight gradients own whole output tiles or sum their slabs in a fixed order; column sums, the
    embedding / dynamic-filter / attention / mask-head reductions have a single owner per output).  bf16, multi-stream, lr 0."""
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    blob = OS.make_blob(320, 416, 6, 60, seed=5)
    blob['labels'][0, 3] = blob['labels'][0, 1]                # a token that occurs twice (embedding rows with two contributions)
    blob['cap_labels'][0, 4] = blob['cap_labels'][0, 2]
    over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
    for variant in ('cycle', 'cycle_response'):
        net = selftest.build_net(opt, over, 'bf16', sd, variant=variant)
        # fixed sampling keys and no dropout (the device RNG counter advances between steps otherwise)
        rs = np.random.RandomState(0)
        nA = 20 * 26 * 12
        net.parity = selftest.parity_from_samp(dict(rpn_fg_keys=rs.permutation(nA).astype(np.uint32), rpn_bg_keys=rs.permutation(nA).astype(np.uint32),
                                                    roi_fg_keys=rs.permutation(300).astype(np.uint32), roi_bg_keys=rs.permutation(300).astype(np.uint32)))
        sgd = SGD(net, 0.0, keep_grad=True)
        grads = []
        for _ in range(3):
            net.train_step(dict(blob), 0, sgd)
            torch.cuda.synchronize()
            grads.append(net.P.grad.clone())
        assert float(grads[0].abs().sum()) > 0
        for gk in grads[1:]:
            assert torch.equal(gk, grads[0]), (variant, int((gk != grads[0]).sum()))


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


@pytest.mark.parametrize('tag', ['test_tiny', 'test_tiny_cycle_response', 'test_tiny_top', 'test_tiny_vgg', 'test_tiny_baseline', 'test_tiny_spatial', 'test_tiny_response'])
def test_test_mode(tag):
    """TEST mode (test_image, _predict_masks_from_boxes_and_labels) against the reference's own TEST-mode outputs."""
    from golden_util import setup_from_fixture_test
    from lang2seg_amd import selftest
    g = load(tag)
    opt, sd, blob, ocfg, _ = setup_from_fixture_test(g)
    net = selftest.build_net(opt, {}, 'f32', sd, variant=variant_of(g))
    from lang2seg_amd.model.config import cfg
    saved_test = {k: cfg.TEST[k] for k in list(ocfg['TEST']) + ['MODE', 'RPN_TOP_N']}
    for k, v in ocfg['TEST'].items():
        cfg.TEST[k] = v                 # 'test_tiny_top': TEST.MODE = 'top' (proposal_top_layer.py), RPN_TOP_N = 200
    net.parity = dict(forced_proposals=(torch.from_numpy(g['int.rois']).cuda(), None))
    tb = {k: blob[k] for k in ('data', 'im_info', 'gt_boxes', 'gt_masks', 'labels')}
    try:
        cls_score, cls_prob, bbox_pred, rois, net_conv = net.test_image(tb)
    finally:
        for k, v in saved_test.items():
            cfg.TEST[k] = v
    own = net._predictions['own_rois'].cpu().numpy()
    D = np.abs(own[:, None, 1:] - g['int.rois'][None, :, 1:]).max(-1)
    assert own.shape == g['int.rois'].shape and D.min(1).max() < 2e-2 and D.min(0).max() < 2e-2
    assert np.allclose(rois, g['int.rois'], atol=1e-5)
    assert np.allclose(cls_score, g['x.cls_score'], atol=2e-4)
    assert np.allclose(cls_prob, g['x.cls_prob'], atol=1e-5)
    assert np.allclose(bbox_pred[:, :24], g['x.bbox_pred'], atol=1e-4)
    check_digest(g, 't.bbox_pred', bbox_pred, rtol=2e-4)
    if 't.mask_prob.sum' not in g:          # VGG16 / Faster R-CNN network: boxes only (network_vgg.py:614)
        assert 'mask_prob' not in net._predictions
        with pytest.raises(NotImplementedError):
            net._predict_masks_from_boxes_and_labels(net_conv, g['int.rois'][:2, 1:], np.array([1, 2]))
        return
    mp = net._predictions['mask_prob'].cpu().numpy().transpose(0, 3, 1, 2)      # (n, 81, 14, 14) like the reference
    check_digest(g, 't.mask_prob', mp, rtol=2e-4)
    assert np.allclose(mp[:4, :6], g['x.mask_prob_0'], atol=1e-4)
    pm = net._predict_masks_from_boxes_and_labels(net_conv, g['pm.boxes'], g['pm.labels']).cpu().numpy()
    assert np.allclose(pm, g['pm.masks'], atol=1e-4)


def test_eval_split_runs():
    """evaluation loop (model/test.py eval_split) end to end on the synthetic loader: TEST-mode network on the HIP kernels +
    host post-processing; the metrics of an untrained network are only checked for sanity."""
    from lang2seg_amd import selftest
    from lang2seg_amd.model.test import eval_split, summarize
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    from oracle import weights as OW
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    net = selftest.build_net(opt, {}, 'bf16', sd)
    loader = SyntheticLoader(num_images=2, sents_per_image=2, H=320, W=416, T=6, vocab_size=60)
    acc, thr, seg_correct, seg_total, cum_I, cum_U, num_sent = eval_split(loader, net, None, 'val', dict(verbose=False))
    text, prec, iou = summarize(thr, seg_correct, seg_total, cum_I, cum_U)
    assert num_sent == seg_total == 4 and thr == [.5, .6, .7, .8, .9] and 0 <= cum_I <= cum_U
    assert 0.0 <= acc <= 1.0 and 0.0 <= iou <= 1.0 and len(prec) == 5 and all(0.0 <= p <= 1.0 for p in prec)
    assert text.count('precision@') == 5 and 'overall IoU' in text


def test_eval_split_vgg_vs_reference():
    """model/test_vgg.py eval_split (boxes only) against the reference's own loop (model/test_vgg.py:185-460 run through the harness,
    tests/golden/make_golden.py eval_split_vgg -> ref_eval_split_vgg.npz): chosen (RoI, class), predicted boxes, box accuracy."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_golden as MG
    from lang2seg_amd import selftest
    from lang2seg_amd.model import test as T, test_vgg as TVG
    from oracle import weights as OW
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_eval_split_vgg.npz')))
    opt = OW.default_opt(vocab_size=60, seq_length=6); opt['C4_feat_dim'] = 512
    net = selftest.build_net(opt, {}, 'f32', MG.eval_state_dict_vgg(opt), variant='vgg')
    imgs = MG.eval_blobs()

    class Loader(object):
        split_ix = {'val': [0, 1]}

        def __init__(self):
            self.i = 0

        def getTestBatch(self, split):
            b = dict(imgs[self.i]); self.i += 1
            b['bounds'] = dict(it_pos_now=self.i, it_max=len(imgs), wrapped=self.i >= len(imgs))
            return b
    picked = []
    orig = TVG.best_detection

    def rec(scores, boxes):
        r = orig(scores, boxes)
        picked.append(r)
        return r
    TVG.best_detection = rec
    try:
        acc, num_sent = TVG.eval_split(Loader(), net, None, 'val', dict(verbose=False))
    finally:
        TVG.best_detection = orig
    assert num_sent == int(g['num_sent']) and acc == float(g['acc'])
    assert [p[1] for p in picked] == list(g['pred_class'])
    assert np.allclose(np.stack([p[2] for p in picked]), g['pred_box'], atol=1e-2)


def test_eval_split_vs_reference():
    """model/test.py eval_split against the reference's own evaluation loop (model/test.py:185-450, run through the harness on the same
    tiny synthetic split by tests/golden/make_golden.py eval_split -> ref_eval_split.npz): chosen (RoI, class), predicted boxes, box
    accuracy, precision@X counts and the cumulative intersection / union pixel counts of the recovered masks."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_golden as MG
    from lang2seg_amd import selftest
    from lang2seg_amd.model import test as T
    from oracle import weights as OW
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_eval_split.npz')))
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    net = selftest.build_net(opt, {}, 'f32', MG.eval_state_dict(opt))
    imgs = MG.eval_blobs()

    class Loader(object):
        split_ix = {'val': [0, 1]}

        def __init__(self):
            self.i = 0

        def getTestBatch(self, split):
            b = dict(imgs[self.i]); self.i += 1
            b['bounds'] = dict(it_pos_now=self.i, it_max=len(imgs), wrapped=self.i >= len(imgs))
            return b
    picked = []
    orig = T.best_detection

    def rec(scores, boxes):
        r = orig(scores, boxes)
        picked.append(r)
        return r
    T.best_detection = rec
    try:
        acc, thr, seg_correct, seg_total, cum_I, cum_U, num_sent = T.eval_split(Loader(), net, None, 'val', dict(verbose=False))
    finally:
        T.best_detection = orig
    assert num_sent == int(g['num_sent']) == seg_total == int(g['seg_total']) and list(thr) == list(g['thr'])
    assert [p[1] for p in picked] == list(g['pred_class'])
    assert np.allclose(np.stack([p[2] for p in picked]), g['pred_box'], atol=1e-2)
    assert acc == float(g['acc']) and list(seg_correct) == list(g['seg_correct'])
    # masks are thresholded at 122/255 after a bilinear resize of fp32 probabilities: a last-bit difference may move a handful of pixels
    assert abs(int(cum_I) - int(g['cum_I'])) <= 3 and abs(int(cum_U) - int(g['cum_U'])) <= 3, (cum_I, cum_U, int(g['cum_I']), int(g['cum_U']))
    assert int(g['cum_I']) > 0


def test_eval_split_vgg_runs():
    """model/test_vgg.py (boxes only) on the VGG16 / Faster R-CNN network in TEST mode"""
    from lang2seg_amd import selftest
    from lang2seg_amd.model.test_vgg import eval_split
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    from oracle import weights as OW
    opt = OW.default_opt(vocab_size=60, seq_length=6); opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant='vgg')
    net = selftest.build_net(opt, {}, 'bf16', sd, variant='vgg')
    loader = SyntheticLoader(num_images=2, sents_per_image=2, H=320, W=416, T=6, vocab_size=60)
    acc, n = eval_split(loader, net, None, 'val', dict(verbose=False))
    assert n == 4 and 0.0 <= acc <= 1.0


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_dp_tape_segments_match_eager():
    """data-parallel replay: the launch tape cut at the gradient-bucket hand-offs (l2s_tape_mark / l2s_tape_run_segment) with a
    one-rank RCCL reducer must train exactly like the eager data-parallel step (same device RNG counter, same inputs)."""
    import os, tempfile
    import torch.distributed as dist
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from lang2seg_amd.parallel import GradReducer
    from oracle import weights as OW, synth as OS
    if not dist.is_initialized():
        f = tempfile.NamedTemporaryFile(delete=False); f.close()
        dist.init_process_group('nccl', init_method='file://' + f.name, rank=0, world_size=1, device_id=torch.device('cuda', 0))
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    blob = OS.make_blob(320, 416, 6, 60, seed=5)
    over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
    res = []
    for tape in (False, True):
        net = selftest.build_net(opt, over, 'f32', sd)
        net.dp = GradReducer(net, 1)
        net.use_tape = tape
        sgd = SGD(net, 0.0, keep_grad=True)   # lr 0: the weights stay put, so the steps are comparable one by one (the proposal list is
                                       # discontinuous in the weights; any update would let fp32 atomic-order noise pick different RoIs)
        # (the first tape call executes the step once and records it while it runs; the later calls replay)
        vals = [net.train_step(dict(blob), 0, sgd) for _ in range(4)]
        torch.cuda.synchronize()
        res.append((vals, net.P.view('resnet.layer3.5.conv2.weight', net.P.grad).clone(), net.P.view('cls_score_net.weight', net.P.grad).clone()))
    for a, b in zip(res[0][0], res[1][0]):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6), (a, b)
    # all-reduced gradients of the last step (fp32 atomics: order noise only)
    assert rel(res[0][1], res[1][1]) < 1e-3 and rel(res[0][2], res[1][2]) < 1e-3


def test_dp_sharded_update_matches_plain_update():
    """parallel.GradReducer(algo='rs_ag', shard_update=...) on ONE rank of a real RCCL process group: reduce-scatter, the update of this rank's
    slice of every bucket on the reducer's stream (l2s_sgd_momentum_range), all-gather of the weights, shadow rewrite - the weights and the
    momentum after four steps are those of the plain single-launch update, bit for bit (world 1: the slice is the bucket), eager and from
    the segmented tape.  (Two ranks: tests/test_host_cpu.py::test_grad_reducer_gloo_world2.)"""
    import tempfile
    import torch.distributed as dist
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from lang2seg_amd.parallel import GradReducer
    from oracle import weights as OW, synth as OS
    if not dist.is_initialized():
        f = tempfile.NamedTemporaryFile(delete=False); f.close()
        dist.init_process_group('nccl', init_method='file://' + f.name, rank=0, world_size=1, device_id=torch.device('cuda', 0))
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    blob = OS.make_blob(320, 416, 6, 60, seed=5)
    over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
    out = {}
    # 'bucket': the unsharded form - every bucket is updated whole right behind its all-reduce (GradReducer.bucket_update), gradients
    # cleared / overwrite-marked as in the single-process step
    for mode in ('plain', 'sharded', 'sharded+tape', 'bucket', 'bucket+tape', 'sh16', 'sh16-cast', 'sh16+tape'):
        net = selftest.build_net(opt, over, 'bf16', sd)
        net.use_tape = mode.endswith('tape')
        if mode.startswith('sh16'):
            # bf16 wire: the update reads the reduce-scattered bf16 shard directly (l2s_sgd_momentum_range_g16) or, '-cast', after a cast back to f32
            net.dp = GradReducer(net, 1, wire='bf16', algo='rs_ag', shard_update=True, rank=0)
            net.dp.shard_g16 = mode != 'sh16-cast'
        elif mode.startswith('sharded'):
            net.dp = GradReducer(net, 1, wire='fp32', algo='rs_ag', shard_update=True, rank=0)
        elif mode.startswith('bucket'):
            net.dp = GradReducer(net, 1, wire='fp32', algo='allreduce', bucket_update=True, rank=0)
        sgd = SGD(net, 1e-3, momentum=0.9, weight_decay=1e-4)
        assert (net.dp is None) or (net.dp.shard_update is sgd) != (net.dp.bucket_update is sgd)
        for i in range(4):
            net.train_step_async(dict(blob), 0, sgd)
            if i == 1:
                # a validation summary between two replayed steps (forward only, TV:389-396): it must not disturb the overwrite marks the next
                # update judges staleness by (Network._fresh used to be reset by forward-only passes, and the bucket update of the
                # next replayed step then cleared gradients that had already been reduced)
                net.get_summary(dict(blob), 0)
        torch.cuda.synchronize(); net.join_update(); torch.cuda.synchronize()
        if mode.startswith('bucket'):
            assert any(sg.flags & 1 for sg in net.P._seg_tables[0])                     # the overwrite marks are in force under the reducer too
            assert float(_grad_outside_overwritten(net.P).abs().max()) == 0.0
        out[mode] = (net.P.param.clone(), net.P.mom.clone(), net.P.shadow.clone())
    for mode in ('sharded', 'sharded+tape', 'bucket', 'bucket+tape'):
        for a, b, nm in zip(out[mode], out['plain'], ('param', 'momentum', 'shadow')):
            assert torch.equal(a, b), (mode, nm, int((a != b).sum()))
    for mode in ('sh16', 'sh16+tape'):                                  # same bits whether the shard is cast back first or read as it is
        for a, b, nm in zip(out[mode], out['sh16-cast'], ('param', 'momentum', 'shadow')):
            assert torch.equal(a, b), (mode, nm, int((a != b).sum()))
    assert rel(out['sh16'][0], out['plain'][0]) < 1e-3 and not torch.equal(out['sh16'][1], out['plain'][1])     # (the gradients really went through bf16)


def test_early_partial_sgd_matches_single_update():
    """optim.SGD.partial: the optimiser updates each finished prefix of the flat buffer during backward (on a side stream).  Every
    segment must be updated exactly once per step: momentum buffers after two lr=0 steps and the weight change of one lr=1e-3 step
    equal those of the single end-of-step launch (up to fp32 atomic-order noise in the gradients), eager and replayed from the tape."""
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    blob = OS.make_blob(320, 416, 6, 60, seed=5)
    over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)

    def run(early, tape, lr, calls):
        net = selftest.build_net(opt, over, 'f32', sd)
        net.use_tape = tape
        sgd = SGD(net, lr)
        sgd.early = early
        w0 = net.P.param.clone()
        for _ in range(calls):
            net.train_step(dict(blob), 0, sgd)
        torch.cuda.synchronize()
        assert sgd._seg_done == 0
        return net.P.mom.clone(), net.P.param - w0

    m_ref, _ = run(False, False, 0.0, 2)
    assert float(m_ref.abs().max()) > 0
    for early, tape, calls in ((True, False, 2), (True, True, 2)):       # first tape call: executed + recorded, second: replayed
        m, dw = run(early, tape, 0.0, calls)
        assert rel(m, m_ref) < 1e-3, (early, tape)
        assert float(dw.abs().max()) == 0.0
    _, d_ref = run(False, False, 1e-3, 1)
    _, d = run(True, False, 1e-3, 1)
    assert float(d_ref.abs().max()) > 0 and rel(d, d_ref) < 1e-3


def test_train_net_snapshot_and_resume(tmp_path):
    """model/train_val.py train_net (TV:327-434) end to end on the HIP step: display / LR step / snapshot cadence, the reference's
    snapshot pair (state dict in its key + shape format, sidecar with RNG states and loader cursors), and resume from the newest
    snapshot with the LR rescaled by the steps already passed (TV:227-310)."""
    import glob, os, pickle
    from lang2seg_amd import selftest
    from lang2seg_amd.model.config import cfg
    from lang2seg_amd.model.train_val import train_net
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    from oracle import weights as OW
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    over = dict(BATCH_SIZE=16, RPN_PRE_NMS_TOP_N=600, RPN_POST_NMS_TOP_N=100, RPN_BATCHSIZE=64)
    saved = {k: cfg.TRAIN[k] for k in ('SNAPSHOT_ITERS', 'DISPLAY', 'STEPSIZE', 'SNAPSHOT_KEPT', 'LEARNING_RATE')}
    cfg.TRAIN.SNAPSHOT_ITERS, cfg.TRAIN.DISPLAY, cfg.TRAIN.STEPSIZE, cfg.TRAIN.SNAPSHOT_KEPT = 2, 1, [3], 3
    out = str(tmp_path / 'out')
    try:
        mk_loader = lambda: SyntheticLoader(num_images=3, sents_per_image=2, H=160, W=224, T=6, vocab_size=60)
        net = selftest.build_net(opt, over, 'f32', sd)
        sw = train_net(net, mk_loader(), out, str(tmp_path / 'tb'), max_iters=4)
        pths = sorted(os.path.basename(f) for f in glob.glob(os.path.join(out, '*.pth')))
        assert cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_4.pth' in pths and cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_2.pth' in pths
        assert abs(sw.optimizer.lr - cfg.TRAIN.LEARNING_RATE * cfg.TRAIN.GAMMA) < 1e-12          # stepped at iter 4 = STEPSIZE + 1
        ck = torch.load(os.path.join(out, cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_4.pth'), map_location='cpu')
        # (the reference's state dict also carries torchvision's unused resnet.fc.*; the oracle's weight set does not)
        assert set(sd.keys()) <= set(ck.keys()) and all(tuple(ck[k].shape) == sd[k].shape for k in sd)
        assert any(float((ck[k].float() - torch.from_numpy(sd[k])).abs().max()) > 0 for k in ('cls_score_net.weight', 'resnet.layer3.5.conv2.weight'))
        w4 = {k: v.clone() for k, v in net.state_dict().items()}
        for k in ck:
            assert torch.equal(ck[k], w4[k]), k
        with open(os.path.join(out, cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_4.pkl'), 'rb') as f:
            pickle.load(f); pickle.load(f)
            it_train = pickle.load(f)
        # resume: a fresh network and loader pick the newest snapshot up and continue to iteration 6
        net2 = selftest.build_net(opt, over, 'f32', sd)
        ld2 = mk_loader()
        sw2 = train_net(net2, ld2, out, str(tmp_path / 'tb'), max_iters=6)
        assert os.path.exists(os.path.join(out, cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_6.pth'))
        assert abs(sw2.optimizer.lr - cfg.TRAIN.LEARNING_RATE * cfg.TRAIN.GAMMA) < 1e-12          # rescaled on resume: 4 > STEPSIZE
        assert isinstance(it_train, int) and int(net2.seed_counter().item()) > 0
        d = max(float((net2.state_dict()[k].float() - w4[k].float()).abs().max()) for k in ('cls_score_net.weight', 'rpn_net.weight'))
        assert 0 < d < 1.0                                  # moved on from the restored weights
    finally:
        for k, v in saved.items():
            cfg.TRAIN[k] = v


def test_resume_from_reference_written_snapshot(tmp_path):
    """f3 pinned against files the REFERENCE wrote: the snapshot pair of train_val_cycle.py:57-104 (tests/golden/ref_snapshot/,
    produced by the reference's own SolverWrapper.snapshot() in tests/golden/make_golden.py; payload records regenerated and CRC-checked) is
    restored by the build's from_snapshot into a freshly initialised network on the device - every tensor equal to what the reference saved,
    cursors / RNG streams / iteration as the reference's from_snapshot restores them - and the step that follows matches the reference's own
    step from those weights (the `tiny` fixture: same synthetic weights, same image) at the f32 tolerance.  Then the reverse direction's
    input: the pair the BUILD writes keeps the reference's key order, shapes and dtypes; its structure goes to gpurun_out/build_snapshot/ for
    tests/golden/make_golden.py `read_build_snapshot`, which has the reference's from_snapshot load it."""
    import json, os, pickle, random, zipfile, zlib
    from golden_util import materialize_ref_snapshot
    from lang2seg_amd import selftest
    from lang2seg_amd.model.train_val import SolverWrapper
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    from lang2seg_amd.nets.variants import loss_names, SLOT
    sfile, nfile, man = materialize_ref_snapshot(str(tmp_path / 'snap'))
    g = load('tiny')
    opt, sd, blob, ocfg, samp = setup_from_fixture(g)
    samp['forced_proposals'] = (g['int.proposal_rois'], g['int.proposal_scores'])
    over = {k[4:]: int(g[k]) for k in g if k.startswith('cfg.')}
    net = selftest.build_net(opt, over, 'f32', None)                        # its own initialisers: everything must come from the snapshot
    ld = SyntheticLoader(num_images=3, sents_per_image=2, H=160, W=224, T=6, vocab_size=60)
    ld.split_ix = {'train': list(range(11)), 'val': list(range(5))}         # (the reference-side stub loader of the fixture: 11 / 5 images)
    sw = SolverWrapper(net, ld, str(tmp_path / 'out'), str(tmp_path / 'tb'))
    sw.construct_graph()
    before = net.state_dict()
    assert any(float((before[k].float() - torch.from_numpy(sd[k])).abs().max()) > 0 for k in ('rpn_net.weight', 'resnet.layer3.5.conv2.weight'))
    np.random.seed(1); random.seed(1)
    last = sw.from_snapshot(sfile, nfile)
    want = man['restore_full']
    assert last == want['last_snapshot_iter'] and ld.iterators['train'] == want['iter_train'] and ld.iterators['val'] == want['iter_val']
    assert [int(x) for x in ld.perm['train']] == want['perm_train'] and [int(x) for x in ld.perm['val']] == want['perm_val']
    assert [float(x) for x in np.random.rand(3)] == want['next_np_rand'] and random.random() == want['next_py_random']
    saved = torch.load(sfile, map_location='cpu')
    got = net.state_dict()
    for k, v in saved.items():
        if k.endswith('num_batches_tracked'):
            continue                                                        # BatchNorm bookkeeping the frozen-BN network has no use for
        assert k in got and torch.equal(got[k], v), k
    # ---- the pair the build writes (before the step: its payloads are then the restored weights, which the container-side read-back can
    # regenerate): the reference's format - key order, shapes, dtypes; sidecar fields in its order ----
    np.random.seed(4321); np.random.rand(2); random.seed(55)
    ld.iterators['train'], ld.iterators['val'] = 3, 1
    bs, bn = sw.snapshot(9)
    ck = torch.load(bs, map_location='cpu')
    ref_keys = [e['key'] for e in man['keys'] if not e['key'].endswith('num_batches_tracked')]
    assert list(ck.keys()) == ref_keys
    for e in man['keys']:
        if e['key'] in ck:
            assert list(ck[e['key']].shape) == e['shape'] and str(ck[e['key']].dtype) == e['dtype'], e['key']
    with open(bn, 'rb') as f:
        st0, st1, it_tr, perm_tr, it_val, perm_val, it9 = [pickle.load(f) for _ in range(7)]
    assert st0[0] == 'MT19937' and it_tr == 3 and it_val == 1 and it9 == 9 and len(perm_tr) == 11 and len(perm_val) == 5
    try:                                                                    # structure + payload CRCs for the container-side reference read-back
        outd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'build_snapshot')
        os.makedirs(outd, exist_ok=True)
        recs = []
        with zipfile.ZipFile(bs) as z:
            for i in z.infolist():
                recs.append(dict(name=i.filename, size=i.file_size, crc32=i.CRC))
                if '/data/' not in i.filename:
                    with open(os.path.join(outd, 'pth.' + i.filename.split('/', 1)[1].replace('/', '.')), 'wb') as f:
                        f.write(z.read(i.filename))
        keys = []
        for i, (k, t) in enumerate(ck.items()):
            raw = t.contiguous().numpy().tobytes()
            keys.append(dict(key=k, shape=list(t.shape), dtype=str(t.dtype), size=len(raw), crc32=zlib.crc32(raw) & 0xFFFFFFFF,
                             sum=float(t.double().sum()), abssum=float(t.double().abs().sum())))
        with open(bn, 'rb') as f, open(os.path.join(outd, os.path.basename(bn)), 'wb') as o:
            o.write(f.read())
        json.dump(dict(pth=os.path.basename(bs), pkl=os.path.basename(bn), iter=9, records=recs, keys=keys, torch=torch.__version__,
                       expect=dict(iter_train=3, iter_val=1, perm_train=[int(x) for x in perm_tr], perm_val=[int(x) for x in perm_val]),
                       written_by='lang2seg_amd/model/train_val.py SolverWrapper.snapshot on the MI355X, right after restoring the reference-written snapshot'),
                  open(os.path.join(outd, 'manifest.json'), 'w'), indent=1)
    except OSError:
        pass
    # the step after the restore = the reference's step from these weights
    net.parity = selftest.parity_from_samp(samp)
    loss = net.forward_backward(net.upload_blob(blob, 0))
    torch.cuda.synchronize()
    lv = loss.cpu().numpy()
    for k in loss_names('cycle'):
        assert abs(lv[SLOT[k]] - float(g['loss.' + k])) < 1e-4 * max(1.0, abs(float(g['loss.' + k]))), (k, lv[SLOT[k]], g['loss.' + k])
    names = _check_grads(g, net, 'f32', 5e-4)
    sw.optimizer.step()
    torch.cuda.synchronize()
    sd1 = net.state_dict()
    for nme in names:
        check_digest(g, 'w1.' + nme, sd1[nme].numpy(), rtol=1e-5, atol=1e-7)


def _edge_blob(kind):
    from oracle import synth as OS
    V = 60
    if kind == 'one_token':                     # shortest expression: a single token (max_len = 1), caption of one word
        b = OS.make_blob(160, 224, 1, V, seed=21)
    elif kind == 'padded_tokens':               # zero padding inside the label row: T = 6 columns, 3 real tokens
        b = OS.make_blob(160, 224, 6, V, seed=22)
        b['labels'][0, 3:] = 0
        b['cap_labels'][0, 4:] = 0
        b['cap_masks'][0, 5:] = 0               # 3 words + BOS + EOS positions count (cycle_loader.py:297-305)
    elif kind == 'small_image':                 # 96x128 image: 6x8 map, 576 anchors < RPN_PRE_NMS_TOP_N, few proposals
        b = OS.make_blob(96, 128, 4, V, seed=23)
    elif kind == 'full_image_box':              # the referred object fills the image
        b = OS.make_blob(160, 224, 5, V, seed=24)
        H, W = 160, 224
        b['gt_boxes'][0, :4] = [0, 0, W - 1, H - 1]
        b['gt_masks'][:] = 1
    elif kind == 'tiny_box':                    # an object smaller than one feature-map cell
        b = OS.make_blob(160, 224, 5, V, seed=25)
        b['gt_boxes'][0, :4] = [100, 60, 109, 70]
        b['gt_masks'][:] = 0
        b['gt_masks'][0, 60:71, 100:110] = 1
    return b


@pytest.mark.parametrize('kind', ['one_token', 'padded_tokens', 'small_image', 'full_image_box', 'tiny_box'])
def test_edge_cases_vs_oracle(kind):
    """ragged / extreme inputs of the blobs contract through the whole HIP step against the oracle (same recorded sampling keys, the
    device's own proposals teacher-forced into the oracle): losses within 1e-3, integer targets exact."""
    import copy
    from lang2seg_amd import selftest
    from oracle import weights as OW, net as ON
    blob = _edge_blob(kind)
    H, W = blob['data'].shape[1:3]
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    over = dict(BATCH_SIZE=16, RPN_PRE_NMS_TOP_N=600, RPN_POST_NMS_TOP_N=100, RPN_BATCHSIZE=64)
    ocfg = copy.deepcopy(ON.DEFAULT_CFG); ocfg['TRAIN'].update(over)
    Hc, Wc = -(-H // 16), -(-W // 16)
    nA = Hc * Wc * 12
    rs = np.random.RandomState(1)
    samp = dict(rpn_fg_keys=rs.permutation(nA).astype(np.uint32), rpn_bg_keys=rs.permutation(nA).astype(np.uint32),
                roi_fg_keys=rs.permutation(100).astype(np.uint32), roi_bg_keys=rs.permutation(100).astype(np.uint32))
    net = selftest.build_net(opt, over, 'f32', sd)
    net.parity = selftest.parity_from_samp(samp)
    lv = net.forward_backward(net.upload_blob(blob, 0)).cpu().numpy()
    n = int(net.t['proposal_n'].item())
    assert n > 0
    samp['forced_proposals'] = (net.t['proposal_rois'].cpu().numpy()[:n], net.t['proposal_scores'].cpu().numpy()[:n])
    onet = ON.OracleNet(sd, opt, ocfg)
    T, L = onet.forward_train(blob, samp)
    for i, k in enumerate(NAMES):
        ref = float(L[k])
        assert np.isfinite(lv[i]) and abs(lv[i] - ref) < 1e-3 * max(1.0, abs(ref)), (kind, k, lv[i], ref)
    assert np.array_equal(net.t['labels'].cpu().numpy().astype(np.int64), np.asarray(T['labels']).reshape(-1).astype(np.int64))
    assert np.array_equal(net.t['rpn_labels'].cpu().numpy().astype(np.int64), np.asarray(T['rpn_labels']).reshape(-1).astype(np.int64))
    assert np.isfinite(net.P.grad.float().abs().sum().item())


def test_only_foreground_rois_through_the_step():
    """proposal_target_layer.py:155-158: no background candidate at all -> all R sampled RoIs are foreground and the mask branch runs on
    every one of them (NET:584-586).  TRAIN.MASK_SLOTS_ALL sizes the mask head for that; the whole step (losses, integer targets)
    against the oracle on proposals that all hug the gt box."""
    import copy
    from lang2seg_amd import selftest
    from lang2seg_amd.model.config import cfg
    from oracle import weights as OW, net as ON, synth as OS
    blob = OS.make_blob(160, 224, 5, 60, seed=31)
    H, W = 160, 224
    blob['gt_boxes'][0, :4] = [40, 30, 180, 130]
    blob['gt_masks'][:] = 0; blob['gt_masks'][0, 30:131, 40:181] = 1
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    over = dict(BATCH_SIZE=16, RPN_PRE_NMS_TOP_N=600, RPN_POST_NMS_TOP_N=100, RPN_BATCHSIZE=64)
    ocfg = copy.deepcopy(ON.DEFAULT_CFG); ocfg['TRAIN'].update(over)
    rs = np.random.RandomState(2)
    nA = 10 * 14 * 12
    props = np.zeros((40, 5), np.float32)
    props[:, 1:] = np.clip(blob['gt_boxes'][0, :4] + rs.normal(0, 4, (40, 4)), 0, [W - 1, H - 1, W - 1, H - 1])
    samp = dict(rpn_fg_keys=rs.permutation(nA).astype(np.uint32), rpn_bg_keys=rs.permutation(nA).astype(np.uint32),
                roi_fg_keys=rs.permutation(100).astype(np.uint32), roi_bg_keys=rs.permutation(100).astype(np.uint32),
                forced_proposals=(props, np.linspace(0.9, 0.5, 40).astype(np.float32)))
    cfg.TRAIN.MASK_SLOTS_ALL = True
    try:
        net = selftest.build_net(opt, over, 'f32', sd)
        net.parity = selftest.parity_from_samp(samp)
        lv = net.forward_backward(net.upload_blob(blob, 0)).cpu().numpy()
        torch.cuda.synchronize()
    finally:
        cfg.TRAIN.MASK_SLOTS_ALL = False
    assert int(net.t['counts'][0].item()) == 16 and int(net.t['counts'][2].item()) == 0          # 16 foreground RoIs, no background candidate
    onet = ON.OracleNet(sd, opt, ocfg)
    T, L = onet.forward_train(blob, samp)
    for i, k in enumerate(NAMES):
        ref = float(L[k])
        assert np.isfinite(lv[i]) and abs(lv[i] - ref) < 1e-3 * max(1.0, abs(ref)), (k, lv[i], ref)
    assert np.array_equal(net.t['labels'].cpu().numpy().astype(np.int64), np.asarray(T['labels']).reshape(-1).astype(np.int64))
    assert (net.t['labels'].cpu().numpy() > 0).all()
    assert np.array_equal(net.t['mask_targets'].cpu().numpy().reshape(16, 14, 14), np.asarray(T['mask_targets']).reshape(16, 14, 14))


def test_tape_over_mixed_shapes_matches_eager():
    """real data feeds a different image size / token count almost every step: the launch tape records a new shape while it executes
    it (no extra step), replays the ones it has seen, and evicts the least recently used tape together with its activation plan.
    Same losses as the eager step for the same sequence of inputs (lr 0: weights stay put)."""
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    over = dict(BATCH_SIZE=16, RPN_PRE_NMS_TOP_N=600, RPN_POST_NMS_TOP_N=100, RPN_BATCHSIZE=64)
    blobs = [OS.make_blob(160, 224, 6, 60, seed=5), OS.make_blob(128, 256, 4, 60, seed=6), OS.make_blob(192, 160, 5, 60, seed=7)]
    order = [0, 1, 0, 2, 1, 0, 2, 2, 1]                   # with max_tapes = 2 the third shape evicts the least recently used one
    res = []
    for tape in (False, True):
        net = selftest.build_net(opt, over, 'f32', sd)
        net.use_tape = tape
        net.max_tapes = 2
        sgd = SGD(net, 0.0)
        res.append([net.train_step(dict(blobs[i]), 0, sgd) for i in order])
        torch.cuda.synchronize()
        if tape:
            assert len(net._tapes) == 2
            live = {k[0] for k in net._tapes}
            assert all(any(t[0] in live for t in users) for users in net._buf_users.values())
    for k, (a, b) in enumerate(zip(res[0], res[1])):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6), (k, order[k], a, b)


def test_pipelined_tape_training_matches_eager_training():
    """weights after ten PIPELINED steps with a real learning rate (no host sync between steps, so the next step's launches are queued while
    the optimiser update of the previous one runs on the weight-gradient stream): replayed from the launch tape vs issued eagerly, f32, same
    data.  Catches a missing cross-stream edge of the update (round 2: a tape recorded before the update moved to the weight-gradient stream
    cleared the gradients on the main queue while the update was still reading them -- nothing learned, every test with lr 0 passed)."""
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    over = dict(BATCH_SIZE=16, RPN_PRE_NMS_TOP_N=600, RPN_POST_NMS_TOP_N=100, RPN_BATCHSIZE=64)
    blob = OS.make_blob(160, 224, 6, 60, seed=5)
    params = []
    for tape in (False, True):
        net = selftest.build_net(opt, over, 'f32', sd)            # (sampling keys and dropout masks: device RNG keyed by the step counter,
        net.use_tape = tape                                      #  the same sequence in both runs)
        sgd = SGD(net, 1e-3, momentum=0.9, weight_decay=1e-4)
        p0 = net.P.param.clone()
        for _ in range(10):
            net.train_step_async(dict(blob), 0, sgd)
        torch.cuda.synchronize()
        net.join_update()
        torch.cuda.synchronize()
        params.append(net.P.param.clone())
        assert bool(torch.isfinite(params[-1]).all()) and float((params[-1] - p0).abs().max()) > 1e-5       # it did learn something
    a, b = params
    assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max())), float((a - b).abs().max())


def _grad_outside_overwritten(P):
    """P.grad with the segments zeroed whose gradients the grouped weight gradients overwrite every step (l2s_sgd_seg.flags = 1: the update
    does not clear those, ParamStore.mark_overwritten)"""
    g = P.grad.clone()
    for sg in P._seg_tables[0]:
        if sg.flags & 1:
            g[int(sg.offset):int(sg.offset + sg.count)] = 0
    return g


def test_wgrad_overwrite_bit_identical():
    """Network.wgrad_overwrite (a tensor's first grouped weight-gradient problem of the step writes dW instead of adding to it, and the update
    leaves those gradients uncleared): K pipelined steps give the same weights and momentum, bit for bit, as K steps that add into cleared
    gradients; the convolution weights are the overwritten ones, everything else is still cleared by the update.
    Semantics matched: optimizer.zero_grad() + backward() + step() (train_val_cycle.py:383-386)."""
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from lang2seg_amd.nets.network import Network
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
    blobs = [OS.make_blob(320, 416, 6, 60, seed=5), OS.make_blob(320, 416, 6, 60, seed=6)]
    out = {}
    for ow, tape, early in ((False, False, False), (False, True, False), (True, False, False), (True, True, False), (True, False, True), (True, True, True)):
        if True:
            net = selftest.build_net(opt, over, 'bf16', sd)
            net.wgrad_overwrite = ow
            net.use_tape = tape
            sgd = SGD(net, 1e-3, momentum=0.9, weight_decay=1e-4)
            sgd.early = early                   # partial updates during backward consume the marks of the previous complete step
            for i in range(6):
                net.train_step_async(dict(blobs[i % 2]), 0, sgd)
            torch.cuda.synchronize()
            net.join_update()
            torch.cuda.synchronize()
            flagged = [sg for sg in net.P._seg_tables[0] if sg.flags & 1]
            assert bool(flagged) == ow
            if ow:
                names = {k for k in net.P.trainable if any(net.P.offsets[k] == int(sg.offset) for sg in flagged)}
                assert names and all(k.endswith('.weight') for k in names) and any('layer4' in k for k in names) and any('layer2' in k for k in names)
                assert any(float(net.P.grad[int(sg.offset):int(sg.offset + sg.count)].abs().max()) > 0 for sg in flagged)
            assert float(_grad_outside_overwritten(net.P).abs().max()) == 0.0
            out[(ow, tape, early)] = (net.P.param.clone(), net.P.mom.clone())
    p0, m0 = out[(False, False, False)]
    assert bool(torch.isfinite(p0).all()) and float(m0.abs().max()) > 0
    for key, (p, m) in out.items():
        assert torch.equal(p, p0), (key, int((p != p0).sum()))
        assert torch.equal(m, m0), (key, int((m != m0).sum()))


@pytest.mark.parametrize('variant', ['cycle', 'vgg'])
def test_deferred_heads_bit_identical(variant):
    """optim.SGD.defer (the heads stage's grouped weight-gradient launches and their part of the update run BEHIND the rest of the update,
    beside the next step's backbone forward; layer2 waits for an event slot, the dynamic filters for the whole stream): K pipelined steps
    with a real learning rate give the same weights and momentum, BIT FOR BIT, as K steps with the undeferred order - eager and replayed
    from the launch tape, bf16, and with a state_dict() read in the middle of the run (a reader outside the step joins the deferred tail).
    Semantics matched: one optimizer.step() per train_step over all parameters (train_val_cycle.py:194-220, network_cycle_res5_2.py:712-715)."""
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    if variant == 'vgg':
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant) if variant == 'vgg' else OW.make_state_dict(opt, seed=3, head_gain=4.0)
    over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
    blobs = [OS.make_blob(320, 416, 6, 60, seed=5), OS.make_blob(320, 416, 6, 60, seed=6)]
    out = {}
    was = SGD.defer
    try:
        for defer in (False, True):
            for tape in (False, True):
                SGD.defer = defer
                net = selftest.build_net(opt, over, 'bf16', sd, variant=variant)
                net.use_tape = tape
                sgd = SGD(net, 1e-3, momentum=0.9, weight_decay=1e-4)
                assert sgd.defer_active == defer and net.defer_heads == defer
                mid = None
                for i in range(8):
                    net.train_step_async(dict(blobs[i % 2]), 0, sgd)
                    if i == 4:
                        mid = {k: v.clone() for k, v in net.state_dict().items()}          # no sync before it: the tail is still in flight
                torch.cuda.synchronize()
                net.join_update()
                torch.cuda.synchronize()
                out[(defer, tape)] = (net.P.param.clone(), net.P.mom.clone(), mid, _grad_outside_overwritten(net.P))
    finally:
        SGD.defer = was
    p0, m0, mid0, g0 = out[(False, False)]
    assert bool(torch.isfinite(p0).all()) and float(m0.abs().max()) > 0
    assert float(g0.abs().max()) == 0.0                       # the update consumed and cleared every gradient
    for key, (p, m, mid, g) in out.items():
        assert torch.equal(p, p0), (variant, key, int((p != p0).sum()))
        assert torch.equal(m, m0), (variant, key, int((m != m0).sum()))
        assert torch.equal(g, g0), (variant, key)
        for k in mid0:
            assert torch.equal(mid[k], mid0[k]), (variant, key, k)


def test_encoder_is_deterministic_beside_other_kernels():
    """round 4: the encoder's LSTM on the language stream, replayed from a launch tape beside the first layers of the VGG
    backbone on the main stream, must give the same hidden state every time.  With packed fp32 VALU ops in the LSTM step kernel
    (v_pk_fma_f32, the compiler's pairing of its four fmaf chains) 573 ... 2850 of 3000 such replays differed by up to 1e-2; the library
    is built without them (__graft_entry__.HIPCC_FLAGS; tests/test_host_cpu.py scans the ISA)."""
    from lang2seg_amd import selftest, ops as O
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6); opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant='vgg')
    net = selftest.build_net(opt, dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64), 'bf16', sd, variant='vgg')
    dev = net.upload_blob(dict(OS.make_blob(320, 416, 6, 60, seed=5)), 0)
    main = torch.cuda.current_stream()
    S = net.streams()
    slist = [main, S['lang'], S['cap'], S['wg'], S['wg2'], S['tr']]
    full_plan = list(net.vgg_plan)
    for nlayers in (3, 4):
        net.vgg_plan = full_plan[:nlayers]
        net.t = {}
        torch.cuda.synchronize()
        h = O.tape_begin(slist)
        net._rec_key = ('fuzz', nlayers)
        try:
            net.sfork(main, S['lang'])
            with torch.cuda.stream(S['lang']):
                hidden = net._encoder_fwd(dev)
            net._backbone_fwd(dev, {})
            net.sfork(S['lang'], main)
        finally:
            O.tape_end(h); net._rec_key = None
        torch.cuda.synchronize()
        ref = hidden.clone()
        bad = 0
        for _ in range(1500):
            O.tape_run(h, slist)
            torch.cuda.synchronize()
            bad += int(not torch.equal(hidden, ref))
        O.tape_destroy(h)
        assert bad == 0, (nlayers, bad)


def test_tape_stops_recording_when_shapes_keep_changing():
    """a stream of inputs whose (image size, token count) key is new almost every step: once fewer than half of the last `tape_window`
    steps were replays, a miss runs eagerly and is not recorded (no tape, no pinned activation plan per key); known keys keep replaying"""
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    over = dict(BATCH_SIZE=16, RPN_PRE_NMS_TOP_N=600, RPN_POST_NMS_TOP_N=100, RPN_BATCHSIZE=64)
    net = selftest.build_net(opt, over, 'f32', sd)
    net.use_tape = True
    net.tape_window = 4
    sgd = SGD(net, 0.0)
    shapes = [(160, 224, 6), (128, 256, 4), (192, 160, 5), (160, 192, 3), (176, 208, 6), (144, 240, 2), (128, 224, 5)]
    for i, (h, w, tk) in enumerate(shapes):
        vals = net.train_step(dict(OS.make_blob(h, w, tk, 60, seed=40 + i)), 0, sgd)
        assert all(np.isfinite(v) for v in vals)
    torch.cuda.synchronize()
    assert len(net._tapes) == 3                               # the fourth step finds the window full of misses: it and the rest ran eagerly
    first = dict(OS.make_blob(160, 224, 6, 60, seed=40))
    a = net.train_step(first, 0, sgd)                         # a key that is on tape still replays
    assert len(net._tapes) == 3 and all(np.isfinite(v) for v in a)


def test_fused_roialign_step_is_bit_identical():
    """cfg.TRAIN.FUSE_ROIALIGN (crop-and-resize + layer4[0].conv1 + layer4[0].downsample in one launch): the bf16 train step with it gives the
    same crop and the same gradient buffer, bit for bit, as the three launches it replaces (same crop arithmetic, same K order)."""
    from lang2seg_amd import selftest
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    blob = OS.make_blob(320, 416, 6, 60, seed=5)
    over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
    out = []
    for fuse in (False, True):
        net = selftest.build_net(opt, over, 'bf16', sd)
        net.fuse_roialign = fuse
        lv = net.forward_backward(net.upload_blob(dict(blob), 0)).cpu().numpy()
        torch.cuda.synchronize()
        out.append((lv.copy(), net.P.grad.clone(), net.t['pool5'].clone()))
    assert np.allclose(out[0][0], out[1][0], rtol=1e-5, atol=1e-6), (out[0][0], out[1][0])     # (the reported loss scalars are summed with atomics)
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize('variant', ['cycle', 'vgg'])
def test_no_gradient_lands_behind_its_bucket(variant):
    """Data parallel: a gradient bucket may only be handed to the reducer when every weight gradient of its stage has been launched
    (ConvOp.wgrad only QUEUES; round 2's VGG backbone handed its bucket over with the queue unflushed, so each rank's local dW was added
    on top of the all-reduced sum).  A stand-in reducer doubles every bucket it is handed - what a two-rank sum of equal gradients does -
    so the finished gradient buffer must be exactly twice the single-process one in EVERY tensor."""
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from lang2seg_amd.parallel import GradReducer
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    if variant == 'vgg':
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant) if variant == 'vgg' else OW.make_state_dict(opt, seed=3, head_gain=4.0)
    over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
    blob = OS.make_blob(320, 416, 6, 60, seed=5)

    class Doubler(GradReducer):
        def _collective(self, buf):
            buf.mul_(2.0)

    grads = []
    for dp in (False, True):
        net = selftest.build_net(opt, over, 'f32', sd, variant=variant)
        net.use_tape = False
        if dp:
            net.dp = Doubler(net, 2)
        sgd = SGD(net, 0.0, grad_scale=0.5 if dp else 1.0, keep_grad=True)
        net.train_step(dict(blob), 0, sgd)
        torch.cuda.synchronize()
        grads.append(net.P.grad.clone())
    g1, g2 = grads
    assert float(g1.abs().max()) > 0
    bad = []
    net_P = net.P
    for k in net_P.trainable:
        o, n = net_P.offsets[k], int(np.prod(net_P.shapes[k]))
        a, b = g1[o:o + n], g2[o:o + n]
        if float(a.abs().max()) == 0:
            continue
        if rel(b, 2.0 * a) > 2e-3:                      # (fp32 summation-order noise only; a late gradient shows as a factor 1 or 1.5)
            bad.append((k, rel(b, 2.0 * a)))
    assert not bad, bad[:8]


@pytest.mark.parametrize('variant,rank', [('cycle', 0), ('cycle', 1), ('vgg', 1), ('baseline', 0)])
def test_sharded_update_stale_masters_are_never_read(variant, rank):
    """with the dtype shadow on the wire of the sharded data-parallel update, the fp32 masters of the OTHER ranks' slices are
    never updated on this rank - and the encoder, the captioner, every bias, the dynamic-filter FCs and the mask head read masters.  Two ranks in
    ONE process: net A runs as rank `rank` of a world of 2 whose other rank sees the same image (the collectives are stand-ins: a sum of two
    equal gradients, and all-gathers that deliver what the other rank would send - taken from net R, the same network stepping alone), and
    every master the reducer reports as behind (GradReducer.stale_master_ranges) is POISONED with NaN as soon as its shadow arrives.  The next
    steps must not see the poison, must train exactly like R, and gather_master() must bring every master back."""
    import types
    from lang2seg_amd import selftest, parallel
    from lang2seg_amd.optim import SGD
    from lang2seg_amd.parallel import GradReducer
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    if variant == 'vgg':
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant)
    over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
    blob = OS.make_blob(320, 416, 6, 60, seed=5)
    W, LR = 2, 2e-3
    R = selftest.build_net(opt, over, 'bf16', sd, variant=variant)
    A = selftest.build_net(opt, over, 'bf16', sd, variant=variant)
    for n_ in (R, A):
        n_.use_tape = False
    gathered = {'shadow': 0, 'param': 0}

    def locate(t):
        for name in ('shadow', 'param'):
            buf = getattr(A.P, name)
            off = (t.data_ptr() - buf.data_ptr()) // buf.element_size()
            if t.dtype == buf.dtype and 0 <= off < buf.numel() and (t.data_ptr() - buf.data_ptr()) % buf.element_size() == 0:
                return name, int(off)
        raise AssertionError('all-gather into a buffer that is neither the shadow nor the masters')

    class FakeDist(object):
        ReduceOp = types.SimpleNamespace(SUM=0)
        is_available = staticmethod(lambda: False)          # (GradReducer._staged: not a gloo group - the stand-ins take device buffers)

        @staticmethod
        def reduce_scatter_tensor(out, inp, op=None):
            per = out.numel()
            out.copy_(inp[rank * per:(rank + 1) * per]); out.mul_(W)          # the other rank contributes the same gradients

        @staticmethod
        def all_reduce(t, op=None):
            t.mul_(W)

        @staticmethod
        def all_gather_into_tensor(out, inp):
            name, off = locate(out)
            per = inp.numel()
            truth = getattr(R.P, name)
            for j in range(W):
                if j == rank:
                    if out[j * per:(j + 1) * per].data_ptr() != inp.data_ptr():
                        out[j * per:(j + 1) * per].copy_(inp)
                    continue
                out[j * per:(j + 1) * per].copy_(truth[off + j * per:off + (j + 1) * per])     # what rank j updated and sent
                if name == 'shadow' and poison[0]:
                    A.P.param[off + j * per:off + (j + 1) * per].fill_(float('nan'))           # its master never arrives here
            gathered[name] += out.numel()
    poison = [True]
    saved = parallel.dist
    parallel.dist = FakeDist
    try:
        A.dp = GradReducer(A, W, wire='fp32', algo='rs_ag', shard_update=True, rank=rank)
        sgd_r = SGD(R, LR)
        sgd_a = SGD(A, LR, grad_scale=1.0 / W)
        assert A.dp.shard_update is sgd_a and A.dp.gather_shadow
        lr_, la_ = [], []
        for step in range(3):
            lr_.append(R.train_step(dict(blob), 0, sgd_r)); torch.cuda.synchronize()
            la_.append(A.train_step(dict(blob), 0, sgd_a)); torch.cuda.synchronize()
        assert gathered['shadow'] > 0 and gathered['param'] > 0, gathered
        assert np.isfinite(np.array(la_)).all(), la_                       # no kernel read a poisoned master
        assert abs(lr_[2][-1] - lr_[0][-1]) > 1e-3 * abs(lr_[0][-1])       # (the steps really trained)
        for a_, r_ in zip(la_, lr_):
            assert np.allclose(a_, r_, rtol=2e-3, atol=2e-4), (la_, lr_)   # same losses as the network that stepped alone
        # what the reducer says is behind is exactly what was poisoned, and only where no kernel reads masters
        stale = torch.zeros(A.P.total, dtype=torch.bool, device=A.P.param.device)
        for l, h in A.dp.stale_master_ranges():
            stale[l:h] = True
        nan = torch.isnan(A.P.param)
        assert bool(nan.any()) and not bool((nan & ~stale).any())
        so = torch.zeros(A.P.total, dtype=torch.bool, device=stale.device)
        for l, h, f in A.P.shadow_only_runs():
            so[l:h] = bool(f)
        assert not bool((stale & ~so).any())
        with pytest.raises(RuntimeError, match='gather_master'):
            A.state_dict()
        with pytest.raises(RuntimeError, match='gather_master'):
            next(iter(A.named_parameters()))
        # the shadow (what the convolutions read) and every master that is read agree with the lone network's
        assert rel(A.P.shadow.float(), R.P.shadow.float()) < 2e-3
        assert rel(A.P.param[~so], R.P.param[~so]) < 1e-4
        poison[0] = False
        A.dp.gather_master()
        assert not A.dp.master_stale and not bool(torch.isnan(A.P.param).any())
        assert rel(A.P.param, R.P.param) < 1e-4
        a_sd, r_sd = A.state_dict(), R.state_dict()
        assert all(rel(a_sd[k].float(), r_sd[k].float()) < 1e-3 for k in r_sd if float(r_sd[k].abs().max()) > 0)
    finally:
        parallel.dist = saved


def _two_rank_worker(rank, world, port, outdir, wire, variant, steps):
    """one REAL data-parallel rank (its own process, its own network, its own image) - both ranks on the box's one GPU, collectives staged
    through the host on a gloo group (GradReducer._staged)"""
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from lang2seg_amd.parallel import GradReducer
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    if variant == 'vgg':
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant)
    net = selftest.build_net(opt, dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64), 'bf16', sd, variant=variant)
    net.rank_seed = rank * 1000003
    net.use_tape = True                                          # the segmented launch tape, cut at the bucket hand-offs
    net.dp = GradReducer(net, world, wire=wire, algo='rs_ag', shard_update=True, rank=rank)
    sgd = SGD(net, 2e-3, grad_scale=1.0 / world)
    blob = OS.make_blob(320, 416, 6, 60, seed=5 + rank)          # every rank its own image and expression
    losses, p1 = [], None
    for i in range(steps):
        losses.append(net.train_step(dict(blob), 0, sgd))
        if i == 0:                                               # the weights after ONE step (a deterministic function of the two first gradients)
            torch.cuda.synchronize(); net.join_update(); torch.cuda.synchronize()
            net.dp.gather_master()
            p1 = net.P.param.cpu().clone()
    torch.cuda.synchronize(); net.join_update(); torch.cuda.synchronize()
    stale, n_stale = bool(net.dp.master_stale), sum(h - l for l, h in net.dp.stale_master_ranges())
    refused = False
    try:
        net.state_dict()
    except RuntimeError:
        refused = True
    net.dp.gather_master()                                       # a collective: both ranks are here
    torch.save(dict(param=net.P.param.cpu(), param1=p1, shadow=net.P.shadow.float().cpu(), losses=np.array(losses), stale=stale, n_stale=n_stale, refused=refused,
                    staged=bool(net.dp._host_staged)), os.path.join(outdir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('variant,wire', [('cycle', 'fp32'), ('cycle', 'bf16'), ('vgg', 'bf16')])      # (baseline-bf16 and cycle_response-fp32 pass too: 25 s each, left out of the suite)
def test_two_ranks_one_gpu_end_to_end(tmp_path, variant, wire):
    """Data parallel with two REAL ranks - two processes, two networks, two different (image, expression) pairs, the segmented launch tape,
    the sharded update with the master / shadow split - on the one GPU of the box (RCCL refuses two ranks on one device, so the wire is a gloo
    group with the reducer's buffers staged through the host: correctness of everything around the collectives, no statement about their speed).
    After two steps both ranks hold the same weights, bit for bit, and they are the weights of ONE process that adds the two images' gradients
    and updates once per step with grad_scale 1/2."""
    import os
    import torch.multiprocessing as mp
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from oracle import weights as OW, synth as OS
    steps, W = 2, 2
    ctx = mp.get_context('spawn')
    port = 29300 + os.getpid() % 500
    procs = [ctx.Process(target=_two_rank_worker, args=(r, W, port, str(tmp_path), wire, variant, steps)) for r in range(W)]
    for p_ in procs:
        p_.start()
    # ---- the single-process reference, meanwhile: one network per image (same weights, the ranks' RNG streams), gradients added, one update ----
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    if variant == 'vgg':
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant)
    nets, sgds, blobs = [], [], []
    for r in range(W):
        n_ = selftest.build_net(opt, dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64), 'bf16', sd, variant=variant)
        n_.rank_seed = r * 1000003; n_.use_tape = False
        nets.append(n_); sgds.append(SGD(n_, 2e-3, grad_scale=1.0 / W, keep_grad=True)); blobs.append(OS.make_blob(320, 416, 6, 60, seed=5 + r))
    p0 = nets[0].P.param.cpu().clone()
    ref_losses, ref_p1 = [], None
    for it_ in range(steps):
        row = []
        for n_, b_ in zip(nets, blobs):
            lv = n_.forward_backward(n_.upload_blob(dict(b_), 0))
            torch.cuda.synchronize()
            row.append(lv.cpu().numpy()[n_._loss_slots()].copy())
        gsum = nets[0].P.grad + nets[1].P.grad
        for n_, s_ in zip(nets, sgds):
            n_.P.grad.copy_(gsum)
            n_._step += 1
            s_.step()
        torch.cuda.synchronize()
        ref_losses.append(row)
        if it_ == 0:
            nets[0].join_update(); torch.cuda.synchronize()
            ref_p1 = nets[0].P.param.cpu().clone()
    for n_ in nets:
        n_.join_update()
    torch.cuda.synchronize()
    assert torch.equal(nets[0].P.param, nets[1].P.param)
    for p_ in procs:
        p_.join(600)
    assert all(p_.exitcode == 0 for p_ in procs), [p_.exitcode for p_ in procs]
    out = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r), weights_only=False) for r in range(W)]
    for r in range(W):
        assert out[r]['staged'] and out[r]['stale'] and out[r]['n_stale'] > 0 and out[r]['refused']      # masters of shadow-gathered slices WERE behind, state_dict() refused
        assert np.isfinite(out[r]['losses']).all()
    # both ranks: the same weights and the same shadow, bit for bit
    assert torch.equal(out[0]['param'], out[1]['param']) and torch.equal(out[0]['shadow'], out[1]['shadow'])
    # ... and after ONE step they are the single-process weights (fp32 wire: rounding of the sum only, measured 1e-4 of the movement; bf16 wire: each
    # rank's bucket is rounded to bf16 once).  Behind the second step only agreement between the ranks is exact: its gradients are taken at weights
    # that already differ by that rounding, and a sampled RoI that changes sides moves a loss by percents (measured: loss_box 0.49 against 0.56).
    assert torch.equal(out[0]['param1'], out[1]['param1'])
    moved, d = float((ref_p1 - p0).abs().max()), float((out[0]['param1'] - ref_p1).abs().max())
    assert moved > 0 and d < (2e-3 if wire == 'fp32' else 2e-2) * moved, (d, moved)
    moved2, d2 = float((nets[0].P.param.cpu() - p0).abs().max()), float((out[0]['param'] - nets[0].P.param.cpu()).abs().max())
    assert d2 < 0.5 * moved2, (d2, moved2)
    # every rank's first step saw the common initial weights: its losses are the reference network's of that image
    for r in range(W):
        assert np.allclose(out[r]['losses'][0], ref_losses[0][r], rtol=1e-4, atol=1e-5), (r, out[r]['losses'][0], ref_losses[0][r])
        assert abs(out[r]['losses'][1][-1] - ref_losses[1][r][-1]) < 0.1 * abs(ref_losses[1][r][-1]), (r, out[r]['losses'][1], ref_losses[1][r])
    # the two ranks really trained on different data
    assert not np.allclose(out[0]['losses'][0], out[1]['losses'][0], rtol=1e-3)


def test_bench_two_ranks_one_gpu():
    """bench.py's N > 1 branch as the driver launches it (torch.distributed.run, one process per rank, --gpus 2) with two ranks on the one GPU:
    tools/ab.py --dp-backend gloo stages the reducer's collectives through the host, everything else is the measured run's code - process group,
    per-rank loaders, the default reducer (bf16 buckets, reduce-scatter + all-gather, sharded update, five buckets), barriers, the MAX over ranks,
    ONE JSON line from rank 0.  The figures are marked invalid (no RCCL); the run must train (finite, moving losses) and count both ranks."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
           str(29700 + os.getpid() % 90), os.path.join(root, 'tools', 'ab.py'), '--dp-backend', 'gloo', '--',
           '--gpus', '2', '--steps', '6', '--warmup', '3', '--no-cpu-baseline', '--extras', '0', '--mixed-shapes', '0']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['scaling'] == 'weak' and d['value'] > 0 and 'INVALID' in d['experiment']
    assert 'dp2' in d['config']['parallelism'] and 'sharded update' in d['config']['parallelism'] and 'bf16' in d['config']['parallelism']
    assert abs(d['value'] - 2 * 1000.0 / d['ms_per_step']) < 1e-6 * d['value']           # whole-job images per second: both ranks' images over the slower rank's time
    assert all(np.isfinite(v) for v in d['final_losses'])


def test_train_entry_point_two_ranks_one_gpu():
    """tools/train_cycle_2.py as experiments/scripts/train_cycle.sh launches it for N > 1 (torch.distributed.run, one process per rank, RANK /
    WORLD_SIZE from the environment) with two ranks on the box's one GPU (TRAIN.DP_BACKEND gloo: the reducer's buffers staged through the host):
    the BASELINE-size network on the synthetic loader, per-rank shards, display, snapshots with per-rank sidecars (gather_master is a collective
    inside snapshot()), and a second launch that resumes both ranks from the newest snapshot."""
    import glob, os, shutil, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outd = os.path.join(root, 'refcoco_unc', 'output_t2r')
    shutil.rmtree(os.path.join(root, 'refcoco_unc', 'output_t2r'), ignore_errors=True)
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
            str(29800 + os.getpid() % 100), os.path.join(root, 'tools', 'train_cycle_2.py'), '--synthetic', '1', '--from_scratch', '1', '--synthetic_images', '4',
            '--output_postfix', 't2r']
    tail = ['--set', 'TRAIN.DP_BACKEND', 'gloo', 'TRAIN.SNAPSHOT_ITERS', '2', 'TRAIN.DISPLAY', '1']
    try:
        r = subprocess.run(base + ['--max_iters', '4'] + tail, capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        assert 'iter: 4 / 4' in r.stdout and 'done solving' in r.stdout
        names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(outd, '*')))
        for it in (2, 4):
            for sfx in ('.pth', '.pkl', '.rank1.pkl'):
                assert 'res101_mask_rcnn_iter_%d%s' % (it, sfx) in names, names
        tot = [float(l.split('total loss:')[1]) for l in r.stdout.splitlines() if 'total loss:' in l]
        assert len(tot) == 4 and all(np.isfinite(tot))
        r2 = subprocess.run(base + ['--max_iters', '6'] + tail, capture_output=True, text=True, timeout=900, cwd=root)
        assert r2.returncode == 0, (r2.stdout[-1500:], r2.stderr[-3000:])
        assert 'Restoring model snapshots' in r2.stdout and 'iter: 6 / 6' in r2.stdout and 'iter: 4 / 6' not in r2.stdout
        assert os.path.exists(os.path.join(outd, 'res101_mask_rcnn_iter_6.pth')) and os.path.exists(os.path.join(outd, 'res101_mask_rcnn_iter_6.rank1.pkl'))
    finally:
        shutil.rmtree(os.path.join(root, 'refcoco_unc', 'output_t2r'), ignore_errors=True)
        shutil.rmtree(os.path.join(root, 'refcoco_unc', 'tb_t2r'), ignore_errors=True)


@pytest.mark.parametrize('variant', ['cycle', 'vgg'])
def test_random_init_network_stays_finite(variant):
    """bench.py and the tools without a checkpoint run the network from its own initialisers: the frozen-BN trunk must keep its
    activations O(1) (a plain He-initialised ResNet-101 with identity BatchNorm on pixel-scale inputs overflows within one step and
    every later loss sits at its all-zero-logit value)."""
    from lang2seg_amd import selftest
    from lang2seg_amd.optim import SGD
    from oracle import weights as OW, synth as OS
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    if variant == 'vgg':
        opt['C4_feat_dim'] = 512
    net = selftest.build_net(opt, dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64), 'bf16', None, variant=variant)
    sgd = SGD(net, 1e-4)
    blob = OS.make_blob(320, 416, 6, 60, seed=5)
    for step in range(4):
        vals = net.train_step(dict(blob), 0, sgd)
        torch.cuda.synchronize()
        assert all(np.isfinite(v) for v in vals), (step, vals)
        for k in ('net_conv_base', 'net_conv'):
            a = net.t[k].float()
            assert bool(torch.isfinite(a).all()) and 1e-4 < float(a.abs().max()) < 1e3, (step, k, float(a.abs().max()))
    assert bool(torch.isfinite(net.P.param).all())


def test_bench_json_contract():
    """bench.py prints exactly one JSON line on stdout with the driver's keys (BASELINE.json's metric string verbatim, the roofline
    and cpu_baseline objects); run at the full BASELINE shape with a handful of steps."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '3', '--warmup', '1', '--cpu-baseline-steps', '1,1'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['metric'] == json.load(open(os.path.join(root, 'BASELINE.json')))['metric']
    for k in ('value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['vs_baseline'] is None and d['dtype'] == 'bf16' and d['scaling'] == 'weak'
    assert abs(d['value'] - 1000.0 / d['ms_per_step']) < 1e-6 * d['value'] and 'workload' in d['config'] and 'model' not in d['config']
    rf, cb = d['roofline'], d['cpu_baseline']
    # `roofline` = the TIME-DOMINANT group of convolution launches; `roofline.best` = the best kernel (layer4@RoIs 3x3)
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9 and rf['achieved'] > 20
    assert rf['ms_per_step'] == max(v['ms_per_step'] for v in rf['groups'].values()) and rf['launches_per_step'] >= 1 and 'igemm' in rf['kernel']
    bk = rf['best']
    assert abs(bk['frac'] - bk['achieved'] / bk['peak']) < 1e-9 and bk['achieved'] > 100 and 'igemm_dma_kernel' in bk['kernel']
    st = rf['stack3x3']
    assert abs(st['frac'] - st['achieved'] / st['peak']) < 1e-9 and 900 < st['gflop_per_step'] < 960 and st['launches_per_step'] > 60
    assert d['mixed_shapes']['value'] > 0 and len(d['mixed_shapes']['shapes']) == 6
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and cb['unit'] == 'img/s' and cb['sample'] and cb['cpu']
    # the synchronous train_step (the reference's unit as it stands) and the PCIe-inclusive rate ride along; neither is `value`
    assert 0 < d['sync_train_step']['value'] <= d['value'] * 1.05 and 0 < d['pcie_inclusive']['value'] <= d['value'] * 1.05
    assert d['pcie_inclusive']['h2d_bytes_per_step'] == 600 * 1000 * 3 * 4
    assert all(np.isfinite(v) for v in d['final_losses'])


def test_bench_workload_learns_when_pipelined():
    """the timed loop of bench.py (pipelined tape replays, update on the weight-gradient stream) must actually train: after 200 steps on its four
    synthetic images the RoI classification loss has left ln(81) = 4.394 and the RPN loss ln 2.  (Round 2: a tape recorded before the update
    moved to the weight-gradient stream cleared the gradients while the update was still reading them; every lr-0 parity test passed and
    the losses stayed at their initial values.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '200', '--warmup', '5', '--no-cpu-baseline', '--extras', '0'],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    rpn_cls, _, cls = d['final_losses'][:3]
    assert cls < 4.3 and rpn_cls < 0.69, d['final_losses']
