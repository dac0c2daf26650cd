#!/usr/bin/env python
"""Generate golden fixtures by running the REFERENCE ITSELF (read-only, imported from
/root/reference) in the dev container.  Runs only here (the GPU box has no reference);
the .npz outputs are committed next to this script.

The reference is Python-2 / torch-0.3 code.  Nothing under /root/reference is edited;
this harness only restores the environment it expects, in-process:
  * sys.modules stubs for easydict, tensorboardX, cv2, pycocotools, `_ext` (TH/THC C
    extensions that cannot be built here; `cpu_nms` is supplied by oracle.boxes.nms,
    restating nms.c:35-63) and scipy.misc.imresize (PIL-backed, scipy<=1.2 semantics);
  * np.float alias; Tensor.cuda()/Module.cuda() identity (no GPU here);
  * affine_grid/grid_sample default align_corners=True (torch 0.3 behaviour);
  * comparison results of `max_overlaps` in proposal_target_layer.py:143-146 returned
    as uint8 (torch-0.3 ByteTensor) so that `(a<b)+(c>=d)==2` keeps its meaning;
  * numpy.random.choice is wrapped only to RECORD what it drew (fixtures carry the
    draws as per-element priority keys).
`Network.forward()` is bypassed (its `.data[0]` raises on 0-dim tensors): the harness
sets the same attributes forward() sets (NET:632-648) and calls `_predict()` /
`_add_losses()` / backward / torch.optim.SGD exactly as NET:650-662,712-715, TV:194-220.

Usage: python tests/golden/make_golden.py [tiny|full|full_variants [names]|leaf|variants|test|all]
"""
import os
import sys
import types
import warnings
import numpy as np

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

import torch
import torch.nn as nn
import torch.nn.functional as F
from PIL import Image

from oracle import boxes as OB
from oracle import weights as OW
from oracle import synth as OS

CHOICE_LOG = []


def install_harness():
    # --- easydict ---
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            d = dict(d or {}, **kw)
            for k, v in d.items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            dict.__setitem__(self, k, v)
            object.__setattr__(self, k, v)
        __setitem__ = __setattr__
    m = types.ModuleType('easydict'); m.EasyDict = EasyDict; sys.modules['easydict'] = m

    # --- scipy.misc.imresize (scipy<=1.2 pilutil semantics) ---
    def bytescale(data):
        if data.dtype == np.uint8:
            return data
        cmin, cmax = data.min(), data.max()
        cscale = cmax - cmin
        if cscale == 0:
            cscale = 1
        scale = 255.0 / cscale
        return ((data - cmin) * scale + 0.5).clip(0, 255).astype(np.uint8)

    def imresize(arr, size, interp='bilinear', mode=None):
        im = Image.fromarray(bytescale(np.asarray(arr)))
        if isinstance(size, (int, np.integer)):
            size = tuple((np.array(im.size) * (size / 100.0)).astype(int))
        elif isinstance(size, float):
            size = tuple((np.array(im.size) * size).astype(int))
        else:
            size = (int(size[1]), int(size[0]))
        func = {'nearest': 0, 'lanczos': 1, 'bilinear': 2, 'bicubic': 3, 'cubic': 3}
        return np.array(im.resize(size, resample=func[interp]))
    import scipy.misc
    scipy.misc.imresize = imresize

    # --- empty third-party stubs ---
    for name in ['tensorboardX', 'cv2', 'pycocotools', 'pycocotools.mask']:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['pycocotools'].mask = sys.modules['pycocotools.mask']
    sys.modules['tensorboardX'].summary = types.SimpleNamespace()
    sys.modules['tensorboardX'].writer = types.SimpleNamespace()

    # --- native extensions (`from _ext import nms`, `from _ext import roi_pooling`) ---
    ext = types.ModuleType('_ext')

    def cpu_nms(keep_out, num_out, boxes, order, areas, thresh):
        dets = boxes.numpy()
        keep = OB.nms(dets, thresh, 'ge')
        keep_out[:len(keep)] = torch.from_numpy(keep)
        num_out[0] = len(keep)
        return 1
    ext.nms = types.SimpleNamespace(cpu_nms=cpu_nms)
    ext.roi_pooling = types.SimpleNamespace()
    sys.modules['_ext'] = ext

    # --- torchvision.models.vgg16 (not installed here): the standard configuration-D module tree (features / classifier) ---
    tv = types.ModuleType('torchvision'); tvm = types.ModuleType('torchvision.models')

    class _VGG(nn.Module):
        def __init__(self):
            nn.Module.__init__(self)
            layers, cin = [], 3
            for v in [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']:
                if v == 'M':
                    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                else:
                    layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                    cin = v
            self.features = nn.Sequential(*layers)
            self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(True),
                                            nn.Dropout(), nn.Linear(4096, 1000))
    tvm.vgg16 = lambda pretrained=False: _VGG()
    tv.models = tvm
    sys.modules['torchvision'] = tv; sys.modules['torchvision.models'] = tvm

    np.float = float
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    _ag, _gs = F.affine_grid, F.grid_sample
    F.affine_grid = lambda theta, size, align_corners=True: _ag(theta, size, align_corners=align_corners)
    F.grid_sample = lambda inp, grid, mode='bilinear', padding_mode='zeros', align_corners=True: \
        _gs(inp, grid, mode=mode, padding_mode=padding_mode, align_corners=align_corners)

    _choice = np.random.choice

    def rec_choice(a, size=None, replace=True, p=None):
        r = _choice(a, size=size, replace=replace, p=p)
        CHOICE_LOG.append(dict(a=np.array(a).copy(), size=size, replace=replace, result=np.array(r).copy()))
        return r
    np.random.choice = rec_choice

    sys.path.insert(0, os.path.join(REF, 'lib'))
    sys.path.insert(0, os.path.join(REF, 'pyutils/mask-faster-rcnn/lib'))

    # ByteTensor comparison semantics for proposal_target_layer.py:143-146
    import layer_utils.proposal_target_layer as PTL
    import utils.bbox as UB

    class ByteCmp(torch.Tensor):
        def __lt__(self, o):
            return torch.Tensor.__lt__(self.as_subclass(torch.Tensor), o).to(torch.uint8)

        def __ge__(self, o):
            return torch.Tensor.__ge__(self.as_subclass(torch.Tensor), o).to(torch.uint8)

    def bbox_overlaps_byte(b, q):
        return UB.bbox_overlaps(b, q).as_subclass(ByteCmp)
    PTL.bbox_overlaps = bbox_overlaps_byte


SOLVER_MODULE = dict(baseline='train_val', spatial='train_val', response='train_val_response', cycle='train_val_cycle',
                     cycle_response='train_val_cycle_response', vgg='train_val_vgg')     # tools/train*.py:22-23


class StubLoader(object):
    """what SolverWrapper.snapshot / from_snapshot touch of the loader (TV:74-77,152-158)"""

    def __init__(self, n_train=11, n_val=5, seed=5):
        rs = np.random.RandomState(seed)
        self.split_ix = {'train': list(range(n_train)), 'val': list(range(100, 100 + n_val))}
        self.iterators = {'train': 4, 'val': 2}
        self.perm = {'train': rs.permutation(n_train), 'val': rs.permutation(n_val)}


def reference_solver(variant, net, output_dir=None, loader=None):
    """the reference's SolverWrapper of this variant around `net` (not yet constructed: construct_graph() calls create_architecture)"""
    import importlib
    import tempfile
    tbx = sys.modules['tensorboardX']

    class FileWriter(object):
        def __init__(self, *a, **k):
            pass

        def add_summary(self, *a, **k):
            pass

        def close(self):
            pass
    tbx.writer.FileWriter = FileWriter
    TV = importlib.import_module('model.' + SOLVER_MODULE[variant])
    d = output_dir or tempfile.mkdtemp(prefix='l2s_ref_solver_')
    sw = TV.SolverWrapper(net, loader or StubLoader(), os.path.join(d, 'output'), os.path.join(d, 'tb'), pretrained_model=None)
    return sw, TV.cfg


def digest(t, nsamp=2048):
    a = t.detach().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    a = a.astype(np.float64).ravel()
    stride = max(1, a.size // nsamp)
    return dict(shape=np.array(t.shape), sum=a.sum(), abssum=np.abs(a).sum(),
                stride=stride, sample=a[::stride][:nsamp].astype(np.float32))


def flat(prefix, d, out):
    for k, v in d.items():
        out[prefix + '.' + k] = v


def keys_from_log(n_total, cand, drawn, disable_mode):
    """Turn one recorded npr.choice into uint32 priority keys (smallest first)."""
    keys = np.full(n_total, 0xFFFFFFFF, dtype=np.uint32)
    drawn = np.asarray(drawn).ravel()
    rest = np.setdiff1d(np.asarray(cand), drawn, assume_unique=False)
    keys[drawn] = np.arange(len(drawn), dtype=np.uint32)
    keys[rest] = len(drawn) + np.arange(len(rest), dtype=np.uint32)
    return keys


def run_reference(tag, H, W, T, V, cfg_over, seed_w=3, seed_blob=1234, head_gain=4.0, full_tensors=True, variant='cycle', top_over=None, resnet_over=None):
    from model.config import cfg
    import importlib
    var = OW.VARIANTS[variant]
    RESM = importlib.import_module('nets.' + var['module'])
    NETM = importlib.import_module('nets.' + var['net'])
    from oracle.net import DEFAULT_CFG
    import copy
    ocfg = copy.deepcopy(DEFAULT_CFG)
    for k, v in cfg_over.items():
        ocfg['TRAIN'][k] = v
        setattr(cfg.TRAIN, k, v)
    cfg.ANCHOR_SCALES = list(ocfg['ANCHOR_SCALES']); cfg.ANCHOR_RATIOS = list(ocfg['ANCHOR_RATIOS'])
    top_saved = {}
    for k, v in (top_over or {}).items():            # top-level switches, e.g. POOLING_ALIGN (NET:569-570)
        top_saved[k] = getattr(cfg, k); setattr(cfg, k, v); ocfg[k] = v
    for k, v in (resnet_over or {}).items():         # cfg.RESNET.* (FIXED_BLOCKS: RES:290-299 freezes layer1..FIXED_BLOCKS)
        top_saved['RESNET.' + k] = getattr(cfg.RESNET, k); setattr(cfg.RESNET, k, v); ocfg[k] = v
    opt = OW.default_opt(vocab_size=V, seq_length=T)
    is_vgg = var.get('backbone') == 'vgg'
    if is_vgg:
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=seed_w, head_gain=head_gain, variant=variant)
    blob = OS.make_blob(H, W, T, V, seed=seed_blob)

    torch.manual_seed(0)
    net = RESM.vgg16(opt, batch_size=1) if is_vgg else RESM.resnetv1(opt, batch_size=1, num_layers=101)
    # The optimiser is the one the reference's OWN solver of this variant builds: SolverWrapper.construct_graph() (train_val.py:167-214 for
    # baseline / spatial, train_val_response.py, train_val_cycle.py, train_val_cycle_response.py, train_val_vgg.py) creates the architecture,
    # then one param group per tensor - with lr x 10 on rnn_encoder / dynamic_fc / response keys in four of the six solvers, and
    # config_vgg's WEIGHT_DECAY / DOUBLE_BIAS for the VGG one (train_val_vgg.py:12).  Only tensorboardX's FileWriter is a stub.
    sw, scfg = reference_solver(variant, net)
    scfg.ANCHOR_SCALES = list(cfg.ANCHOR_SCALES); scfg.ANCHOR_RATIOS = list(cfg.ANCHOR_RATIOS)     # (tools/train*.py --set, the solver's own cfg)
    lr0, optimizer = sw.construct_graph()
    ref_sd = net.state_dict()
    for k, v in sd.items():
        assert k in ref_sd and tuple(ref_sd[k].shape) == v.shape, k
        ref_sd[k].copy_(torch.from_numpy(v))
    group_of = {id(g['params'][0]): g for g in optimizer.param_groups}
    assert all(len(g['params']) == 1 for g in optimizer.param_groups)
    net.train()
    for mod in net.modules():                       # dropout off (parity runs inject masks = identity)
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    np.random.seed(cfg.RNG_SEED)
    del CHOICE_LOG[:]
    # NET:632-648 attribute setup (forward() bypassed)
    net._image = torch.from_numpy(blob['data'].transpose([0, 3, 1, 2]).copy())
    net._im_info = blob['im_info']
    net._gt_boxes = torch.from_numpy(blob['gt_boxes'])
    net._gt_masks = blob['gt_masks']
    net._labels = torch.from_numpy(blob['labels'])
    net._cap_labels = blob['cap_labels']; net._cap_masks = blob['cap_masks']
    net._mode = 'TRAIN'
    net._image_gt_summaries = {}
    pred_out = net._predict()
    net_conv = pred_out[0]
    net._predictions['net_conv'] = net_conv
    net._add_losses()
    L = {k: float(v) for k, v in net._losses.items()}
    optimizer.zero_grad()
    net._losses['total_loss'].backward()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters() if p.requires_grad}
    optimizer.step()

    out = dict(meta_H=H, meta_W=W, meta_T=T, meta_V=V, meta_seed_w=seed_w, meta_seed_blob=seed_blob,
               meta_head_gain=head_gain, meta_variant=variant)
    for k, v in cfg_over.items():
        out['cfg.' + k] = v
    # the solver's configuration and its param-group table, as the reference built them
    for k in ('LEARNING_RATE', 'MOMENTUM', 'WEIGHT_DECAY', 'DOUBLE_BIAS', 'BIAS_DECAY', 'GAMMA'):
        out['solver.' + k] = getattr(scfg.TRAIN, k)
    out['solver.module'] = SOLVER_MODULE[variant]
    names_tr = [k for k, p in net.named_parameters() if p.requires_grad]
    out['solver.keys'] = np.array(names_tr)
    out['solver.lr'] = np.array([group_of[id(p)]['lr'] for k, p in net.named_parameters() if p.requires_grad], np.float64)
    out['solver.wd'] = np.array([group_of[id(p)]['weight_decay'] for k, p in net.named_parameters() if p.requires_grad], np.float64)
    assert float(lr0) == float(scfg.TRAIN.LEARNING_RATE) and optimizer.defaults['momentum'] == scfg.TRAIN.MOMENTUM
    for k, v in L.items():
        out['loss.' + k] = v
    # sampling draws -> keys
    Hc, Wc = net_conv.shape[2], net_conv.shape[3]
    A = net._num_anchors
    n_anchor = Hc * Wc * A
    anchors = net._anchors.numpy()
    inside = np.where((anchors[:, 0] >= 0) & (anchors[:, 1] >= 0) & (anchors[:, 2] < W) & (anchors[:, 3] < H))[0]
    log = list(CHOICE_LOG)
    # order of calls in one TRAIN forward: [ATL fg?] [ATL bg?] PTL fg, PTL bg
    n_post = int(net._predictions['rois'].shape[0])  # sampled rois (256); proposals count recorded below
    ptl_bg = log[-1]; ptl_fg = log[-2]; atl = log[:-2]
    rpn_fg_keys = np.full(n_anchor, 0xFFFFFFFF, np.uint32)
    rpn_bg_keys = np.full(n_anchor, 0xFFFFFFFF, np.uint32)
    rpn_lab = net._anchor_targets['rpn_labels'].numpy().reshape(A, Hc, Wc).transpose(1, 2, 0).reshape(-1)
    for e in atl:
        cand_inside = e['a']                                   # indices into `inside`
        cand = inside[cand_inside]
        drawn = inside[e['result']]
        keys = keys_from_log(n_anchor, cand, drawn, True)
        # fg call draws from labels==1 candidates, bg call from labels==0
        if len(atl) == 2 and e is atl[0]:
            rpn_fg_keys = keys
        else:
            # single call: decide by whether candidates contain a positive anchor
            if (rpn_lab[cand] == 1).any() and not (rpn_lab[cand] == 0).any() and len(atl) == 1 and len(cand) < 2000:
                rpn_fg_keys = keys
            else:
                rpn_bg_keys = keys
    out['samp.rpn_fg_keys'] = rpn_fg_keys; out['samp.rpn_bg_keys'] = rpn_bg_keys
    # PTL draws are positions into fg_inds / bg_inds of the proposal list
    prop = net._proposal_targets
    # recompute candidate lists from the reference's own proposals
    from utils.bbox import bbox_overlaps as ub
    all_rois = PROPOSALS['rois']
    ov = ub(torch.from_numpy(all_rois[:, 1:5]), torch.from_numpy(blob['gt_boxes'][:, :4])).numpy()
    mx = ov.max(1)
    fg_inds = np.where(mx >= cfg.TRAIN.FG_THRESH)[0]
    bg_inds = np.where((mx < cfg.TRAIN.BG_THRESH_HI) & (mx >= cfg.TRAIN.BG_THRESH_LO))[0]
    n_prop = all_rois.shape[0]
    print('n_prop', n_prop, 'fg', len(fg_inds), 'bg', len(bg_inds), 'log', [(len(e['a']), e['size'], e['replace']) for e in log])
    if len(fg_inds) == 0:
        # PTL:159-167: no proposal overlaps the GT box -> the reference appends the GT boxes to the roi list and retries; the
        # only fg candidates are the appended rows (the oracle / kernels give them key 0), proposals keep their bg draws
        assert len(ptl_fg['a']) == blob['gt_boxes'].shape[0] and len(ptl_bg['a']) == len(bg_inds)
        out['samp.roi_fg_keys'] = np.full(n_prop, 0xFFFFFFFF, np.uint32)
    else:
        assert len(ptl_fg['a']) == len(fg_inds) and len(ptl_bg['a']) == len(bg_inds), (len(ptl_fg['a']), len(fg_inds), len(ptl_bg['a']), len(bg_inds))
        out['samp.roi_fg_keys'] = keys_from_log(n_prop, fg_inds, fg_inds[ptl_fg['result']], False)
    assert not ptl_bg['replace'], 'fixture expects the without-replacement path'
    out['samp.roi_bg_keys'] = keys_from_log(n_prop, bg_inds, bg_inds[ptl_bg['result']], False)
    out['int.proposal_rois'] = all_rois
    out['int.proposal_scores'] = PROPOSALS['scores']
    out['int.rpn_labels'] = net._anchor_targets['rpn_labels'].numpy().astype(np.int8)
    out['int.rois'] = prop['rois'].detach().numpy()
    out['int.labels'] = prop['labels'].numpy().astype(np.int64).reshape(-1)
    out['int.mask_targets'] = prop['mask_targets'].numpy().astype(np.uint8)
    out['int.num_fg'] = prop['mask_targets'].shape[0]
    tens = dict(net_conv=net_conv, response=net._predictions.get('response'), rpn_cls_prob=net._predictions['rpn_cls_prob'],
                rpn_bbox_pred=net._predictions['rpn_bbox_pred'], rpn_bbox_targets=net._anchor_targets['rpn_bbox_targets'],
                rpn_bbox_outside=net._anchor_targets['rpn_bbox_outside_weights'],
                bbox_targets=prop['bbox_targets'], cls_score=net._predictions['cls_score'],
                bbox_pred=net._predictions['bbox_pred'], mask_score=net._predictions.get('mask_score'))
    for k, v in tens.items():
        if v is not None:
            flat('t.' + k, digest(v), out)
    # a few exact small tensors
    out['x.cls_score'] = net._predictions['cls_score'].detach().numpy()[:, :8]
    out['x.bbox_pred'] = net._predictions['bbox_pred'].detach().numpy()[:8, :16]
    gsel = ['resnet.layer2.0.conv1.weight', 'resnet.layer3.22.conv2.weight', 'resnet.layer4.2.conv3.weight',
            'resnet.layer4.0.downsample.0.weight', 'rpn_net.weight', 'rpn_cls_score_net.bias', 'cls_score_net.weight',
            'bbox_pred_net.bias', 'mask_up_sampling.weight', 'mask_pred_net.weight', 'dynamic_fc_3.weight',
            'response_fc.weight', 'rnn_encoder.embedding.weight', 'rnn_encoder.rnn.weight_hh_l0_reverse',
            'rnn_encoder.mlp.0.bias', 'caption_model.att_embed.0.weight', 'caption_model.logit.weight',
            'caption_model.core.h2h.weight', 'caption_model.core.a2c.bias', 'caption_model.core.attention.alpha_net.weight',
            'caption_model.embed.0.weight', 'caption_model.ctx2att.weight']
    if is_vgg:
        gsel = ['vgg.features.10.weight', 'vgg.features.17.bias', 'vgg.features.28.weight', 'vgg.classifier.0.weight', 'vgg.classifier.3.bias',
                'rpn_net.weight', 'rpn_cls_score_net.bias', 'cls_score_net.weight', 'bbox_pred_net.bias', 'dynamic_fc_3.weight', 'response_fc.weight',
                'rnn_encoder.embedding.weight', 'rnn_encoder.rnn.weight_hh_l0_reverse', 'rnn_encoder.mlp.0.bias']
    if (resnet_over or {}).get('FIXED_BLOCKS', 1) == 0:
        gsel = gsel + ['resnet.layer1.0.conv1.weight', 'resnet.layer1.2.conv2.weight', 'resnet.layer1.0.downsample.0.weight']
    if var['nfilt'] == 1:
        gsel = [k for k in gsel if not k.startswith(('dynamic_fc_', 'response_fc'))] + ['dynamic_fc.weight', 'dynamic_fc.bias']
    if var['cap'] is None:
        gsel = [k for k in gsel if not k.startswith('caption_model.')]
    for k in gsel:
        g = grads[k]
        flat('g.' + k, digest(g if g is not None else torch.zeros(1)), out)
        flat('w1.' + k, digest(dict(net.named_parameters())[k]), out)
    for k, v in (top_over or {}).items():
        out['top.' + k] = int(v)
    for k, v in (resnet_over or {}).items():
        out['resnet.' + k] = int(v)
    for k, v in top_saved.items():
        if k.startswith('RESNET.'):
            setattr(cfg.RESNET, k[7:], v)
        else:
            setattr(cfg, k, v)
    np.savez_compressed(os.path.join(HERE, 'ref_%s.npz' % tag), **out)
    print(tag, 'losses', L, 'num_fg', out['int.num_fg'], 'n_prop', n_prop, 'choices', [(len(e['a']), e['size']) for e in log])
    return out


def run_reference_test(tag, H, W, T, V, seed_w=3, seed_blob=1234, head_gain=4.0, variant='cycle', test_mode='nms', top_n=0):
    """TEST mode (test_image, NET:684-699; _predict_masks_from_boxes_and_labels, NET:595-626) of the reference."""
    from model.config import cfg
    import importlib
    var = OW.VARIANTS[variant]
    RESM = importlib.import_module('nets.' + var['module'])
    opt = OW.default_opt(vocab_size=V, seq_length=T)
    if var.get('backbone') == 'vgg':
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=seed_w, head_gain=head_gain, variant=variant)
    blob = OS.make_blob(H, W, T, V, seed=seed_blob)
    from oracle.net import DEFAULT_CFG
    for k, v in DEFAULT_CFG['TEST'].items():
        setattr(cfg.TEST, k, v)
    for k in ['BATCH_SIZE', 'RPN_PRE_NMS_TOP_N', 'RPN_POST_NMS_TOP_N', 'RPN_BATCHSIZE']:
        setattr(cfg.TRAIN, k, DEFAULT_CFG['TRAIN'][k])
    cfg.ANCHOR_SCALES = list(DEFAULT_CFG['ANCHOR_SCALES']); cfg.ANCHOR_RATIOS = list(DEFAULT_CFG['ANCHOR_RATIOS'])
    cfg.TEST.MODE = test_mode                       # 'top': proposal_top_layer instead of proposal_layer (NET:261-266)
    if top_n:
        cfg.TEST.RPN_TOP_N = top_n
    torch.manual_seed(0)
    is_vgg = var.get('backbone') == 'vgg'
    net = RESM.vgg16(opt, batch_size=1) if is_vgg else RESM.resnetv1(opt, batch_size=1, num_layers=101)
    net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    ref_sd = net.state_dict()
    for k, v in sd.items():
        ref_sd[k].copy_(torch.from_numpy(v))
    net.eval()
    # NET:632-662 with mode == 'TEST' (forward() itself is bypassed: `.data[0]` on a 0-dim tensor)
    net._image = torch.from_numpy(blob['data'].transpose([0, 3, 1, 2]).copy())
    net._im_info = blob['im_info']
    net._gt_boxes = torch.from_numpy(blob['gt_boxes'])
    net._gt_masks = blob['gt_masks']
    net._labels = torch.from_numpy(blob['labels'])
    net._cap_labels = None; net._cap_masks = None
    net._mode = 'TEST'
    net._image_gt_summaries = {}
    with torch.no_grad():
        pr = net._predict()
        net_conv, rois, cls_prob, bbox_pred = pr[:4]
        mask_prob = pr[4] if len(pr) > 4 else None          # network_vgg.py:614 returns no mask_prob
        stds = bbox_pred.data.new(cfg.TRAIN.BBOX_NORMALIZE_STDS).repeat(net._num_classes).unsqueeze(0).expand_as(bbox_pred)
        means = bbox_pred.data.new(cfg.TRAIN.BBOX_NORMALIZE_MEANS).repeat(net._num_classes).unsqueeze(0).expand_as(bbox_pred)
        bbox_pred = bbox_pred.mul(stds).add(means)
        boxes = rois.numpy()[:5, 1:5].copy()
        labels = np.array([3, 17, 1, 80, 42])
        masks = net._predict_masks_from_boxes_and_labels(net_conv, boxes, labels) if mask_prob is not None else None
    out = dict(meta_H=H, meta_W=W, meta_T=T, meta_V=V, meta_seed_w=seed_w, meta_seed_blob=seed_blob, meta_head_gain=head_gain,
               meta_variant=variant, meta_test_mode=test_mode, meta_top_n=top_n)
    cfg.TEST.MODE = 'nms'
    out['int.rois'] = rois.numpy()
    out['x.cls_score'] = net._predictions['cls_score'].numpy()
    out['x.cls_prob'] = cls_prob.numpy()
    out['x.bbox_pred'] = bbox_pred.numpy()[:, :24]
    flat('t.bbox_pred', digest(bbox_pred), out)
    flat('t.net_conv', digest(net_conv), out)
    if mask_prob is not None:
        flat('t.mask_prob', digest(mask_prob), out)
        out['x.mask_prob_0'] = mask_prob.numpy()[:4, :6]
        out['pm.boxes'] = boxes; out['pm.labels'] = labels; out['pm.masks'] = masks.numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_%s.npz' % tag), **out)
    print(tag, 'rois', rois.shape, 'cls_prob max', float(cls_prob.max()))
    return out


PROPOSALS = {}


def hook_proposals(variant='cycle'):
    """Record what proposal_layer returned (its output is consumed, not stored, by NET:256-259)."""
    import importlib
    NETM = importlib.import_module('nets.' + OW.VARIANTS[variant]['net'])
    if getattr(NETM, '_l2s_hooked', False):
        return
    NETM._l2s_hooked = True
    orig = NETM.proposal_layer

    def rec(*a, **k):
        rois, scores = orig(*a, **k)
        PROPOSALS['rois'] = rois.detach().numpy().copy()
        PROPOSALS['scores'] = scores.detach().numpy().reshape(-1).copy()
        return rois, scores
    NETM.proposal_layer = rec


def run_leaf():
    """Leaf functions imported straight from the reference (no substitution)."""
    from layer_utils.snippets import generate_anchors_pre
    from layer_utils.generate_anchors import generate_anchors
    from model.bbox_transform import bbox_transform, bbox_transform_inv, clip_boxes
    from utils.bbox import bbox_overlaps
    rs = np.random.RandomState(7)
    out = {}
    out['anchors.base'] = generate_anchors()
    a, n = generate_anchors_pre(5, 7, [16, ], (4, 8, 16, 32), (0.5, 1, 2))
    out['anchors.pre_5x7'] = a
    ex = rs.uniform(0, 300, (50, 4)).astype(np.float32); ex[:, 2:] += ex[:, :2]
    gt = rs.uniform(0, 300, (50, 4)).astype(np.float32); gt[:, 2:] += gt[:, :2]
    out['bt.ex'] = ex; out['bt.gt'] = gt
    out['bt.targets'] = bbox_transform(torch.from_numpy(ex), torch.from_numpy(gt)).numpy()
    d = rs.normal(0, 0.5, (50, 4)).astype(np.float32)
    out['bt.deltas'] = d
    inv = bbox_transform_inv(torch.from_numpy(ex), torch.from_numpy(d))
    out['bt.inv'] = inv.numpy()
    out['bt.clip'] = clip_boxes(inv, (200, 320)).numpy()
    out['bt.iou'] = bbox_overlaps(torch.from_numpy(ex), torch.from_numpy(gt[:7])).numpy()
    # language encoder + captioner + criterion, small vocab, imported as-is
    from layers.lang_encoder import RNNEncoder
    import caption_models
    import misc.utils as mutils
    opt = OW.default_opt(vocab_size=37, seq_length=6)
    sd = OW.make_state_dict(opt, seed=11)
    enc = RNNEncoder(37, 512, 512, 512, bidirectional=True, input_dropout_p=0.0, dropout_p=0.0, n_layers=1,
                     rnn_type='lstm', variable_lengths=True)
    enc.load_state_dict({k[len('rnn_encoder.'):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith('rnn_encoder.')})
    labels = torch.from_numpy(rs.randint(1, 37, (1, 6)).astype(np.int64))
    _, hidden, _ = enc(labels)
    out['enc.labels'] = labels.numpy(); out['enc.hidden'] = hidden.detach().numpy()
    cap = caption_models.setup(opt)
    cap.load_state_dict({k[len('caption_model.'):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith('caption_model.')})
    for mod in cap.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    att = torch.from_numpy(rs.normal(0, 1, (1, 14, 14, 4096)).astype(np.float32))
    seq = np.zeros((1, 8), np.int64); seq[0, 1:7] = rs.randint(1, 37, 6)
    lp = cap(torch.zeros(1, 4096), att, torch.from_numpy(seq))
    msk = torch.ones(1, 8)
    loss = mutils.LanguageModelCriterion()(lp, torch.from_numpy(seq)[:, 1:], msk[:, 1:])
    out['cap.att'] = att.numpy().astype(np.float16); out['cap.seq'] = seq
    out['cap.logprobs'] = lp.detach().numpy(); out['cap.loss'] = float(loss)
    np.savez_compressed(os.path.join(HERE, 'ref_leaf.npz'), **out)
    print('leaf done', out['cap.loss'])


def run_leaf_eval():
    """Post-processing leaves of the evaluation path, imported from the reference (model/test.py, utils/mask_utils.py)."""
    import importlib
    from utils.mask_utils import recover_masks
    from model.bbox_transform import bbox_transform_inv
    sys.modules['cv2'].resize = None
    sys.modules['pycocotools.mask'].encode = None
    # model/test.py imports utils.visualization (PIL fonts) and nms_wrapper: import lazily and tolerate their absence
    rs = np.random.RandomState(21)
    out = {}
    masks = rs.uniform(0, 1, (6, 14, 14)).astype(np.float32)
    masks[3] = 0.25                                            # constant mask: bytescale's cscale == 0 branch
    rois = np.array([[10.3, 20.7, 80.2, 90.9], [-5.0, -3.0, 40.0, 33.3], [100.0, 50.0, 219.9, 146.9], [0, 0, 13, 13],
                     [150.5, 100.5, 400.0, 300.0], [30, 40, 30.4, 40.2]], np.float32)
    out['rm.masks'] = masks.copy(); out['rm.rois'] = rois.copy()
    rec = recover_masks(masks.copy(), rois.copy(), 147, 220)
    out['rm.out'] = rec
    out['rm.bin'] = (rec > 122.).astype(np.uint8)
    boxes = rs.uniform(0, 200, (20, 4)).astype(np.float32); boxes[:, 2:] += boxes[:, :2]
    deltas = rs.normal(0, 0.3, (20, 4 * 5)).astype(np.float32)
    out['bt.boxes'] = boxes; out['bt.deltas'] = deltas
    out['bt.pred'] = bbox_transform_inv(torch.from_numpy(boxes), torch.from_numpy(deltas)).numpy()
    import scipy.misc
    gm = (rs.uniform(0, 1, (150, 230)) > 0.6).astype(np.uint8)
    out['nn.mask'] = gm
    out['nn.out'] = scipy.misc.imresize(gm, size=(94, 147), interp='nearest')
    np.savez_compressed(os.path.join(HERE, 'ref_leaf_eval.npz'), **out)
    print('leaf_eval done', rec.shape, int(rec.sum()))


class _Labels(object):
    """the loader hands eval_split a torch-0.3 Variable of token ids; the loop only slices it and evaluates
    `(label != 0).sum().data[0]` (test.py:236), which has no equivalent on 0-dim tensors of a current torch: same idiom, numpy inside"""

    def __init__(self, a):
        self.a = np.asarray(a)
        self.shape = self.a.shape

    def __getitem__(self, k):
        return _Labels(self.a[k])

    def __ne__(self, v):
        return _Labels(self.a != v)

    def sum(self):
        class _S(object):
            pass
        r = _S(); r.data = [int(self.a.sum())]
        return r


def eval_blobs(seed0=41, n_img=2, n_sent=2, H=160, W=224, T=6, V=60):
    """the synthetic evaluation set of ref_eval_split.npz: n_img images with n_sent referred objects / expressions each"""
    out = []
    for i in range(n_img):
        bl = [OS.make_blob(H, W, T, V, seed=seed0 + 10 * i + j) for j in range(n_sent)]
        lab = np.zeros((n_sent, T), np.int64)
        for j, b in enumerate(bl):
            n = T - (i + j) % 3                     # ragged lengths: zero padding at the end of a row
            lab[j, :n] = b['labels'][0, :n]
        gtb = np.concatenate([b['gt_boxes'] for b in bl]); gtm = np.concatenate([b['gt_masks'] for b in bl])
        # the referred objects sit where this untrained network's best anchors fall (right edge of the image), with different overlaps:
        # the fixture is about the accumulation of box accuracy / intersection / union, which needs non-empty intersections
        for j in range(n_sent):
            x1, y1, x2, y2 = [(188, 90, 223, 159), (150, 60, 223, 159), (196, 100, 223, 150), (120, 0, 223, 70)][(2 * i + j) % 4]
            gtb[j, :4] = [x1, y1, x2, y2]
            gtm[j] = 0; gtm[j, y1 + (j % 2) * 8:y2 + 1, x1 + 4 * i:x2 + 1] = 1
        out.append(dict(data=bl[0]['data'], im_info=bl[0]['im_info'], gt_boxes=gtb, gt_masks=gtm, labels=lab, file_name='synthetic_%d.jpg' % i))
    return out


def eval_state_dict_vgg(opt):
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant='vgg')
    for k in ('bbox_pred_net', 'rpn_bbox_pred_net'):          # RoIs = anchors, predicted boxes = RoIs (as eval_state_dict)
        sd[k + '.weight'] = np.zeros_like(sd[k + '.weight'])
        sd[k + '.bias'] = np.zeros_like(sd[k + '.bias'])
    return sd


def eval_state_dict(opt):
    """weights of the eval_split fixture: the usual random set with both box-regression heads zeroed, so that the predicted box is the
    chosen anchor itself (untrained regressors throw every box to the image border and all intersections are empty)"""
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    for k in ('bbox_pred_net', 'rpn_bbox_pred_net'):          # RoIs = anchors, predicted boxes = RoIs
        sd[k + '.weight'] = np.zeros_like(sd[k + '.weight'])
        sd[k + '.bias'] = np.zeros_like(sd[k + '.bias'])
    return sd


def run_eval_split():
    """the reference's evaluation loop itself (model/test.py:185-450 eval_split, :97-129 im_detect, mask_utils.recover_masks, NET:595-626)
    on a tiny synthetic split: box accuracy, precision@X counts, cumulative intersection / union.  Only Network.test_image is replaced by
    the harness's bypass of forward() (as in run_reference_test); everything else runs as it is."""
    import importlib
    from model.config import cfg
    import model.test as MT
    from oracle.net import DEFAULT_CFG
    sys.modules['cv2'].resize = None
    H, W, T, V = 160, 224, 6, 60
    opt = OW.default_opt(vocab_size=V, seq_length=T)
    sd = eval_state_dict(opt)
    for k, v in DEFAULT_CFG['TEST'].items():
        setattr(cfg.TEST, k, v)
    cfg.TEST.MODE = 'nms'
    cfg.ANCHOR_SCALES = list(DEFAULT_CFG['ANCHOR_SCALES']); cfg.ANCHOR_RATIOS = list(DEFAULT_CFG['ANCHOR_RATIOS'])
    RESM = importlib.import_module('nets.' + OW.VARIANTS['cycle']['module'])
    torch.manual_seed(0)
    net = RESM.resnetv1(opt, batch_size=1, num_layers=101)
    net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    ref_sd = net.state_dict()
    for k, v in sd.items():
        ref_sd[k].copy_(torch.from_numpy(v))
    net.eval()

    def test_image(blobs):                      # NET:671-699 with forward() bypassed (NET:632-662, mode TEST)
        net._image = torch.from_numpy(np.ascontiguousarray(blobs['data'].transpose([0, 3, 1, 2])))
        net._im_info = blobs['im_info']
        net._gt_boxes = torch.from_numpy(blobs['gt_boxes'])
        net._gt_masks = blobs['gt_masks']
        net._labels = torch.from_numpy(np.asarray(blobs['labels'].a))
        net._cap_labels = None; net._cap_masks = None
        net._mode = 'TEST'
        net._image_gt_summaries = {}
        with torch.no_grad():
            net_conv, rois, cls_prob, bbox_pred, mask_prob = net._predict()
            stds = bbox_pred.data.new(cfg.TRAIN.BBOX_NORMALIZE_STDS).repeat(net._num_classes).unsqueeze(0).expand_as(bbox_pred)
            means = bbox_pred.data.new(cfg.TRAIN.BBOX_NORMALIZE_MEANS).repeat(net._num_classes).unsqueeze(0).expand_as(bbox_pred)
            bbox_pred = bbox_pred.mul(stds).add(means)
        return (net._predictions['cls_score'].numpy(), cls_prob.numpy(), bbox_pred.numpy(), rois.numpy(), net_conv)
    net.test_image = test_image
    pm = net._predict_masks_from_boxes_and_labels
    net._predict_masks_from_boxes_and_labels = lambda nc, b, l: pm(nc, b, l).detach()

    imgs = eval_blobs(H=H, W=W, T=T, V=V)

    class Loader(object):
        def __init__(self):
            self.i = 0

        def getTestBatch(self, split):
            b = dict(imgs[self.i]); self.i += 1
            b['labels'] = _Labels(b['labels'])
            b['bounds'] = dict(it_pos_now=self.i, it_max=len(imgs), wrapped=self.i >= len(imgs))
            return b
    picked = []
    orig_detect = MT.im_detect

    def rec_detect(model, blobs):
        r = orig_detect(model, blobs)
        sc, bx = r[0], r[1]
        pr = np.where(sc == np.max(sc[:, 1:]))
        picked.append((int(pr[0][0]), int(pr[1][0]), bx[pr[0][0], pr[1][0] * 4:(pr[1][0] + 1) * 4].copy()))
        return r
    MT.im_detect = rec_detect
    with torch.no_grad():
        acc, thr, seg_correct, seg_total, cum_I, cum_U, num_sent = MT.eval_split(Loader(), net, None, 'val', dict(verbose=False))
    MT.im_detect = orig_detect
    out = dict(acc=float(acc), thr=np.asarray(thr, np.float64), seg_correct=np.asarray(seg_correct), seg_total=int(seg_total), cum_I=int(cum_I),
               cum_U=int(cum_U), num_sent=int(num_sent), pred_roi=np.array([p[0] for p in picked]), pred_class=np.array([p[1] for p in picked]),
               pred_box=np.stack([p[2] for p in picked]).astype(np.float32))
    np.savez_compressed(os.path.join(HERE, 'ref_eval_split.npz'), **out)
    print('eval_split: acc %.3f seg_correct %s / %d  I %d U %d  sents %d' % (acc, list(seg_correct), seg_total, cum_I, cum_U, num_sent))
    print(out['pred_class'], out['pred_box'])


SNAP_DIR = os.path.join(HERE, 'ref_snapshot')


def _pth_records(path):
    import zipfile
    with zipfile.ZipFile(path) as z:
        return [(i.filename, i.file_size, i.CRC, i.compress_type) for i in z.infolist()], {i.filename: z.read(i.filename) for i in z.infolist()
                                                                                          if '/data/' not in i.filename}


def run_snapshot():
    """f3: a snapshot PAIR written by the reference's own SolverWrapper.snapshot() (train_val_cycle.py:57-104) and what its from_snapshot()
    (:106-165) makes of it - full match, and the `[:, :-1]` partial rule (:121-124) on a file whose rpn_net.weight lacks the last input channel.
    The .pth of the tiny cycle network is ~230 MB, so what is committed is everything of it EXCEPT the tensor payloads: the zip's non-payload
    records as the reference wrote them (data.pkl = the pickled OrderedDict structure with its storage references, version, byteorder) and a
    manifest with every payload record's size + CRC-32 and the key it belongs to.  The payloads are the deterministic synthetic weights
    (oracle.weights.make_state_dict(seed=3, head_gain=4); keys the generator does not cover are zeroed before the snapshot or small enough to
    sit in the manifest), so tests/test_snapshot_*.py rebuild the file bit for bit (checked record by record against the CRCs) and hand it to the
    build's from_snapshot.  The .pkl sidecar is committed as the reference wrote it."""
    import base64
    import json
    import random
    import shutil
    import tempfile
    import zlib
    from model.config import cfg
    import importlib
    H, W, T, V = 320, 416, 6, 60
    variant = 'cycle'
    var = OW.VARIANTS[variant]
    RESM = importlib.import_module('nets.' + var['module'])
    opt = OW.default_opt(vocab_size=V, seq_length=T)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant)
    tmp = tempfile.mkdtemp(prefix='l2s_ref_snapshot_')

    def fresh(loader):
        torch.manual_seed(0)
        net = RESM.resnetv1(opt, batch_size=1, num_layers=101)
        sw, scfg = reference_solver(variant, net, output_dir=tmp, loader=loader)
        sw.construct_graph()
        return net, sw
    ld = StubLoader()
    net, sw = fresh(ld)
    ref_sd = net.state_dict()
    src = {}
    for k, t in ref_sd.items():
        if k in sd:
            t.copy_(torch.from_numpy(sd[k])); src[k] = 'gen'
        elif t.numel() * t.element_size() <= 64:
            src[k] = 'inline'                                  # BatchNorm num_batches_tracked (int64 scalars)
        else:
            t.zero_(); src[k] = 'zeros'                        # resnet.fc.*: in the module, never used (RES:133)
    # host RNG streams in a known, non-initial state
    np.random.seed(1234); np.random.rand(5)
    random.seed(77); random.random()
    sfile, nfile = sw.snapshot(7)
    recs, small = _pth_records(sfile)
    # storage key of every tensor: torch.save numbers the storages in traversal order; confirmed by the payload CRCs
    prefix = recs[0][0].split('/')[0]
    by_name = {r[0]: r for r in recs}
    keys = []
    for i, (k, t) in enumerate(ref_sd.items()):
        r = by_name['%s/data/%d' % (prefix, i)]
        raw = t.detach().contiguous().numpy().tobytes()
        assert r[1] == len(raw) and r[2] == (zlib.crc32(raw) & 0xFFFFFFFF) and r[3] == 0, (k, r)
        e = dict(key=k, shape=list(t.shape), dtype=str(t.dtype), record=r[0], size=r[1], crc32=r[2], source=src[k])
        if src[k] == 'inline':
            e['bytes'] = base64.b64encode(raw).decode()
        keys.append(e)
    os.makedirs(SNAP_DIR, exist_ok=True)
    for name, data in small.items():
        with open(os.path.join(SNAP_DIR, 'pth.' + name.split('/', 1)[1].replace('/', '.')), 'wb') as f:
            f.write(data)
    shutil.copy(nfile, os.path.join(SNAP_DIR, os.path.basename(nfile)))
    man = dict(prefix=prefix, pth=os.path.basename(sfile), pkl=os.path.basename(nfile), iter=7, records=[dict(name=r[0], size=r[1], crc32=r[2]) for r in recs],
               keys=keys, meta=dict(H=H, W=W, T=T, V=V, seed_w=3, head_gain=4.0, variant=variant, torch=torch.__version__),
               written_by='pyutils/mask-faster-rcnn/lib/model/train_val_cycle.py SolverWrapper.snapshot (reference, run by tests/golden/make_golden.py snapshot)')

    # ---- what the reference's from_snapshot restores from its own pair ----
    def restore(spath):
        ld2 = StubLoader(seed=99); ld2.iterators = {'train': 0, 'val': 0}
        net2, sw2 = fresh(ld2)
        for k, t in net2.state_dict().items():               # current values that a partial copy must leave alone
            if t.dtype.is_floating_point:
                t.fill_(0.25)
        np.random.seed(1); random.seed(1)
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            last = sw2.from_snapshot(spath, nfile)
        out = dict(last_snapshot_iter=int(last), iter_train=int(ld2.iterators['train']), iter_val=int(ld2.iterators['val']),
                   perm_train=[int(x) for x in ld2.perm['train']], perm_val=[int(x) for x in ld2.perm['val']],
                   next_np_rand=[float(x) for x in np.random.rand(3)], next_py_random=random.random(),
                   printed=[l for l in buf.getvalue().splitlines() if l.startswith('size ')])
        return net2, out
    net2, full = restore(sfile)
    for k, t in net2.state_dict().items():
        assert torch.equal(t, ref_sd[k]), k
    man['restore_full'] = full
    # the same pair with rpn_net.weight saved one input channel short: TV:121-124 copies it into param[:, :-1] and keeps the rest
    sd_part = torch.load(sfile)
    sd_part['rpn_net.weight'] = sd_part['rpn_net.weight'][:, :-1].clone()
    pfile = os.path.join(tmp, 'partial.pth')
    torch.save(sd_part, pfile)
    net3, part = restore(pfile)
    w = net3.state_dict()['rpn_net.weight']
    assert torch.equal(w[:, :-1], ref_sd['rpn_net.weight'][:, :-1]) and bool((w[:, -1] == 0.25).all())
    part['rpn_net.weight.sum'] = float(w.double().sum()); part['rpn_net.weight.last_channel'] = 0.25
    man['restore_partial'] = part
    with open(os.path.join(SNAP_DIR, 'manifest.json'), 'w') as f:
        json.dump(man, f, indent=1)
    print('snapshot: %d records, %d keys (%s); restore_full %s; partial %s' % (
        len(recs), len(keys), {v: sum(1 for e in keys if e['source'] == v) for v in ('gen', 'zeros', 'inline')}, full['printed'], part['printed']))
    shutil.rmtree(tmp, ignore_errors=True)


def run_solver_tables():
    """the param-group table (key, lr, weight decay) of every variant's reference solver in BOTH branches of construct_graph(): the usual one
    (cfg.TRAIN.FROM_FRCN False) and the detector fine-tuning rule (True: lr x GAMMA for everything but the mask branch, train_val.py:175-185).
    No network is run: construct_graph() only builds the architecture and the optimiser.  -> tests/golden/ref_solver_tables.json"""
    import importlib
    import json
    out = {}
    for variant in ('baseline', 'spatial', 'response', 'cycle', 'cycle_response', 'vgg'):
        var = OW.VARIANTS[variant]
        RESM = importlib.import_module('nets.' + var['module'])
        opt = OW.default_opt(vocab_size=60, seq_length=6)
        if var.get('backbone') == 'vgg':
            opt['C4_feat_dim'] = 512
        for frcn in (False, True):
            torch.manual_seed(0)
            net = RESM.vgg16(opt, batch_size=1) if var.get('backbone') == 'vgg' else RESM.resnetv1(opt, batch_size=1, num_layers=101)
            sw, scfg = reference_solver(variant, net)
            was = scfg.TRAIN.FROM_FRCN
            scfg.TRAIN.FROM_FRCN = frcn
            try:
                import io, contextlib
                with contextlib.redirect_stdout(io.StringIO()):
                    lr0, optimizer = sw.construct_graph()
            finally:
                scfg.TRAIN.FROM_FRCN = was
            group_of = {id(g['params'][0]): g for g in optimizer.param_groups}
            rows = [(k, group_of[id(p)]['lr'], group_of[id(p)]['weight_decay']) for k, p in net.named_parameters() if p.requires_grad]
            out['%s/%s' % (variant, 'from_frcn' if frcn else 'default')] = dict(
                module=SOLVER_MODULE[variant], LEARNING_RATE=scfg.TRAIN.LEARNING_RATE, GAMMA=scfg.TRAIN.GAMMA, WEIGHT_DECAY=scfg.TRAIN.WEIGHT_DECAY,
                DOUBLE_BIAS=bool(scfg.TRAIN.DOUBLE_BIAS), BIAS_DECAY=bool(scfg.TRAIN.BIAS_DECAY), momentum=optimizer.defaults['momentum'],
                keys=[r[0] for r in rows], lr=[r[1] for r in rows], wd=[r[2] for r in rows])
            print(variant, frcn, len(rows), sorted(set(round(r[1], 10) for r in rows)))
    json.dump(out, open(os.path.join(HERE, 'ref_solver_tables.json'), 'w'))


def read_build_snapshot(src=None):
    """f3, reverse direction: the snapshot pair the BUILD wrote on the MI355X (tests/test_train_step_gpu.py::
    test_resume_from_reference_written_snapshot leaves its zip structure, sidecar and per-tensor CRCs in gpurun_out/build_snapshot/; its
    payloads are the synthetic weights it had just restored, regenerated here and checked against those CRCs) is loaded by the REFERENCE's
    own SolverWrapper.from_snapshot (train_val_cycle.py:106-165) into a fresh reference network.  The transcript is committed as
    tests/golden/build_snapshot_readback.json together with the build-written structure (tests/golden/build_snapshot/)."""
    import io
    import contextlib
    import json
    import random
    import shutil
    import tempfile
    import zipfile
    import zlib
    import importlib
    src = src or os.path.join(ROOT, 'gpurun_out', 'build_snapshot')
    man = json.load(open(os.path.join(src, 'manifest.json')))
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant='cycle')
    tmp = tempfile.mkdtemp(prefix='l2s_build_snapshot_')
    prefix = man['records'][0]['name'].split('/')[0]
    by_rec = {'%s/data/%d' % (prefix, i): e for i, e in enumerate(man['keys'])}
    sfile = os.path.join(tmp, man['pth'])
    with zipfile.ZipFile(sfile, 'w', zipfile.ZIP_STORED) as z:
        for r in man['records']:
            name = r['name']
            if name in by_rec:
                e = by_rec[name]
                dt = np.dtype(e['dtype'].replace('torch.', ''))
                raw = (np.ascontiguousarray(sd[e['key']], dtype=dt) if e['key'] in sd else np.zeros(e['shape'], dt)).tobytes()
                assert len(raw) == e['size'] and (zlib.crc32(raw) & 0xFFFFFFFF) == e['crc32'], ('payload differs from what the build wrote', e['key'])
            else:
                raw = open(os.path.join(src, 'pth.' + name.split('/', 1)[1].replace('/', '.')), 'rb').read()
            assert len(raw) == r['size'] and (zlib.crc32(raw) & 0xFFFFFFFF) == r['crc32'], name
            z.writestr(zipfile.ZipInfo(name), raw)
    nfile = os.path.join(src, man['pkl'])
    RESM = importlib.import_module('nets.' + OW.VARIANTS['cycle']['module'])
    torch.manual_seed(0)
    net = RESM.resnetv1(opt, batch_size=1, num_layers=101)
    ld = StubLoader(seed=99); ld.iterators = {'train': 0, 'val': 0}
    sw, scfg = reference_solver('cycle', net, output_dir=tmp, loader=ld)
    sw.construct_graph()
    for k, t in net.state_dict().items():
        if t.dtype.is_floating_point:
            t.fill_(0.25)
    np.random.seed(1); random.seed(1)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        last = sw.from_snapshot(sfile, nfile)
    got = net.state_dict()
    bad = [k for k, v in sd.items() if not np.array_equal(got[k].numpy(), v)]
    exp = man['expect']
    res = dict(reader='pyutils/mask-faster-rcnn/lib/model/train_val_cycle.py SolverWrapper.from_snapshot (reference)', written_by=man['written_by'],
               printed=[l for l in buf.getvalue().splitlines() if l.startswith('size ')], last_snapshot_iter=int(last),
               iter_train=int(ld.iterators['train']), iter_val=int(ld.iterators['val']),
               perm_train=[int(x) for x in ld.perm['train']], perm_val=[int(x) for x in ld.perm['val']],
               keys_in_file=len(man['keys']), keys_of_reference_net=len(got), tensors_equal_to_what_the_build_saved=len(sd) - len(bad), mismatching=bad,
               keys_the_file_lacks=sorted(set(got) - {e['key'] for e in man['keys']})[:4] + ['...'],
               ok=bool(not bad and int(last) == man['iter'] and ld.iterators['train'] == exp['iter_train'] and ld.iterators['val'] == exp['iter_val']
                       and [int(x) for x in ld.perm['train']] == exp['perm_train'] and [int(x) for x in ld.perm['val']] == exp['perm_val']))
    dst = os.path.join(HERE, 'build_snapshot')
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    json.dump(res, open(os.path.join(HERE, 'build_snapshot_readback.json'), 'w'), indent=1)
    print('read_build_snapshot:', {k: v for k, v in res.items() if k not in ('perm_train', 'perm_val')})
    shutil.rmtree(tmp, ignore_errors=True)


def run_eval_split_vgg():
    """the reference's boxes-only evaluation loop of the VGG16 network (model/test_vgg.py:185-460 eval_split, :97-131 im_detect) on the tiny synthetic
    split of run_eval_split: per sentence the chosen (RoI, class) and box, the box accuracy.  Only Network.test_image is replaced by the harness's
    bypass of forward() (as in run_reference_test)."""
    import importlib
    from model.config import cfg
    import model.test_vgg as MT
    from oracle.net import DEFAULT_CFG
    sys.modules['cv2'].resize = None
    H, W, T, V = 160, 224, 6, 60
    opt = OW.default_opt(vocab_size=V, seq_length=T); opt['C4_feat_dim'] = 512
    sd = eval_state_dict_vgg(opt)
    for k, v in DEFAULT_CFG['TEST'].items():
        setattr(cfg.TEST, k, v)
    cfg.TEST.MODE = 'nms'
    cfg.ANCHOR_SCALES = list(DEFAULT_CFG['ANCHOR_SCALES']); cfg.ANCHOR_RATIOS = list(DEFAULT_CFG['ANCHOR_RATIOS'])
    RESM = importlib.import_module('nets.' + OW.VARIANTS['vgg']['module'])
    torch.manual_seed(0)
    net = RESM.vgg16(opt, batch_size=1)
    net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    ref_sd = net.state_dict()
    for k, v in sd.items():
        ref_sd[k].copy_(torch.from_numpy(v))
    net.eval()

    def test_image(blobs):                      # network_vgg.py:701-718 with forward() bypassed (mode TEST)
        net._image = torch.from_numpy(np.ascontiguousarray(blobs['data'].transpose([0, 3, 1, 2])))
        net._im_info = blobs['im_info']
        net._gt_boxes = torch.from_numpy(blobs['gt_boxes'])
        net._gt_masks = blobs['gt_masks']
        net._labels = torch.from_numpy(np.asarray(blobs['labels'].a))
        net._cap_labels = None; net._cap_masks = None
        net._mode = 'TEST'
        net._image_gt_summaries = {}
        with torch.no_grad():
            pr = net._predict()
            net_conv, rois, cls_prob, bbox_pred = pr[:4]
            stds = bbox_pred.data.new(cfg.TRAIN.BBOX_NORMALIZE_STDS).repeat(net._num_classes).unsqueeze(0).expand_as(bbox_pred)
            means = bbox_pred.data.new(cfg.TRAIN.BBOX_NORMALIZE_MEANS).repeat(net._num_classes).unsqueeze(0).expand_as(bbox_pred)
            bbox_pred = bbox_pred.mul(stds).add(means)
        return (net._predictions['cls_score'].numpy(), cls_prob.numpy(), bbox_pred.numpy(), rois.numpy(), net_conv)
    net.test_image = test_image
    imgs = eval_blobs(H=H, W=W, T=T, V=V)

    class Loader(object):
        def __init__(self):
            self.i = 0

        def getTestBatch(self, split):
            b = dict(imgs[self.i]); self.i += 1
            b['labels'] = _Labels(b['labels'])
            b['bounds'] = dict(it_pos_now=self.i, it_max=len(imgs), wrapped=self.i >= len(imgs))
            return b
    picked = []
    orig_detect = MT.im_detect

    def rec_detect(model, blobs):
        r = orig_detect(model, blobs)
        sc, bx = r[0], r[1]
        pr = np.where(sc == np.max(sc[:, 1:]))
        picked.append((int(pr[0][0]), int(pr[1][0]), bx[pr[0][0], pr[1][0] * 4:(pr[1][0] + 1) * 4].copy()))
        return r
    MT.im_detect = rec_detect
    with torch.no_grad():
        acc, num_sent = MT.eval_split(Loader(), net, None, 'val', dict(verbose=False))
    MT.im_detect = orig_detect
    out = dict(acc=float(acc), num_sent=int(num_sent), pred_roi=np.array([p[0] for p in picked]), pred_class=np.array([p[1] for p in picked]),
               pred_box=np.stack([p[2] for p in picked]).astype(np.float32))
    np.savez_compressed(os.path.join(HERE, 'ref_eval_split_vgg.npz'), **out)
    print('eval_split_vgg: acc %.3f sents %d' % (acc, num_sent), out['pred_class'], out['pred_box'])


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    install_harness()
    if what in ('leaf', 'all'):
        run_leaf()
    if what in ('leaf_eval', 'all'):
        run_leaf_eval()
    if what in ('eval_split', 'all'):
        run_eval_split()
    if what in ('eval_split_vgg', 'all'):
        run_eval_split_vgg()
    if what in ('tiny', 'all'):
        hook_proposals()
        run_reference('tiny', 320, 416, 6, 60, dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300,
                                                    RPN_BATCHSIZE=64), head_gain=float(os.environ.get('HG', '4')))
    if what in ('variants', 'all'):
        # the other ResNet network variants of the reference (BASELINE.json configs 0, 1, 3 + train_response.sh), tiny size
        for v in (sys.argv[2:] or ['baseline', 'spatial', 'response', 'cycle_response', 'vgg']):
            hook_proposals(v)
            run_reference('tiny_' + v, 320, 416, 6, 60, dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300,
                                                             RPN_BATCHSIZE=64), head_gain=4.0, variant=v)
    if what in ('align', 'all'):
        # cfg.POOLING_ALIGN: _crop_pool_layer_align (NET:151-182) = image-space affine grid, 14x14 crop + 2x2 max pool
        hook_proposals()
        run_reference('tiny_align', 320, 416, 6, 60, dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64),
                      head_gain=4.0, top_over=dict(POOLING_ALIGN=True))
    if what in ('snapshot', 'all'):
        run_snapshot()
    if what in ('solver_tables', 'all'):
        run_solver_tables()
    if what == 'read_build_snapshot':
        read_build_snapshot(sys.argv[2] if len(sys.argv) > 2 else None)
    if what in ('fb0', 'all'):
        # cfg.RESNET.FIXED_BLOCKS = 0 (RES:290-299): layer1 trains too - its weight gradients and its part of the update are the last of the step
        hook_proposals()
        run_reference('tiny_fb0', 320, 416, 6, 60, dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64),
                      head_gain=4.0, resnet_over=dict(FIXED_BLOCKS=0))
    if what in ('test', 'all'):
        run_reference_test('test_tiny', 320, 416, 6, 60)
        run_reference_test('test_tiny_cycle_response', 320, 416, 6, 60, variant='cycle_response')
        for v in ('baseline', 'spatial', 'response'):         # (round 6: TEST mode of the remaining ResNet variants)
            run_reference_test('test_tiny_' + v, 320, 416, 6, 60, variant=v)
    if what in ('test_vgg', 'all'):
        run_reference_test('test_tiny_vgg', 320, 416, 6, 60, variant='vgg')
    if what in ('test_top', 'all'):
        run_reference_test('test_tiny_top', 320, 416, 6, 60, test_mode='top', top_n=200)
    if what in ('full', 'all'):
        hook_proposals()
        run_reference('full', 600, 1000, 20, 3349, dict(BATCH_SIZE=256, RPN_PRE_NMS_TOP_N=12000,
                                                         RPN_POST_NMS_TOP_N=2000, RPN_BATCHSIZE=256))
    if what == 'full_variants':
        # BASELINE.json configs 2, 4, 5 (and 1) at their stated size (round 6: also `full_variants baseline response`): 600x1000, 12000 -> 2000 proposals, 256 RoIs; expression length / vocabulary
        # of the config's dataset (refcoco(+) unc: 10 tokens, V = 1999; refcocog umd: 20 tokens, V = 3349).  Not part of 'all': minutes each.
        TV = dict(baseline=(10, 1999), spatial=(10, 1999), response=(10, 1999), cycle_response=(20, 3349), vgg=(10, 1999))
        for v in (sys.argv[2:] or ['spatial', 'cycle_response', 'vgg']):
            hook_proposals(v)
            run_reference('full_' + v, 600, 1000, TV[v][0], TV[v][1], dict(BATCH_SIZE=256, RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000,
                                                                          RPN_BATCHSIZE=256), variant=v)
