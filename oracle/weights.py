"""Deterministic synthetic weights with the reference's state_dict key names and
OIHW shapes (test oracle + fixtures + bench share this generator, so no weight
file ever needs to be committed).  Shapes follow
pyutils/mask-faster-rcnn/lib/nets/resnet_v1_cycle_res5_2.py:117-133,275-335 (RES),
lib/layers/lang_encoder.py:11-25 and lib/caption_models/AttModel.py:27-57,426-443.
Initial scales follow RES:135-141 / network_cycle_res5_2.py:333-355 where the
reference defines them; BN running stats/affine are drawn non-trivially so the
frozen-BN folding is exercised."""
import numpy as np

RESNET_LAYERS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}

# The reference's ResNet network variants (one nets/network_*.py + nets/resnet_v1_*.py pair each):
#   nfilt: 1 = dynamic_fc (network.py:475-479), 7 = dynamic_fc_0..6 + response_fc (network_7f.py:475-534)
#   gate : 'linear'  net_conv * response, 'sigmoid' net_conv * sigmoid(response) + response BCE loss
#          (network_7f_response.py:411-419,543-545; network_cycle_response.py:415-423,568-570)
#   cap  : None, 'mask' (network_cycle_res5_2.py:415-440), 'before_after' (network_cycle_response.py:425-439)
VARIANTS = {
    'baseline': dict(nfilt=1, gate='linear', cap=None, module='resnet_v1', net='network'),
    'spatial': dict(nfilt=7, gate='linear', cap=None, module='resnet_v1_7f', net='network_7f'),
    'response': dict(nfilt=7, gate='sigmoid', cap=None, module='resnet_v1_7f_response', net='network_7f_response'),
    'cycle': dict(nfilt=7, gate='linear', cap='mask', module='resnet_v1_cycle_res5_2', net='network_cycle_res5_2'),
    'cycle_response': dict(nfilt=7, gate='sigmoid', cap='before_after', module='resnet_v1_cycle_response', net='network_cycle_response'),
    # VGG16 / Faster R-CNN variant (nets/vgg16.py + nets/network_vgg.py, train_vgg.sh): conv5_3 map (512 ch), 14x14 crop +
    # 2x2 max pool, fc6/fc7, no mask branch, sigmoid gating + response loss
    'vgg': dict(nfilt=7, gate='sigmoid', cap=None, module='vgg16', net='network_vgg', backbone='vgg', mask=False),
}
# The solver each variant is trained with (tools/train*.py:22-24 import one model/train_val*.py each; construct_graph there):
#   lang_lr_mult: learning-rate factor of the parameters whose key contains 'rnn_encoder', 'dynamic_fc' or 'response'
#                 (train_val.py:193-198, train_val_response.py:193-198, train_val_vgg.py:193-198: x10; the two cycle solvers have the
#                 rule commented out, train_val_cycle.py:199-204, train_val_cycle_response.py:193-198)
#   cfg         : overrides of model/config.py's TRAIN defaults in the config module the solver imports (train_val_vgg.py:12 ->
#                 model/config_vgg.py:28,40: WEIGHT_DECAY 5e-4, DOUBLE_BIAS True)
SOLVERS = {
    'baseline': dict(module='train_val', lang_lr_mult=10.0, cfg={}),
    'spatial': dict(module='train_val', lang_lr_mult=10.0, cfg={}),
    'response': dict(module='train_val_response', lang_lr_mult=10.0, cfg={}),
    'cycle': dict(module='train_val_cycle', lang_lr_mult=1.0, cfg={}),
    'cycle_response': dict(module='train_val_cycle_response', lang_lr_mult=1.0, cfg={}),
    'vgg': dict(module='train_val_vgg', lang_lr_mult=10.0, cfg=dict(WEIGHT_DECAY=5e-4, DOUBLE_BIAS=True)),
}
LANG_KEYS = ('rnn_encoder', 'dynamic_fc', 'response')


def param_group(variant, key, train_cfg):
    """(lr factor, weight decay) of one parameter as the variant's construct_graph() sets them (FROM_FRCN False)."""
    sv = SOLVERS[variant]
    ct = dict(train_cfg, **sv['cfg'])
    is_bias = 'bias' in key
    if ct.get('FROM_FRCN'):                       # train_val.py:175-185: the mask branch at lr, everything else at lr x GAMMA
        mult = 1.0 if 'mask' in key else ct['GAMMA']
    else:
        mult = sv['lang_lr_mult'] if any(t in key for t in LANG_KEYS) else 1.0
    if is_bias:
        return mult * ((2.0 if ct['DOUBLE_BIAS'] else 1.0)), (ct['WEIGHT_DECAY'] if ct['BIAS_DECAY'] else 0.0)
    return mult, ct['WEIGHT_DECAY']


VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]   # torchvision cfg 'D' minus the last pool
LOSS_KEYS = ['rpn_cross_entropy', 'rpn_loss_box', 'cross_entropy', 'loss_box', 'loss_mask', 'loss_response', 'loss_caption', 'total_loss']


def loss_keys(variant):
    """order of the floats Network.train_step returns (NET:702-719 and its variants)."""
    v = VARIANTS[variant]
    return [k for k in LOSS_KEYS if not (k == 'loss_response' and v['gate'] != 'sigmoid') and not (k == 'loss_caption' and v['cap'] is None)
            and not (k == 'loss_mask' and not v.get('mask', True))]


def vgg_feature_indices():
    """[(index in vgg.features, Cin, Cout)] of the 13 convolutions and the indices of the 4 max-pools that are kept."""
    convs, pools, i, cin = [], [], 0, 3
    for v in VGG_CFG:
        if v == 'M':
            pools.append(i); i += 1
        else:
            convs.append((i, cin, v)); cin = v; i += 2
    return convs, pools


def default_opt(vocab_size=1999, seq_length=10, cap_loss_weight=1.0):
    """tools/opt_cycle_2.py:4-128 effective defaults."""
    return dict(vocab_size=vocab_size, word_embedding_size=512, word_vec_size=512,
                rnn_hidden_size=512, bidirectional=1, word_drop_out=0.5, rnn_drop_out=0.2,
                rnn_num_layers=1, rnn_type='lstm', variable_lengths=1, C4_feat_dim=1024,
                cap_loss_weight=cap_loss_weight, caption_model='att2in2',
                input_encoding_size=512, rnn_size=512, num_layers=1, drop_prob_lm=0.5,
                seq_length=seq_length, fc_feat_size=4096, att_feat_size=4096,
                att_hid_size=512, start_from=None, dataset_splitBy='refcoco_unc')


def param_shapes(opt, num_layers=101, num_classes=81, num_anchors=12, variant='cycle'):
    """Ordered {name: shape} for every tensor the hot path reads."""
    s = {}
    var = VARIANTS[variant]
    V = opt['vocab_size']; E = opt['word_embedding_size']; WV = opt['word_vec_size']
    Hh = opt['rnn_hidden_size']
    s['rnn_encoder.embedding.weight'] = (V, E)
    s['rnn_encoder.mlp.0.weight'] = (WV, E); s['rnn_encoder.mlp.0.bias'] = (WV,)
    for sfx in ['', '_reverse']:
        s['rnn_encoder.rnn.weight_ih_l0' + sfx] = (4 * Hh, WV)
        s['rnn_encoder.rnn.weight_hh_l0' + sfx] = (4 * Hh, Hh)
        s['rnn_encoder.rnn.bias_ih_l0' + sfx] = (4 * Hh,)
        s['rnn_encoder.rnn.bias_hh_l0' + sfx] = (4 * Hh,)
    R = opt['rnn_size']; IE = opt['input_encoding_size']; AH = opt['att_hid_size']
    cap_keys_from = len(s)
    s['caption_model.embed.0.weight'] = (V + 1, IE)
    s['caption_model.att_embed.0.weight'] = (R, opt['att_feat_size']); s['caption_model.att_embed.0.bias'] = (R,)
    s['caption_model.logit.weight'] = (V + 1, R); s['caption_model.logit.bias'] = (V + 1,)
    s['caption_model.ctx2att.weight'] = (AH, R); s['caption_model.ctx2att.bias'] = (AH,)
    s['caption_model.core.a2c.weight'] = (2 * R, R); s['caption_model.core.a2c.bias'] = (2 * R,)
    s['caption_model.core.i2h.weight'] = (5 * R, IE); s['caption_model.core.i2h.bias'] = (5 * R,)
    s['caption_model.core.h2h.weight'] = (5 * R, R); s['caption_model.core.h2h.bias'] = (5 * R,)
    s['caption_model.core.attention.h2att.weight'] = (AH, R); s['caption_model.core.attention.h2att.bias'] = (AH,)
    s['caption_model.core.attention.alpha_net.weight'] = (1, AH); s['caption_model.core.attention.alpha_net.bias'] = (1,)
    if var['cap'] is None:
        for k in list(s.keys())[cap_keys_from:]:
            del s[k]

    def bn(p, c):
        for k in ['weight', 'bias', 'running_mean', 'running_var']:
            s[p + '.' + k] = (c,)
    if var.get('backbone') == 'vgg':
        return _vgg_shapes(s, opt, num_classes, num_anchors)
    s['resnet.conv1.weight'] = (64, 3, 7, 7); bn('resnet.bn1', 64)
    inpl = 64
    for li, (planes, nb) in enumerate(zip([64, 128, 256, 512], RESNET_LAYERS[num_layers]), 1):
        for b in range(nb):
            p = 'resnet.layer%d.%d' % (li, b)
            s[p + '.conv1.weight'] = (planes, inpl, 1, 1); bn(p + '.bn1', planes)
            s[p + '.conv2.weight'] = (planes, planes, 3, 3); bn(p + '.bn2', planes)
            s[p + '.conv3.weight'] = (planes * 4, planes, 1, 1); bn(p + '.bn3', planes * 4)
            if b == 0:
                s[p + '.downsample.0.weight'] = (planes * 4, inpl, 1, 1); bn(p + '.downsample.1', planes * 4)
            inpl = planes * 4
    C4 = opt['C4_feat_dim']; HD = opt['rnn_num_layers'] * (2 if opt['bidirectional'] else 1) * Hh
    if var['nfilt'] == 1:
        s['dynamic_fc.weight'] = (C4, HD); s['dynamic_fc.bias'] = (C4,)
    else:
        for k in range(7):
            s['dynamic_fc_%d.weight' % k] = (C4, HD); s['dynamic_fc_%d.bias' % k] = (C4,)
        s['response_fc.weight'] = (7, HD); s['response_fc.bias'] = (7,)
    s['rpn_net.weight'] = (512, C4, 3, 3); s['rpn_net.bias'] = (512,)
    s['rpn_cls_score_net.weight'] = (2 * num_anchors, 512, 1, 1); s['rpn_cls_score_net.bias'] = (2 * num_anchors,)
    s['rpn_bbox_pred_net.weight'] = (4 * num_anchors, 512, 1, 1); s['rpn_bbox_pred_net.bias'] = (4 * num_anchors,)
    s['cls_score_net.weight'] = (num_classes, 2048); s['cls_score_net.bias'] = (num_classes,)
    s['bbox_pred_net.weight'] = (4 * num_classes, 2048); s['bbox_pred_net.bias'] = (4 * num_classes,)
    s['mask_up_sampling.weight'] = (2048, 256, 2, 2); s['mask_up_sampling.bias'] = (256,)
    s['mask_pred_net.weight'] = (num_classes, 256, 1, 1); s['mask_pred_net.bias'] = (num_classes,)
    return s


def _vgg_shapes(s, opt, num_classes, num_anchors):
    convs, _ = vgg_feature_indices()
    for i, cin, cout in convs:
        s['vgg.features.%d.weight' % i] = (cout, cin, 3, 3); s['vgg.features.%d.bias' % i] = (cout,)
    s['vgg.classifier.0.weight'] = (4096, 512 * 7 * 7); s['vgg.classifier.0.bias'] = (4096,)
    s['vgg.classifier.3.weight'] = (4096, 4096); s['vgg.classifier.3.bias'] = (4096,)
    C4 = opt['C4_feat_dim']; HD = opt['rnn_num_layers'] * (2 if opt['bidirectional'] else 1) * opt['rnn_hidden_size']
    for k in range(7):
        s['dynamic_fc_%d.weight' % k] = (C4, HD); s['dynamic_fc_%d.bias' % k] = (C4,)
    s['response_fc.weight'] = (7, HD); s['response_fc.bias'] = (7,)
    s['rpn_net.weight'] = (512, C4, 3, 3); s['rpn_net.bias'] = (512,)
    s['rpn_cls_score_net.weight'] = (2 * num_anchors, 512, 1, 1); s['rpn_cls_score_net.bias'] = (2 * num_anchors,)
    s['rpn_bbox_pred_net.weight'] = (4 * num_anchors, 512, 1, 1); s['rpn_bbox_pred_net.bias'] = (4 * num_anchors,)
    s['cls_score_net.weight'] = (num_classes, 4096); s['cls_score_net.bias'] = (num_classes,)
    s['bbox_pred_net.weight'] = (4 * num_classes, 4096); s['bbox_pred_net.bias'] = (4 * num_classes,)
    return s


def make_state_dict(opt, seed=3, num_layers=101, num_classes=81, num_anchors=12, head_gain=1.0, variant='cycle'):
    """name -> float32 ndarray.  One RandomState stream consumed in param_shapes order.
    `head_gain` > 1 scales the N(0,0.01) head initialisers so scores/deltas are not
    degenerate in parity fixtures (the reference init makes every RPN score ~0.5)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, shp in param_shapes(opt, num_layers, num_classes, num_anchors, variant).items():
        n = int(np.prod(shp))
        if name.endswith('running_mean'):
            a = rs.normal(0, 0.1, n)
        elif name.endswith('running_var'):
            a = rs.uniform(0.5, 1.5, n)
        elif '.bn' in name or 'downsample.1' in name:
            a = rs.uniform(0.8, 1.2, n) if name.endswith('weight') else rs.normal(0, 0.1, n)
            if name.endswith('bn3.weight') or name.endswith('downsample.1.weight'):
                a = a * (0.15 if name.endswith('bn3.weight') else 0.7)   # keep the residual stream's variance flat through 33 blocks
        elif name.startswith('vgg.features.') and len(shp) == 4:
            a = rs.normal(0, np.sqrt(2.0 / (shp[1] * 9)), n)                            # He fan-in keeps 13 ReLU layers at O(1)
            if name == 'vgg.features.0.weight':
                a = a * 0.02                                                           # inputs are pixel-scale (sigma 50)
        elif name.startswith('vgg.features.') and name.endswith('bias'):
            a = rs.normal(0, 0.05, n)
        elif name.startswith('vgg.classifier.') and name.endswith('weight'):
            a = rs.normal(0, np.sqrt(2.0 / shp[1]), n)
        elif name.startswith('resnet.') and len(shp) == 4:
            a = rs.normal(0, np.sqrt(2.0 / (shp[2] * shp[3] * shp[0])), n)            # RES:137-138
            if name == 'resnet.conv1.weight':
                a = a * 0.05                                                           # inputs are pixel-scale (sigma 50)
        elif name.startswith(('rpn_', 'cls_score', 'mask_')) and name.endswith('weight'):
            a = rs.normal(0, 0.01 * head_gain, n)
        elif name.startswith('bbox_pred_net') and name.endswith('weight'):
            a = rs.normal(0, 0.001 * head_gain, n)
        elif name.startswith(('rpn_', 'cls_score', 'mask_', 'bbox_pred')) and name.endswith('bias'):
            a = rs.normal(0, 0.01, n) if head_gain != 1.0 else np.zeros(n)
        elif 'embedding.weight' in name or 'embed.0.weight' in name:
            a = rs.normal(0, 1.0, n)
        else:   # nn.Linear / nn.LSTM default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            fan_in = shp[-1] if len(shp) > 1 else 512
            k = 1.0 / np.sqrt(fan_in)
            a = rs.uniform(-k, k, n)
        sd[name] = a.astype(np.float32).reshape(shp)
    return sd
