"""torch-CPU fp32 restatement of the lang2seg cycle network train step (test oracle).

Citations (relative to /root/reference):
  NET = pyutils/mask-faster-rcnn/lib/nets/network_cycle_res5_2.py
  RES = pyutils/mask-faster-rcnn/lib/nets/resnet_v1_cycle_res5_2.py
  TV  = pyutils/mask-faster-rcnn/lib/model/train_val_cycle.py
  ENC = lib/layers/lang_encoder.py     ATT = lib/caption_models/AttModel.py
  CRIT = lib/misc/utils.py
Semantics restored to the reference's torch-0.3 behaviour: grid_sample/affine_grid
with align_corners=True (NET:142-147), ByteTensor add at proposal_target_layer.py:146.
Dropout is expressed through explicit masks (already scaled by 1/(1-p)); None = off.
"""
import numpy as np
import torch
import torch.nn.functional as F
from . import boxes as B

DEFAULT_CFG = dict(
    TRAIN=dict(LEARNING_RATE=1e-4, MOMENTUM=0.9, WEIGHT_DECAY=1e-4, GAMMA=0.1, DOUBLE_BIAS=False,
               BIAS_DECAY=False, BATCH_SIZE=256, FG_FRACTION=0.25, FG_THRESH=0.5, BG_THRESH_HI=0.5,
               BG_THRESH_LO=0.0, BBOX_NORMALIZE_MEANS=(0.0, 0.0, 0.0, 0.0),
               BBOX_NORMALIZE_STDS=(0.1, 0.1, 0.2, 0.2), BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0),
               RPN_POSITIVE_OVERLAP=0.7, RPN_NEGATIVE_OVERLAP=0.3, RPN_FG_FRACTION=0.5,
               RPN_BATCHSIZE=256, RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000),
    TEST=dict(RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300),
    ANCHOR_SCALES=(4, 8, 16, 32), ANCHOR_RATIOS=(0.5, 1, 2), POOLING_SIZE=7, MASK_SIZE=14,
    FIXED_BLOCKS=1, NMS_CMP='ge')


def spatial_masks(H, W):
    """NET:530-557 (py2 true division + int()): 7 x (H,W) {0,1} masks."""
    m = np.zeros((7, H, W), np.float32)
    m[0] = 1
    m[1, :int(H / 2), :] = 1
    m[2, int(H / 2):, :] = 1
    m[3, :, :int(W / 2)] = 1
    m[4, :, int(W / 2):] = 1
    m[5, int(H / 4):int(H * 3 / 4), :] = 1
    m[6, :, int(W / 4):int(W * 3 / 4)] = 1
    return m


class OracleNet(object):
    def __init__(self, sd, opt, cfg=None, num_layers=101, num_classes=81, variant='cycle'):
        from .weights import VARIANTS
        self.variant = variant; self.var = VARIANTS[variant]
        self.cfg = DEFAULT_CFG if cfg is None else cfg
        self.opt = opt
        self.num_classes = num_classes
        self.nblocks = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[num_layers]
        self.A = len(self.cfg['ANCHOR_SCALES']) * len(self.cfg['ANCHOR_RATIOS'])
        self.p = {k: torch.from_numpy(np.ascontiguousarray(v)).clone() for k, v in sd.items()}
        self.trainable = []
        for k, t in self.p.items():
            if self._is_trainable(k):
                t.requires_grad_(True)
                self.trainable.append(k)
        self.momentum = {k: torch.zeros_like(self.p[k]) for k in self.trainable}

    def _is_trainable(self, k):
        """RES:290-306: conv1/bn1/layer1 (FIXED_BLOCKS=1) and every BN tensor are frozen; vgg16.py:49-51: features[0..9]."""
        if k.startswith('vgg.features.'):
            return int(k.split('.')[2]) >= 10
        if k.startswith('resnet.'):
            if '.bn' in k or 'downsample.1' in k or k.startswith('resnet.bn1') or k.startswith('resnet.conv1'):
                return False
            for fb in range(1, self.cfg['FIXED_BLOCKS'] + 1):
                if k.startswith('resnet.layer%d.' % fb):
                    return False
        return True

    # ---- backbone -------------------------------------------------------
    def _bn(self, x, p):
        return F.batch_norm(x, self.p[p + '.running_mean'], self.p[p + '.running_var'],
                            self.p[p + '.weight'], self.p[p + '.bias'], False, 0.0, 1e-5)

    def _bottleneck(self, x, p, stride):
        """RES:94-114, stride on the first 1x1 (RES:83)."""
        o = F.relu(self._bn(F.conv2d(x, self.p[p + '.conv1.weight'], stride=stride), p + '.bn1'))
        o = F.relu(self._bn(F.conv2d(o, self.p[p + '.conv2.weight'], padding=1), p + '.bn2'))
        o = self._bn(F.conv2d(o, self.p[p + '.conv3.weight']), p + '.bn3')
        if (p + '.downsample.0.weight') in self.p:
            x = self._bn(F.conv2d(x, self.p[p + '.downsample.0.weight'], stride=stride), p + '.downsample.1')
        return F.relu(o + x)

    def _layer(self, x, li, stride):
        for b in range(self.nblocks[li - 1]):
            x = self._bottleneck(x, 'resnet.layer%d.%d' % (li, b), stride if b == 0 else 1)
        return x

    def image_to_head(self, image_nchw):
        """RES:261-265,309-310; VGG: vgg16.py:53-54,78-82 (features without the last max-pool)."""
        if self.var.get('backbone') == 'vgg':
            from .weights import vgg_feature_indices
            convs, pools = vgg_feature_indices()
            x = image_nchw
            ci = {i: None for i, _, _ in convs}
            for i in range(30):
                if i in ci:
                    x = F.relu(F.conv2d(x, self.p['vgg.features.%d.weight' % i], self.p['vgg.features.%d.bias' % i], padding=1))
                elif i in pools:
                    x = F.max_pool2d(x, 2, 2)
            return x
        x = F.relu(self._bn(F.conv2d(image_nchw, self.p['resnet.conv1.weight'], stride=2, padding=3), 'resnet.bn1'))
        x = F.max_pool2d(x, 3, 2, 1)
        self.t_stem = x
        x = self._layer(x, 1, 1)
        self.t_layer1 = x
        x = self._layer(x, 2, 2)
        self.t_layer2 = x
        return self._layer(x, 3, 2)

    def head_to_tail(self, x, drops=None):
        if self.var.get('backbone') == 'vgg':                     # vgg16.py:84-88: fc6 / fc7 (+ReLU, dropout)
            h = F.relu(F.linear(x.reshape(x.shape[0], -1), self.p['vgg.classifier.0.weight'], self.p['vgg.classifier.0.bias']))
            if drops is not None and drops.get('fc6') is not None:
                h = h * drops['fc6']
            h = F.relu(F.linear(h, self.p['vgg.classifier.3.weight'], self.p['vgg.classifier.3.bias']))
            if drops is not None and drops.get('fc7') is not None:
                h = h * drops['fc7']
            return h
        return self._layer(x, 4, 1)       # RES:271-273, layer4 stride 1 (RES:131)

    # ---- language encoder (ENC:27-82), batch 1 ----------------------------
    @staticmethod
    def trim_labels(labels):
        """forward(), NET:629-630: `max_len = (labels != 0).sum(1).max(); labels = labels[:, :max_len]` (zero padding at the row end)"""
        labels = np.asarray(labels)
        return np.ascontiguousarray(labels[:, :int((labels != 0).sum(1).max())])

    def rnn_encoder(self, labels, word_drop=None):
        emb = self.p['rnn_encoder.embedding.weight'][labels[0]]          # (T, E)
        if word_drop is not None:
            emb = emb * word_drop
        x = F.relu(F.linear(emb, self.p['rnn_encoder.mlp.0.weight'], self.p['rnn_encoder.mlp.0.bias']))
        Hh = self.opt['rnn_hidden_size']

        def run(sfx, seq):
            h = torch.zeros(Hh); c = torch.zeros(Hh)
            wi = self.p['rnn_encoder.rnn.weight_ih_l0' + sfx]; wh = self.p['rnn_encoder.rnn.weight_hh_l0' + sfx]
            bi = self.p['rnn_encoder.rnn.bias_ih_l0' + sfx]; bh = self.p['rnn_encoder.rnn.bias_hh_l0' + sfx]
            for t in seq:
                g = F.linear(x[t], wi, bi) + F.linear(h, wh, bh)       # gate order i,f,g,o
                i, f, gg, o = g[:Hh], g[Hh:2 * Hh], g[2 * Hh:3 * Hh], g[3 * Hh:]
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
            return h
        T = x.shape[0]
        hf = run('', range(T)); hb = run('_reverse', range(T - 1, -1, -1))
        return torch.cat((hf, hb)).unsqueeze(0)                           # ENC:76-80

    # ---- dynamic filters (NET:504-562) ------------------------------------
    def dynamic_filter(self, net_conv, hidden):
        H, W = net_conv.shape[2], net_conv.shape[3]
        if self.var['nfilt'] == 1:                                          # network.py:475-479
            f = torch.tanh(F.linear(hidden, self.p['dynamic_fc.weight'], self.p['dynamic_fc.bias']))
            response = F.conv2d(net_conv, f.view(1, -1, 1, 1))
        else:
            masks = torch.from_numpy(spatial_masks(H, W))
            resp = []
            for k in range(7):
                f = torch.tanh(F.linear(hidden, self.p['dynamic_fc_%d.weight' % k], self.p['dynamic_fc_%d.bias' % k]))
                resp.append(F.conv2d(net_conv * masks[k][None, None], f.view(1, -1, 1, 1)))
            r = torch.tanh(F.linear(hidden, self.p['response_fc.weight'], self.p['response_fc.bias']))
            response = F.conv2d(torch.cat(resp, 1), r.view(1, 7, 1, 1))
        self.t_response = response
        if self.var['gate'] == 'sigmoid':                                   # network_7f_response.py:543-545
            return net_conv * torch.sigmoid(response)
        return net_conv * response

    # ---- crop pool (NET:107-149, max_pool False via RES:258-259) -----------
    def crop_pool(self, bottom, rois, size=None, im_info=None):
        """NET:107-149 (_crop_pool_layer) and, with cfg POOLING_ALIGN, NET:151-182 (_crop_pool_layer_align: theta from the RoI in image
        pixels over im_info's size, always followed by the 2x2 max pool).  cfg RESNET_MAX_POOL = RES:252-253."""
        size = size or self.cfg['POOLING_SIZE']
        rois = rois.detach()
        align = bool(self.cfg.get('POOLING_ALIGN', False))
        if align:
            im_info = self._im_info if im_info is None else im_info
            x1 = rois[:, 1:2]; y1 = rois[:, 2:3]; x2 = rois[:, 3:4]; y2 = rois[:, 4:5]
            height, width = float(im_info[0][0]), float(im_info[0][1])
        else:
            x1 = rois[:, 1:2] / 16.0; y1 = rois[:, 2:3] / 16.0
            x2 = rois[:, 3:4] / 16.0; y2 = rois[:, 4:5] / 16.0
            height, width = bottom.shape[2], bottom.shape[3]
        zero = torch.zeros(rois.shape[0], 1)
        theta = torch.cat([(x2 - x1) / (width - 1), zero, (x1 + x2 - width + 1) / (width - 1),
                           zero, (y2 - y1) / (height - 1), (y1 + y2 - height + 1) / (height - 1)], 1).view(-1, 2, 3)
        # network_vgg.py:139-143: 14x14 crop + 2x2 max pool
        if align or self.var.get('backbone') == 'vgg' or self.cfg.get('RESNET_MAX_POOL', False):
            grid = F.affine_grid(theta, (rois.shape[0], 1, 2 * size, 2 * size), align_corners=True)
            return F.max_pool2d(F.grid_sample(bottom.expand(rois.shape[0], -1, -1, -1), grid, align_corners=True), 2, 2)
        grid = F.affine_grid(theta, (rois.shape[0], 1, size, size), align_corners=True)
        return F.grid_sample(bottom.expand(rois.shape[0], -1, -1, -1), grid, align_corners=True)

    # ---- att2in2 captioner (ATT:60-101, 406-466) ---------------------------
    def caption(self, att_feats, seq, drops=None):
        """att_feats (1,196,4096); seq (1,L+2) int64. drops: dict of masks or None.
        Returns log-probs (1, steps, V+1)."""
        R = self.opt['rnn_size']
        P = self.p
        a = F.relu(F.linear(att_feats.view(-1, att_feats.shape[-1]), P['caption_model.att_embed.0.weight'],
                            P['caption_model.att_embed.0.bias']))
        if drops is not None and drops.get('att') is not None:
            a = a * drops['att']
        pa = F.linear(a, P['caption_model.ctx2att.weight'], P['caption_model.ctx2att.bias'])
        h = torch.zeros(1, R); c = torch.zeros(1, R)
        outs = []
        for i in range(seq.shape[1] - 1):
            if i >= 1 and int(seq[:, i].sum()) == 0:                      # ATT:92-93
                break
            xt = F.relu(P['caption_model.embed.0.weight'][seq[:, i]])
            if drops is not None and drops.get('xt') is not None:
                xt = xt * drops['xt'][i]
            att_h = F.linear(h, P['caption_model.core.attention.h2att.weight'], P['caption_model.core.attention.h2att.bias'])
            dot = torch.tanh(pa + att_h)
            dot = F.linear(dot, P['caption_model.core.attention.alpha_net.weight'], P['caption_model.core.attention.alpha_net.bias']).view(1, -1)
            wgt = F.softmax(dot, 1)
            att_res = wgt @ a                                               # (1,R)
            s = F.linear(xt, P['caption_model.core.i2h.weight'], P['caption_model.core.i2h.bias']) + \
                F.linear(h, P['caption_model.core.h2h.weight'], P['caption_model.core.h2h.bias'])
            sg = torch.sigmoid(s[:, :3 * R])
            ig, fg, og = sg[:, :R], sg[:, R:2 * R], sg[:, 2 * R:3 * R]
            it = s[:, 3 * R:] + F.linear(att_res, P['caption_model.core.a2c.weight'], P['caption_model.core.a2c.bias'])
            it = torch.max(it[:, :R], it[:, R:])
            c = fg * c + ig * it
            h = og * torch.tanh(c)
            out = h
            if drops is not None and drops.get('out') is not None:
                out = out * drops['out'][i]
            outs.append(F.log_softmax(F.linear(out, P['caption_model.logit.weight'], P['caption_model.logit.bias']), 1))
        return torch.stack(outs, 1)

    # ---- forward (NET:488-593) + losses (NET:375-454) ----------------------
    @staticmethod
    def smooth_l1(pred, tgt, inw, outw, sigma, dims):
        """NET:360-373."""
        s2 = sigma ** 2
        d = inw * (pred - tgt)
        ad = d.abs()
        sign = (ad < 1.0 / s2).float()
        l = d.pow(2) * (s2 / 2.0) * sign + (ad - 0.5 / s2) * (1.0 - sign)
        l = outw * l
        for i in sorted(dims, reverse=True):
            l = l.sum(i)
        return l.mean()

    def forward_train(self, blob, samp, drops=None):
        """blob: data (1,H,W,3) f32, im_info (1,3), gt_boxes (1,5), gt_masks (1,H,W) u8, labels (1,T) i64,
        cap_labels (1,T+2) i64, cap_masks (1,T+2) f32.
        samp: dict(rpn_fg_keys, rpn_bg_keys (per anchor uint32), roi_fg_keys, roi_bg_keys (per post-NMS roi),
        roi_bg_rand) or dict(rng=RandomState).  Returns dict of tensors (self.t)."""
        cfg = self.cfg; T = {}
        image = torch.from_numpy(blob['data'].transpose(0, 3, 1, 2).copy())
        im_info = blob['im_info']
        self._im_info = im_info
        base = self.image_to_head(image)
        T['net_conv_base'] = base
        hidden = self.rnn_encoder(torch.from_numpy(self.trim_labels(blob['labels'])), None if drops is None else drops.get('word'))
        T['hidden'] = hidden
        net_conv = self.dynamic_filter(base, hidden)
        T['net_conv'] = net_conv; T['response'] = self.t_response
        H, W = net_conv.shape[2], net_conv.shape[3]
        anchors, _ = B.generate_anchors_pre(H, W, 16, cfg['ANCHOR_SCALES'], cfg['ANCHOR_RATIOS'])
        A = self.A
        # RPN (NET:235-275)
        rpn = F.relu(F.conv2d(net_conv, self.p['rpn_net.weight'], self.p['rpn_net.bias'], padding=1))
        cls = F.conv2d(rpn, self.p['rpn_cls_score_net.weight'], self.p['rpn_cls_score_net.bias'])
        cls_r = cls.view(1, 2, -1, W)
        prob = F.softmax(cls_r, 1).view_as(cls).permute(0, 2, 3, 1)
        cls_rs = cls_r.permute(0, 2, 3, 1).contiguous()
        bbp = F.conv2d(rpn, self.p['rpn_bbox_pred_net.weight'], self.p['rpn_bbox_pred_net.bias']).permute(0, 2, 3, 1).contiguous()
        T['rpn_cls_score_reshape'] = cls_rs; T['rpn_cls_prob'] = prob; T['rpn_bbox_pred'] = bbp
        ct = cfg['TRAIN']
        rois, rscores, order, keep = B.proposal_layer(prob.detach().numpy(), bbp.detach().numpy(), im_info[0], anchors, A,
                                                      ct['RPN_PRE_NMS_TOP_N'], ct['RPN_POST_NMS_TOP_N'], ct['RPN_NMS_THRESH'],
                                                      cfg['NMS_CMP'])
        T['proposal_rois'] = rois; T['proposal_order'] = order; T['proposal_keep'] = keep
        if samp.get('forced_proposals') is not None:
            # parity aid: sort order / NMS keeps are discontinuous in the fp32 scores, so tests may
            # teacher-force the proposal list (rois (k,5), scores (k,)) recorded from the other side.
            rois, rscores = samp['forced_proposals']
        rng = samp.get('rng')
        lab, tg, inw, outw = B.anchor_target_layer(H, W, blob['gt_boxes'], im_info[0], anchors, A, ct,
                                                   samp.get('rpn_fg_keys'), samp.get('rpn_bg_keys'), rng)
        T['rpn_labels'] = lab; T['rpn_bbox_targets'] = tg; T['rpn_bbox_inside'] = inw; T['rpn_bbox_outside'] = outw
        (srois, sscores, slabels, bt, bi, bo, mt, skeep) = B.proposal_target_layer(
            rois, rscores, blob['gt_boxes'], blob['gt_masks'], self.num_classes, ct, cfg['MASK_SIZE'],
            samp.get('roi_fg_keys'), samp.get('roi_bg_keys'), rng, samp.get('roi_bg_rand'))
        T['rois'] = srois; T['labels'] = slabels; T['bbox_targets'] = bt; T['bbox_inside'] = bi
        T['bbox_outside'] = bo; T['mask_targets'] = mt; T['roi_keep'] = skeep
        pool5 = self.crop_pool(net_conv, torch.from_numpy(srois))
        T['pool5'] = pool5
        has_mask = self.var.get('mask', True)
        if self.var.get('backbone') == 'vgg':
            fc7 = self.head_to_tail(pool5, drops)
        else:
            fc7s = self.head_to_tail(pool5)
            T['spatial_fc7'] = fc7s
            fc7 = fc7s.mean(3).mean(2)
        cls_score = F.linear(fc7, self.p['cls_score_net.weight'], self.p['cls_score_net.bias'])
        bbox_pred = F.linear(fc7, self.p['bbox_pred_net.weight'], self.p['bbox_pred_net.bias'])
        T['cls_score'] = cls_score; T['bbox_pred'] = bbox_pred
        nfg = mt.shape[0]
        if has_mask:
            up = F.relu(F.conv_transpose2d(fc7s[:nfg], self.p['mask_up_sampling.weight'], self.p['mask_up_sampling.bias'], stride=2))
            mscore = F.conv2d(up, self.p['mask_pred_net.weight'], self.p['mask_pred_net.bias'])
            T['mask_score'] = mscore
        # ---- losses (NET:375-413) ----
        L = {}
        rl = torch.from_numpy(lab).view(-1).long()
        sel = (rl != -1).nonzero().view(-1)
        L['rpn_cross_entropy'] = F.cross_entropy(cls_rs.view(-1, 2)[sel], rl[sel])
        L['rpn_loss_box'] = self.smooth_l1(bbp, torch.from_numpy(tg), torch.from_numpy(inw), torch.from_numpy(outw), 3.0, [1, 2, 3])
        label = torch.from_numpy(slabels).view(-1).long()
        L['cross_entropy'] = F.cross_entropy(cls_score, label)
        L['loss_box'] = self.smooth_l1(bbox_pred, torch.from_numpy(bt), torch.from_numpy(bi), torch.from_numpy(bo), 1.0, [1])
        total = L['cross_entropy'] + L['loss_box'] + L['rpn_cross_entropy'] + L['rpn_loss_box']
        if has_mask:
            fgl = label[:nfg].view(nfg, 1, 1, 1).expand(nfg, 1, cfg['MASK_SIZE'], cfg['MASK_SIZE'])
            L['loss_mask'] = F.binary_cross_entropy_with_logits(torch.gather(mscore, 1, fgl).squeeze(1), torch.from_numpy(mt))
            total = total + L['loss_mask']
        if self.var['gate'] == 'sigmoid':
            # response loss (network_cycle_response.py:415-423): PIL-NEAREST resize of the uint8 GT mask to the C4 map
            rp = self.t_response[0, 0]
            rt = B.imresize_nearest_u8(blob['gt_masks'][0], (rp.shape[0], rp.shape[1])).astype(np.float32)
            T['response_targets'] = rt
            L['loss_response'] = F.binary_cross_entropy_with_logits(rp, torch.from_numpy(rt))
            total = total + L['loss_response']
        if self.var['cap'] is not None:
            # ---- caption features (NET:415-435; network_cycle_response.py:425-439) ----
            feats = self.head_to_tail(net_conv)
            T['feats_all'] = feats
            att_all = F.adaptive_avg_pool2d(feats, [14, 14]).permute(0, 2, 3, 1)
            if self.var['cap'] == 'mask':
                gm = torch.from_numpy(blob['gt_masks']).unsqueeze(1).float()
                gm = F.adaptive_avg_pool2d(gm, [feats.shape[2], feats.shape[3]])
                gm = (gm >= 0.5).float()
                T['gt_mask_small'] = gm
                att_mask = F.adaptive_avg_pool2d(feats * gm, [14, 14]).permute(0, 2, 3, 1)
                att = torch.cat((att_all, att_mask), 3).contiguous().view(1, 196, -1)
            else:
                feats_b = self.head_to_tail(base)
                T['feats_before_all'] = feats_b
                att_b = F.adaptive_avg_pool2d(feats_b, [14, 14]).permute(0, 2, 3, 1)
                att = torch.cat((att_b, att_all), 3).contiguous().view(1, 196, -1)
            T['att_feats'] = att
            logp = self.caption(att, torch.from_numpy(blob['cap_labels']), drops)
            T['cap_logprobs'] = logp
            tgt = torch.from_numpy(blob['cap_labels'])[:, 1:][:, :logp.shape[1]]
            msk = torch.from_numpy(blob['cap_masks'])[:, 1:][:, :logp.shape[1]]
            L['loss_caption'] = (-logp.gather(2, tgt.unsqueeze(2)).squeeze(2) * msk).sum() / msk.sum()   # CRIT:43-53
            total = total + self.opt['cap_loss_weight'] * L['loss_caption']                                # NET:448
        L['total_loss'] = total
        self.t = T; self.losses = L
        return T, L


    # ---- TEST mode (NET:488-593 with mode == 'TEST', NET:650-658, test_image NET:684-699) ----
    def forward_test(self, blob, forced_proposals=None):
        """returns dict(cls_score, cls_prob, bbox_pred (de-normalised, NET:655-658), rois, net_conv, mask_prob)."""
        cfg = self.cfg
        with torch.no_grad():
            image = torch.from_numpy(blob['data'].transpose(0, 3, 1, 2).copy())
            im_info = blob['im_info']
            self._im_info = im_info
            base = self.image_to_head(image)
            hidden = self.rnn_encoder(torch.from_numpy(self.trim_labels(blob['labels'])))
            net_conv = self.dynamic_filter(base, hidden)
            H, W = net_conv.shape[2], net_conv.shape[3]
            anchors, _ = B.generate_anchors_pre(H, W, 16, cfg['ANCHOR_SCALES'], cfg['ANCHOR_RATIOS'])
            rpn = F.relu(F.conv2d(net_conv, self.p['rpn_net.weight'], self.p['rpn_net.bias'], padding=1))
            cls = F.conv2d(rpn, self.p['rpn_cls_score_net.weight'], self.p['rpn_cls_score_net.bias'])
            prob = F.softmax(cls.view(1, 2, -1, W), 1).view_as(cls).permute(0, 2, 3, 1)
            bbp = F.conv2d(rpn, self.p['rpn_bbox_pred_net.weight'], self.p['rpn_bbox_pred_net.bias']).permute(0, 2, 3, 1).contiguous()
            ct = cfg['TEST']
            if ct.get('MODE', 'nms') == 'top':
                rois, rscores = B.proposal_top_layer(prob.numpy(), bbp.numpy(), im_info[0], anchors, self.A, ct['RPN_TOP_N'])
            else:
                rois, rscores, order, keep = B.proposal_layer(prob.numpy(), bbp.numpy(), im_info[0], anchors, self.A, ct['RPN_PRE_NMS_TOP_N'],
                                                              ct['RPN_POST_NMS_TOP_N'], ct['RPN_NMS_THRESH'], cfg['NMS_CMP'])
            own = rois
            if forced_proposals is not None:
                rois = forced_proposals
            out = self.roi_heads_test(net_conv, rois)
            out.update(rois=rois, own_rois=own, net_conv=net_conv, response=self.t_response)
        return out

    def roi_heads_test(self, net_conv, rois):
        pool5 = self.crop_pool(net_conv, torch.from_numpy(np.ascontiguousarray(rois, dtype=np.float32)))
        vgg = self.var.get('backbone') == 'vgg'
        fc7s = self.head_to_tail(pool5)
        fc7 = fc7s if vgg else fc7s.mean(3).mean(2)
        cls_score = F.linear(fc7, self.p['cls_score_net.weight'], self.p['cls_score_net.bias'])
        bbox_pred = F.linear(fc7, self.p['bbox_pred_net.weight'], self.p['bbox_pred_net.bias'])
        ct = self.cfg['TRAIN']
        stds = torch.tensor(ct['BBOX_NORMALIZE_STDS'], dtype=torch.float32).repeat(self.num_classes)
        means = torch.tensor(ct['BBOX_NORMALIZE_MEANS'], dtype=torch.float32).repeat(self.num_classes)
        if vgg:                                               # network_vgg.py:604-614: no mask branch
            return dict(cls_score=cls_score, cls_prob=F.softmax(cls_score, 1), bbox_pred=bbox_pred * stds + means)
        up = F.relu(F.conv_transpose2d(fc7s, self.p['mask_up_sampling.weight'], self.p['mask_up_sampling.bias'], stride=2))
        mask_prob = torch.sigmoid(F.conv2d(up, self.p['mask_pred_net.weight'], self.p['mask_pred_net.bias']))
        return dict(cls_score=cls_score, cls_prob=F.softmax(cls_score, 1), bbox_pred=bbox_pred * stds + means, mask_prob=mask_prob)

    def predict_masks_from_boxes_and_labels(self, net_conv, boxes, labels):
        """NET:595-626: (n,4) boxes + (n,) labels -> (n,14,14) mask probabilities of the labelled class."""
        with torch.no_grad():
            rois = np.hstack([np.zeros((boxes.shape[0], 1)), boxes]).astype(np.float32)
            mp = self.roi_heads_test(net_conv, rois)['mask_prob']
            idx = torch.from_numpy(np.asarray(labels)).long().view(-1, 1, 1, 1).expand(-1, 1, mp.shape[2], mp.shape[3])
            return torch.gather(mp, 1, idx).squeeze(1)

    def backward(self):
        for k in self.trainable:
            self.p[k].grad = None
        self.losses['total_loss'].backward()
        return {k: (self.p[k].grad if self.p[k].grad is not None else torch.zeros_like(self.p[k])) for k in self.trainable}

    def sgd_step(self, lr=None):
        """torch.optim.SGD as the variant's solver configures it (weights.SOLVERS / param_group: train_val.py:186-205 and its five
        siblings): momentum .9, one group per tensor - weight decay on non-bias tensors only, DOUBLE_BIAS, and lr x 10 on the
        rnn_encoder / dynamic_fc / response keys outside the two cycle solvers; config_vgg's WEIGHT_DECAY / DOUBLE_BIAS for VGG."""
        from .weights import param_group
        ct = self.cfg['TRAIN']
        lr = ct['LEARNING_RATE'] if lr is None else lr
        with torch.no_grad():
            for k in self.trainable:
                p = self.p[k]
                if p.grad is None:                       # torch.optim.SGD skips a parameter without a gradient (resnet.fc)
                    continue
                mult, wd = param_group(self.variant, k, ct)
                d = p.grad + wd * p
                self.momentum[k].mul_(ct['MOMENTUM']).add_(d)
                p.add_(self.momentum[k], alpha=-lr * mult)

    def train_step(self, blob, samp, drops=None, lr=None):
        """NET:702-719: forward, losses, backward, SGD.  Returns the 7 floats."""
        _, L = self.forward_train(blob, samp, drops)
        self.backward()
        self.sgd_step(lr)
        from .weights import loss_keys
        return tuple(float(L[k]) for k in loss_keys(self.variant))
