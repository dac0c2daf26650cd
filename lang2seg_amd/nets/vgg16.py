"""vgg16 — the reference's VGG16 / Faster R-CNN variant (pyutils/mask-faster-rcnn/lib/nets/vgg16.py "VGG" +
nets/network_vgg.py "NETV", experiments/scripts/train_vgg.sh) on the MI355X kernels: conv5_3 map (512 channels, stride 16,
layers before conv3 frozen: VGG:49-51), 7 spatial dynamic filters with sigmoid gating and the response loss
(NETV:411-428,544-546), RPN on 512 channels, 14x14 crop + 2x2 max pool (NETV:139-143), fc6 / fc7 head (VGG:84-88), class and
box losses only (no mask branch).  `train_step` returns 6 floats (rpn_loss_cls, rpn_loss_box, loss_cls, loss_box,
loss_response, total: NETV:736-752).

Everything that is not backbone / RoI head is inherited from nets/resnet_v1.py (encoder, dynamic filters, RPN, proposal and
target layers, losses, streams, launch tape).  fc6 runs as the 7x7 'valid' convolution it is (weight columns permuted from the
reference's (c, y, x) flattening to NHWC at load time, nets/params.py), so it uses the same implicit-GEMM kernels."""
import numpy as np
import torch

from .. import ops as O
from ..model.config import cfg
from .network import ConvOp
from .params import ParamStore
from .variants import solver_cfg
from .resnet_v1 import resnetv1, f32
from . import anchors as ANC


class vgg16(resnetv1):
    variant = 'vgg'

    def __init__(self, opt, batch_size=1):
        resnetv1.__init__(self, opt, batch_size=batch_size, num_layers=101, variant='vgg')

    # ------------------------------------------------------------------ VGG:43-76
    def _init_modules(self):
        self.P = ParamStore(self.opt, 101, self._num_classes, self._num_anchors, 0, self.device, self.dt, 'vgg')
        P = self.P
        C4 = self._C4_feat_dim
        assert C4 == 512, 'conv5_3 has 512 channels (opt C4_feat_dim)'
        A, nc = self._num_anchors, self._num_classes
        self.layers = {}
        self.vgg_plan = []                      # ('conv', idx, ConvOp or None, cin, cout) | ('pool', idx)
        first_trainable_seen = False
        for l in ParamStore.vgg_layers():
            if l[0] == 'pool':
                self.vgg_plan.append(('pool', l[1])); continue
            _, i, cin, cout = l
            wk, bk = 'vgg.features.%d.weight' % i, 'vgg.features.%d.bias' % i
            if i == 0:
                self.vgg_plan.append(('conv', i, None, cin, cout)); continue          # 3-channel input: its own kernel
            trainable = wk in P.offsets
            need_dx = trainable and first_trainable_seen      # nothing trainable below the first trainable conv (conv3_1)
            first_trainable_seen = first_trainable_seen or trainable
            self.vgg_plan.append(('conv', i, ConvOp(self, wk, cin, cout, 3, 1, 1, bias_key=bk, need_dgrad=need_dx), cin, cout))
        self.fc6 = ConvOp(self, 'vgg.classifier.0.weight', 512, 4096, 7, 1, 0, bias_key='vgg.classifier.0.bias', full_map=True)
        self.fc7 = ConvOp(self, 'vgg.classifier.3.weight', 4096, 4096, 1, 1, 0, bias_key='vgg.classifier.3.bias')
        self.rpn_conv = ConvOp(self, 'rpn_net.weight', C4, 512, 3, 1, 1, bias_key='rpn_net.bias')
        self.rpn_heads = ConvOp(self, None, 512, 6 * A, group=('rpn_head_w', 'rpn_head_b'), Cout_pad=P.rpn_npad)
        self.rcnn_heads = ConvOp(self, None, 4096, 5 * nc, group=('rcnn_w', 'rcnn_b'), Cout_pad=P.rcnn_npad)
        self._NFP = 7 * C4 + 7
        self.base_anchors = torch.from_numpy(ANC.base_anchors(self._anchor_scales, self._anchor_ratios)).to(self.device)
        self.init_weights()
        sc = solver_cfg(self.variant)           # the param groups of this variant's solver (model/train_val.make_optimizer rebuilds them)
        P.build_segments(double_bias=sc.TRAIN.DOUBLE_BIAS, bias_decay=sc.TRAIN.BIAS_DECAY)
        self.load_state_dict(self._initial_state, strict=False)
        del self._initial_state

    def init_weights(self):
        """NETV:333-355 for the heads; the VGG trunk keeps torchvision's defaults in the reference (it is normally overwritten by
        the pretrained detector): here He-normal convs / N(0, 0.01) classifier, host RNG, once."""
        g = torch.Generator().manual_seed(cfg.RNG_SEED)
        sd = {}
        for k, shp in self.P.shapes.items():
            n = int(np.prod(shp))
            if k.startswith('vgg.features.') and len(shp) == 4:
                t = torch.randn(n, generator=g) * float(np.sqrt(2.0 / (shp[1] * 9)))
                if k == 'vgg.features.0.weight':
                    t = t * 0.02                                      # inputs are pixel-scale (sigma ~50)
            elif k.startswith('vgg.classifier.') and len(shp) == 2:
                t = torch.randn(n, generator=g) * 0.01
            elif k.startswith('vgg.'):
                t = torch.zeros(n)
            elif k.startswith(('rpn_', 'cls_score')) and k.endswith('weight'):
                t = torch.randn(n, generator=g) * 0.01
            elif k.startswith('bbox_pred_net') and k.endswith('weight'):
                t = torch.randn(n, generator=g) * 0.001
            elif k.startswith(('rpn_', 'cls_score', 'bbox_pred')) and k.endswith('bias'):
                t = torch.zeros(n)
            elif 'embedding.weight' in k:
                t = torch.randn(n, generator=g)
            else:
                kk = 1.0 / float(np.sqrt(shp[-1] if len(shp) > 1 else 512))
                t = (torch.rand(n, generator=g) * 2 - 1) * kk
            sd[k] = t.view(shp)
        self._initial_state = sd

    def _make_transposes(self):
        class _T(object):
            pass
        P = self.P
        self.wT, self.extra_transposes = {}, []
        def add(name, src, N, K):
            e = _T(); e.w_master, e.scale, e.Np, e.k, e.Cin, e.force_f32 = src, None, N, 1, K, 1
            e.wb = torch.zeros(K * N, dtype=torch.float32, device=self.device)
            self.extra_transposes.append(e)
            self.wT[name] = (e.wb, N, K)
        for sfx in ['', '_reverse']:                         # (row-batch data gradients need no copy: resnet_v1.bwd_x)
            w = 'rnn_encoder.rnn.weight_hh_l0'
            add(w + sfx, P.view(w + sfx), *P.shapes[w + sfx])

    # ------------------------------------------------------------------ backbone (VGG:53-54,78-82)
    def _backbone_fwd(self, d, saved):
        P = self.P
        # The frozen prefix (conv1_1 .. conv2_2: 56 GMAC at 600x1000, ~0.35 ms) runs beside the previous step's tail - VGG's update moves 3.3 GB - and
        # the main queue joins that tail in front of the launch that WRITES the first trainable convolution's input (the pooling behind conv2_2): the
        # previous step's weight gradient of conv3_1 reads that buffer (the race of resnet_v1._backbone_fwd, round 5).
        first_tr = next((k for k, l in enumerate(self.vgg_plan) if l[0] == 'conv' and l[2] is not None and l[2].trainable), 0)
        join_at = first_tr - 1 if (first_tr > 0 and self.vgg_plan[first_tr - 1][0] == 'pool') else first_tr
        H, W = int(d['data'].shape[1]), int(d['data'].shape[2])
        x, h, w, c = None, H, W, 3
        acts = []                               # (kind, idx, input tensor, h, w, channels, output tensor)
        for k, l in enumerate(self.vgg_plan):
            if k == join_at:
                self.join_update(full=False)
                self._mark('frozen prefix done, update joined')
            if l[0] == 'pool':
                oh, ow = h // 2, w // 2
                y = self.buf('vgg.p%d' % l[1], (oh * ow, c))
                O.maxpool2x2_fwd(x, y, 1, h, w, c)
                acts.append(('pool', l[1], x, h, w, c, y))
                x, h, w = y, oh, ow
                continue
            _, i, op, cin, cout = l
            y = self.buf('vgg.a%d' % i, (h * w, cout))
            if op is None:
                O.conv3x3_c3(d['data'], P.frozen['vgg.features.0.weight'], P.frozen['vgg.features.0.bias'], y, h, w)
            else:
                op.fwd(x, 1, h, w, y, relu=True)
            acts.append(('conv', i, x, h, w, cin, y))
            x, c = y, cout
        saved['vgg'] = acts
        return x, h, w

    def _backbone_bwd(self, dbase, saved, S, main, dp):
        """conv5_3 .. conv3_1 (VGG:49-51: conv1_x, conv2_x fixed).  g is always the gradient w.r.t. a conv's pre-activation."""
        ops = {l[1]: l[2] for l in self.vgg_plan if l[0] == 'conv'}
        acts = saved['vgg']
        g = dbase                                            # masked by conv5_3's output > 0 in the dynamic-filter backward
        k = len(acts) - 1
        while k >= 0:
            kind, i, xin, h, w, cin, y = acts[k]
            if kind == 'pool':
                # g here is the gradient w.r.t. the pooled map; un-pool onto the ReLU output that fed the pool
                dx = self.buf('vgg.dp%d' % i, (h * w, cin))
                O.maxpool2x2_bwd(g, xin, dx, 1, h, w, cin, True)
                g = dx; k -= 1
                continue
            op = ops[i]
            if op is None or not op.trainable:
                break
            op.wgrad(g, xin, 1, h, w)
            if not op.need_dgrad:
                break
            dx = self.buf('vgg.dx%d' % i, (h * w, cin))
            prev_is_pool = k > 0 and acts[k - 1][0] == 'pool'
            op.dgrad(g, 1, h, w, dx, ref=None if prev_is_pool else xin)
            g = dx; k -= 1
        self.flush_wgrads('vgg backbone')                   # before the bucket goes to the reducer (dp_ready flushes too: belt and braces)
        if dp is not None:
            self.dp_ready('layer3')

    # ------------------------------------------------------------------ RoI head (NETV:139-143, VGG:84-88, NETV:274-288)
    def _crop_max_pool(self):
        return True                                       # network_vgg.py:139 _crop_pool_layer(bottom, rois, max_pool=True)

    def _roi_head_fwd(self, net_conv, Hc, Wc, rois, R, FGM, saved):
        P, t = self.P, self.t
        C4, nc = self._C4_feat_dim, self._num_classes
        PS = int(cfg.POOLING_SIZE)
        pool5 = self._rois_pool_fwd(net_conv, Hc, Wc, rois, R, saved)      # 14x14 crop + 2x2 max pool (Network._crop_pool_layer default)
        h6 = self.buf('roi.fc6', (R, 4096))
        self.fc6.fwd(pool5, R, PS, PS, h6, relu=True)
        d6 = self._drop('fc6', (R, 4096), 0.5)
        h6d = h6
        if d6 is not None:
            h6d = self.buf('roi.fc6d', (R, 4096)); O.scale_mask(h6, d6, None, h6d)
        h7 = self.buf('roi.fc7', (R, 4096))
        self.fc7.fwd(h6d, R, 1, 1, h7, relu=True)
        d7 = self._drop('fc7', (R, 4096), 0.5)
        h7d = h7
        if d7 is not None:
            h7d = self.buf('roi.fc7d', (R, 4096)); O.scale_mask(h7, d7, None, h7d)
        NPC = P.rcnn_npad
        cheads = self.buf('roi.heads', (R, NPC), f32)
        self.rcnn_heads.fwd(h7d, R, 1, 1, cheads, out_f32=True)
        t.update({'pool5': pool5, 'rcnn_heads': cheads})
        saved['roi'] = (pool5, h6, h6d, d6, h7, h7d, d7)
        return cheads, NPC, None

    def _roi_heads_test(self, net_conv, Hc, Wc, rois, n, labels=None):
        """TEST mode (network_vgg.py:588-614): crop-pool -> fc6 -> fc7 -> class scores / probabilities / de-normalised deltas; no masks."""
        P = self.P
        nc, PS = self._num_classes, int(cfg.POOLING_SIZE)
        pool5 = self._rois_pool_fwd(net_conv, Hc, Wc, rois, n, {})
        h6 = self.buf('roi.fc6', (n, 4096))
        self.fc6.fwd(pool5, n, PS, PS, h6, relu=True)
        h7 = self.buf('roi.fc7', (n, 4096))
        self.fc7.fwd(h6, n, 1, 1, h7, relu=True)
        NPC = P.rcnn_npad
        cheads = self.buf('roi.heads', (n, NPC), f32)
        self.rcnn_heads.fwd(h7, n, 1, 1, cheads, out_f32=True)
        cst = self._consts()
        cls_prob = self.buf('test.cls_prob', (n, nc), f32); bbox_pred = self.buf('test.bbox_pred', (n, 4 * nc), f32)
        O.rcnn_predict(cheads, NPC, n, nc, cst['stds'], cst['means'], cls_prob, bbox_pred)
        return cheads, cls_prob, bbox_pred, None

    def _predict_masks_from_boxes_and_labels(self, net_conv, boxes, labels):
        raise NotImplementedError('the VGG16 / Faster R-CNN network has no mask branch (network_vgg.py:614; model/test_vgg.py evaluates boxes only)')

    def _roi_head_bwd(self, d_cheads, dscore, labels, counts, rois, Hc, Wc, R, FGM, saved):
        C4 = self._C4_feat_dim
        PS = int(cfg.POOLING_SIZE)
        pool5, h6, h6d, d6, h7, h7d, d7 = saved['roi']
        self.rcnn_heads.wgrad(d_cheads, h7d, R, 1, 1)
        g7 = self.buf('roi.dfc7', (R, 4096))
        if d7 is None:
            self.rcnn_heads.dgrad(d_cheads, R, 1, 1, g7, ref=h7)          # -> gradient of fc7's pre-activation
        else:
            self.rcnn_heads.dgrad(d_cheads, R, 1, 1, g7)
            O.scale_mask(g7, d7, h7, g7)
        self.fc7.wgrad(g7, h6d, R, 1, 1)
        g6 = self.buf('roi.dfc6', (R, 4096))
        if d6 is None:
            self.fc7.dgrad(g7, R, 1, 1, g6, ref=h6)
        else:
            self.fc7.dgrad(g7, R, 1, 1, g6)
            O.scale_mask(g6, d6, h6, g6)
        self.fc6.wgrad(g6, pool5, R, PS, PS)
        dpool5 = self.buf('roi.dpool5', (R * PS * PS, C4))
        self.fc6.dgrad(g6, R, PS, PS, dpool5)
        self._mark('roi head bwd')
        return self._rois_pool_bwd(dpool5, Hc, Wc, rois, R, saved)
