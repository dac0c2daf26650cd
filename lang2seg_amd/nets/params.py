"""Parameter storage for the MI355X train step.

All trainable tensors live in ONE flat fp32 device buffer (plus flat gradient and momentum
buffers of the same layout) so that: the fused SGD kernel updates everything in one launch,
the data-parallel all-reduce works on contiguous buckets, and a dtype "shadow" copy (bf16 or
f32, frozen-BN scale folded in) is rewritten by the same SGD launch.  Key names and shapes of
`state_dict()` are the reference's checkpoint format (RES:275-337, ENC:11-25,
ATT:27-57,426-443); internally conv weights are stored [Cout][KH][KW][Cin] (NHWC kernels) and
the 2x2 ConvTranspose as [Cin][dy][dx][Cout]."""
import ctypes as C
import numpy as np
import torch

from .. import ops as O
from .._lib import SgdSeg

RESNET_LAYERS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}
BN_EPS = 1e-5


def to_internal(name, t):
    if name == 'vgg.classifier.0.weight':
        # fc6 reads pool5.view(n, -1) of an NCHW tensor (vgg16.py:84-86): columns (c, y, x) -> the NHWC order (y, x, c), i.e. the
        # weight of a 7x7 'valid' convolution [O][KH][KW][I]
        return t.view(t.shape[0], 512, 7, 7).permute(0, 2, 3, 1).contiguous().view(t.shape[0], -1)
    if t.dim() == 4 and name == 'mask_up_sampling.weight':
        return t.permute(0, 2, 3, 1).contiguous()          # (Cin,Cout,2,2) -> [ci][dy][dx][co]
    if t.dim() == 4:
        return t.permute(0, 2, 3, 1).contiguous()          # OIHW -> OHWI
    return t.contiguous()


def from_internal(name, t, ref_shape):
    if name == 'vgg.classifier.0.weight':
        return t.view(ref_shape[0], 7, 7, 512).permute(0, 3, 1, 2).contiguous().view(ref_shape)
    if len(ref_shape) == 4 and name == 'mask_up_sampling.weight':
        ci, co, kh, kw = ref_shape
        return t.view(ci, kh, kw, co).permute(0, 3, 1, 2).contiguous()
    if len(ref_shape) == 4:
        o, i, kh, kw = ref_shape
        return t.view(o, kh, kw, i).permute(0, 3, 1, 2).contiguous()
    return t.view(ref_shape).contiguous()


class ParamStore(object):
    def __init__(self, opt, num_layers, num_classes, num_anchors, fixed_blocks, device, dt, variant='cycle'):
        from .variants import VARIANTS
        self.opt, self.device, self.dt = opt, device, dt
        self.variant, self.var = variant, VARIANTS[variant]
        self.num_classes, self.A = num_classes, num_anchors
        self.nblocks = RESNET_LAYERS[num_layers]
        self.is_vgg = self.var.get('backbone') == 'vgg'
        self.fc7_dim = 4096 if self.is_vgg else 2048
        self.fixed_blocks = fixed_blocks
        self.shapes = self._shapes()
        self._layout()

    # ---- registry ------------------------------------------------------
    def _shapes(self):
        o = self.opt; s = {}
        V, E, WV, Hh = o['vocab_size'], o['word_embedding_size'], o['word_vec_size'], o['rnn_hidden_size']
        s['rnn_encoder.embedding.weight'] = (V, E)
        s['rnn_encoder.mlp.0.weight'] = (WV, E); s['rnn_encoder.mlp.0.bias'] = (WV,)
        for sfx in ['', '_reverse']:
            s['rnn_encoder.rnn.weight_ih_l0' + sfx] = (4 * Hh, WV)
            s['rnn_encoder.rnn.weight_hh_l0' + sfx] = (4 * Hh, Hh)
            s['rnn_encoder.rnn.bias_ih_l0' + sfx] = (4 * Hh,)
            s['rnn_encoder.rnn.bias_hh_l0' + sfx] = (4 * Hh,)
        if self.var['cap'] is not None:
            R, IE, AH = o['rnn_size'], o['input_encoding_size'], o['att_hid_size']
            s['caption_model.embed.0.weight'] = (V + 1, IE)
            s['caption_model.att_embed.0.weight'] = (R, o['att_feat_size']); s['caption_model.att_embed.0.bias'] = (R,)
            s['caption_model.logit.weight'] = (V + 1, R); s['caption_model.logit.bias'] = (V + 1,)
            s['caption_model.ctx2att.weight'] = (AH, R); s['caption_model.ctx2att.bias'] = (AH,)
            s['caption_model.core.a2c.weight'] = (2 * R, R); s['caption_model.core.a2c.bias'] = (2 * R,)
            s['caption_model.core.i2h.weight'] = (5 * R, IE); s['caption_model.core.i2h.bias'] = (5 * R,)
            s['caption_model.core.h2h.weight'] = (5 * R, R); s['caption_model.core.h2h.bias'] = (5 * R,)
            s['caption_model.core.attention.h2att.weight'] = (AH, R); s['caption_model.core.attention.h2att.bias'] = (AH,)
            s['caption_model.core.attention.alpha_net.weight'] = (1, AH); s['caption_model.core.attention.alpha_net.bias'] = (1,)

        def bn(p, c):
            for k in ['weight', 'bias', 'running_mean', 'running_var']:
                s[p + '.' + k] = (c,)
        if self.is_vgg:
            return self._shapes_vgg(s)
        s['resnet.conv1.weight'] = (64, 3, 7, 7); bn('resnet.bn1', 64)
        inpl = 64
        for li, (planes, nb) in enumerate(zip([64, 128, 256, 512], self.nblocks), 1):
            for b in range(nb):
                p = 'resnet.layer%d.%d' % (li, b)
                s[p + '.conv1.weight'] = (planes, inpl, 1, 1); bn(p + '.bn1', planes)
                s[p + '.conv2.weight'] = (planes, planes, 3, 3); bn(p + '.bn2', planes)
                s[p + '.conv3.weight'] = (planes * 4, planes, 1, 1); bn(p + '.bn3', planes * 4)
                if b == 0:
                    s[p + '.downsample.0.weight'] = (planes * 4, inpl, 1, 1); bn(p + '.downsample.1', planes * 4)
                inpl = planes * 4
        s['resnet.fc.weight'] = (1000, 2048); s['resnet.fc.bias'] = (1000,)     # present in the module, never used (RES:133)
        C4 = o['C4_feat_dim']; HD = o['rnn_num_layers'] * (2 if o['bidirectional'] else 1) * Hh
        if self.var['nfilt'] == 1:
            s['dynamic_fc.weight'] = (C4, HD); s['dynamic_fc.bias'] = (C4,)
        else:
            for k in range(7):
                s['dynamic_fc_%d.weight' % k] = (C4, HD); s['dynamic_fc_%d.bias' % k] = (C4,)
            s['response_fc.weight'] = (7, HD); s['response_fc.bias'] = (7,)
        A, nc = self.A, self.num_classes
        s['rpn_net.weight'] = (512, C4, 3, 3); s['rpn_net.bias'] = (512,)
        s['rpn_cls_score_net.weight'] = (2 * A, 512, 1, 1); s['rpn_cls_score_net.bias'] = (2 * A,)
        s['rpn_bbox_pred_net.weight'] = (4 * A, 512, 1, 1); s['rpn_bbox_pred_net.bias'] = (4 * A,)
        s['cls_score_net.weight'] = (nc, 2048); s['cls_score_net.bias'] = (nc,)
        s['bbox_pred_net.weight'] = (4 * nc, 2048); s['bbox_pred_net.bias'] = (4 * nc,)
        s['mask_up_sampling.weight'] = (2048, 256, 2, 2); s['mask_up_sampling.bias'] = (256,)
        s['mask_pred_net.weight'] = (nc, 256, 1, 1); s['mask_pred_net.bias'] = (nc,)
        return s

    VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]   # torchvision 'D' minus the last pool

    @classmethod
    def vgg_layers(cls):
        """[('conv', index in vgg.features, Cin, Cout) | ('pool', index)] in execution order (vgg16.py:53-54)."""
        out, i, cin = [], 0, 3
        for v in cls.VGG_CFG:
            if v == 'M':
                out.append(('pool', i)); i += 1
            else:
                out.append(('conv', i, cin, v)); cin = v; i += 2
        return out

    def _shapes_vgg(self, s):
        o = self.opt
        for l in self.vgg_layers():
            if l[0] == 'conv':
                s['vgg.features.%d.weight' % l[1]] = (l[3], l[2], 3, 3); s['vgg.features.%d.bias' % l[1]] = (l[3],)
        s['vgg.classifier.0.weight'] = (4096, 512 * 7 * 7); s['vgg.classifier.0.bias'] = (4096,)
        s['vgg.classifier.3.weight'] = (4096, 4096); s['vgg.classifier.3.bias'] = (4096,)
        C4 = o['C4_feat_dim']; HD = o['rnn_num_layers'] * (2 if o['bidirectional'] else 1) * o['rnn_hidden_size']
        for k in range(7):
            s['dynamic_fc_%d.weight' % k] = (C4, HD); s['dynamic_fc_%d.bias' % k] = (C4,)
        s['response_fc.weight'] = (7, HD); s['response_fc.bias'] = (7,)
        A, nc = self.A, self.num_classes
        s['rpn_net.weight'] = (512, C4, 3, 3); s['rpn_net.bias'] = (512,)
        s['rpn_cls_score_net.weight'] = (2 * A, 512, 1, 1); s['rpn_cls_score_net.bias'] = (2 * A,)
        s['rpn_bbox_pred_net.weight'] = (4 * A, 512, 1, 1); s['rpn_bbox_pred_net.bias'] = (4 * A,)
        s['cls_score_net.weight'] = (nc, 4096); s['cls_score_net.bias'] = (nc,)
        s['bbox_pred_net.weight'] = (4 * nc, 4096); s['bbox_pred_net.bias'] = (4 * nc,)
        return s

    def is_trainable(self, k):
        if k.startswith('vgg.features.'):
            return int(k.split('.')[2]) >= 10              # vgg16.py:49-51: layers before conv3 are fixed
        """RES:290-306: conv1/bn1, layer1..FIXED_BLOCKS and every BN tensor are frozen; resnet.fc never gets a gradient."""
        if k.startswith('resnet.'):
            if '.bn' in k or 'downsample.1' in k or k.startswith('resnet.bn1') or k.startswith('resnet.conv1') or k.startswith('resnet.fc'):
                return False
            for fb in range(1, self.fixed_blocks + 1):
                if k.startswith('resnet.layer%d.' % fb):
                    return False
        return True

    def shadow_only(self, k):
        """True when every kernel reads trainable tensor `k` through the dtype shadow ONLY (ParamStore.shadow = bf16(rowscale * param)) and
        never through the fp32 master in ParamStore.param: the weights of the trainable ConvOps in bf16 mode - forward from ConvOp.wf (a
        view of the shadow), data gradient from the transposed copy refresh_weights builds FROM the shadow (nets/network.py).  Everything
        else is read as fp32 master somewhere: every bias (ConvOp.bias), the language encoder and the captioner (resnet_v1.py _encoder /
        _caption*), the dynamic-filter FCs (dyn_w / dyn_b), mask_pred_net.weight (maskpred_bwd), mask_up_sampling.weight (its forward
        operand is a transpose of the master), and in f32 mode all of it.  The data-parallel sharded update puts the shadow on the wire only
        for ranges of such tensors and the masters for the rest (parallel.GradReducer);
        tests/test_train_step_gpu.py::test_sharded_update_stale_masters_are_never_read poisons the masters this predicate says nobody reads."""
        from .._lib import BF16
        if self.dt != BF16 or not k.endswith('.weight') or k not in self.offsets:
            return False
        if k.startswith(('resnet.layer', 'vgg.features.', 'vgg.classifier.')):
            return True
        return k in ('rpn_net.weight', 'cls_score_net.weight', 'bbox_pred_net.weight', 'rpn_cls_score_net.weight', 'rpn_bbox_pred_net.weight',
                     'caption_model.att_embed.0.weight')

    def shadow_only_runs(self):
        """the flat buffer as maximal runs [(lo, hi, shadow_only)] in offset order (alignment gaps / group padding go with the tensor in front)"""
        runs = []
        ks = self.trainable
        for i, k in enumerate(ks):
            lo = self.offsets[k] if i else 0
            hi = self.offsets[ks[i + 1]] if i + 1 < len(ks) else self.total
            so = self.shadow_only(k)
            if runs and runs[-1][2] == so:
                runs[-1][1] = hi
            else:
                runs.append([lo, hi, so])
        return [tuple(r) for r in runs]

    @staticmethod
    def bn_of(k):
        """BN module whose frozen scale folds into conv weight `k` (None if none)."""
        if not k.startswith('resnet.') or not k.endswith('.weight'):
            return None
        if k == 'resnet.conv1.weight':
            return 'resnet.bn1'
        if '.conv' in k:
            return k.replace('.conv', '.bn')[:-len('.weight')]
        if 'downsample.0' in k:
            return k.replace('downsample.0', 'downsample.1')[:-len('.weight')]
        return None

    def _layout(self):
        """Flat order = reverse execution order (gradients become final in this order during backward, so
        contiguous buckets can be all-reduced while earlier layers are still back-propagating)."""
        names = list(self.shapes.keys())
        tr = [k for k in names if self.is_trainable(k)]
        order = []
        def take(pred):
            sel = [k for k in tr if pred(k) and k not in order]
            order.extend(sel)
        # (att_embed last: its weight gradient is a convolution weight gradient of the heads stage, so that the tensors of that stage -
        # att_embed, layer4, RoI / mask heads, RPN - form ONE contiguous range, `defer_range`, see optim.SGD.defer)
        take(lambda k: k.startswith('caption_model.') and '.att_embed.' not in k)
        n_cap_rest = len(order)
        # (bias in front of the weight: the weight is read through the dtype shadow only and then adjoins layer4's, shadow_only() below)
        take(lambda k: k.startswith('caption_model.') and k.endswith('.bias'))
        take(lambda k: k.startswith('caption_model.'))
        take(lambda k: k.startswith('vgg.classifier.3.'))
        take(lambda k: k.startswith('vgg.classifier.0.'))
        for b in reversed(range(self.nblocks[3])):
            take(lambda k: k.startswith('resnet.layer4.%d.' % b))
        # grouped heads (contiguous on purpose: they are used as one concatenated GEMM operand)
        self.groups = {
            'rcnn_w': ['cls_score_net.weight', 'bbox_pred_net.weight'], 'rcnn_b': ['cls_score_net.bias', 'bbox_pred_net.bias'],
            'rpn_head_w': ['rpn_cls_score_net.weight', 'rpn_bbox_pred_net.weight'],
            'rpn_head_b': ['rpn_cls_score_net.bias', 'rpn_bbox_pred_net.bias'],
            'dyn_w': ['dynamic_fc.weight'] if self.var['nfilt'] == 1 else ['dynamic_fc_%d.weight' % k for k in range(7)] + ['response_fc.weight'],
            'dyn_b': ['dynamic_fc.bias'] if self.var['nfilt'] == 1 else ['dynamic_fc_%d.bias' % k for k in range(7)] + ['response_fc.bias'],
        }
        plan = ['@rcnn_w', '@rcnn_b'] + ([] if self.is_vgg else ['mask_up_sampling.weight', 'mask_up_sampling.bias', 'mask_pred_net.weight', 'mask_pred_net.bias']) + \
               ['rpn_net.weight', 'rpn_net.bias', '@rpn_head_w', '@rpn_head_b', '@dyn_w', '@dyn_b']
        self.offsets, self.group_off = {}, {}
        off = 0
        def place(k, align=True):
            nonlocal off
            if align:
                off = (off + 63) // 64 * 64      # 256-byte aligned tensors (16-byte vector loads in every dtype)
            self.offsets[k] = off
            off += int(np.prod(self.shapes[k]))
        for k in list(order):
            place(k)
        nc = self.num_classes
        self.rcnn_n = 5 * nc; self.rcnn_npad = (5 * nc + 7) // 8 * 8
        self.rpn_n = 6 * self.A; self.rpn_npad = (6 * self.A + 7) // 8 * 8
        for item in plan:
            if item.startswith('@'):
                g = item[1:]
                off = (off + 63) // 64 * 64
                self.group_off[g] = off
                for k in self.groups[g]:
                    place(k, align=False)        # members of a group are back to back
                if g == 'rcnn_w':
                    off += (self.rcnn_npad - self.rcnn_n) * self.fc7_dim
                elif g == 'rcnn_b':
                    off += self.rcnn_npad - self.rcnn_n
                elif g == 'rpn_head_w':
                    off += (self.rpn_npad - self.rpn_n) * 512
                elif g == 'rpn_head_b':
                    off += self.rpn_npad - self.rpn_n
            else:
                off = (off + 63) // 64 * 64
                place(item)
            order.append(item)
        rest_pred = [lambda k: k.startswith('rnn_encoder.')]
        for p in rest_pred:
            for k in [k for k in tr if p(k) and k not in self.offsets]:
                off = (off + 63) // 64 * 64
                place(k)
        for sfx in ('.bias', '.weight'):                  # the VGG trunk: its biases (read as fp32 masters) together, then its weights (shadow_only())
            for i in reversed(range(31)):
                for k in [k for k in tr if k.startswith('vgg.features.%d.' % i) and k.endswith(sfx) and k not in self.offsets]:
                    off = (off + 63) // 64 * 64
                    place(k)
        for li in (3, 2, 1):
            for b in reversed(range(self.nblocks[li - 1])):
                for k in [k for k in tr if k.startswith('resnet.layer%d.%d.' % (li, b)) and k not in self.offsets]:
                    off = (off + 63) // 64 * 64
                    place(k)
        missing = [k for k in tr if k not in self.offsets]
        assert not missing, missing
        self.total = (off + 63) // 64 * 64
        self.trainable = sorted(self.offsets.keys(), key=lambda k: self.offsets[k])
        # [lo, hi) of the flat buffer whose gradients come from the heads stage's grouped weight-gradient launches (+ their biases)
        self.defer_range = (self.offsets[order[n_cap_rest]] if n_cap_rest < len(order) else 0, self.group_off['dyn_w'])
        dev = self.device
        self.param = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.mom = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(self.total, dtype=O.TORCH_DT[self.dt], device=dev)
        self.frozen = {k: torch.zeros(self.shapes[k], dtype=torch.float32, device=dev) for k in self.shapes if k not in self.offsets}
        # frozen-BN folded scale / bias per conv (RES:301-306,363-368: BN always in eval mode)
        self.bn_scale, self.bn_bias = {}, {}
        for k in self.shapes:
            if self.bn_of(k) is not None:        # allocated once: kernels keep raw pointers to these
                self.bn_scale[k] = torch.ones(self.shapes[k][0], dtype=torch.float32, device=dev)
                self.bn_bias[k] = torch.zeros(self.shapes[k][0], dtype=torch.float32, device=dev)
        # per-row gradient/shadow scale table for the SGD kernel
        rs_list, self.rowscale_off = [], {}
        n = 0
        for k in self.trainable:
            bn = self.bn_of(k)
            if bn is not None:
                self.rowscale_off[k] = n
                n += self.shapes[k][0]
        self.rowscale = torch.ones(max(n, 1), dtype=torch.float32, device=dev)

    # ---- views ------------------------------------------------------------
    def view(self, k, buf=None):
        buf = self.param if buf is None else buf
        o = self.offsets[k]
        return buf[o:o + int(np.prod(self.shapes[k]))]

    def gview(self, g, count, buf=None):
        buf = self.param if buf is None else buf
        o = self.group_off[g]
        return buf[o:o + count]

    # ---- (de)serialisation --------------------------------------------------
    def load_state_dict(self, sd, strict=False):
        """name+shape matched copy (TV:262-281 semantics live in the solver); tolerates missing
        num_batches_tracked and torch-0.3 checkpoints."""
        for k, shp in self.shapes.items():
            if k not in sd:
                if strict:
                    raise KeyError(k)
                continue
            t = sd[k]
            t = torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t.detach().cpu()
            t = t.float()
            assert tuple(t.shape) == tuple(shp), (k, tuple(t.shape), shp)
            ti = to_internal(k, t).reshape(-1).to(self.device)
            if k in self.offsets:
                self.view(k).copy_(ti)
            else:
                self.frozen[k].copy_(ti.view(self.frozen[k].shape))
        self._fold_bn()
        self.refresh_shadow_full()

    def state_dict(self):
        out = {}
        for k, shp in self.shapes.items():
            if k in self.offsets:
                out[k] = from_internal(k, self.view(k).detach().clone(), shp).cpu()
            else:
                # frozen tensors hold the internal (OHWI) order inside a reference-shaped allocation
                out[k] = from_internal(k, self.frozen[k].detach().clone().reshape(-1), shp).cpu()
        return out

    def _fold_bn(self):
        for k in self.shapes:
            bn = self.bn_of(k)
            if bn is None:
                continue
            g, b = self.frozen[bn + '.weight'], self.frozen[bn + '.bias']
            m, v = self.frozen[bn + '.running_mean'], self.frozen[bn + '.running_var']
            s = g / torch.sqrt(v + BN_EPS)
            self.bn_scale[k].copy_(s)
            self.bn_bias[k].copy_(b - m * s)
            if k in self.rowscale_off:
                o = self.rowscale_off[k]
                self.rowscale[o:o + s.numel()].copy_(s)

    # ---- SGD segment table: the param groups of the variant's construct_graph() ------------
    def param_group(self, k, double_bias=False, bias_decay=False, lang_lr_mult=None, from_frcn=False, gamma=0.1):
        """(lr factor, weight-decay switch) of tensor `k` as the variant's solver groups it (FROM_FRCN False): train_val.py:186-205,
        train_val_response.py:186-205, train_val_vgg.py:186-205 give keys containing rnn_encoder / dynamic_fc / response lr x 10 (biases
        x 10 x (DOUBLE_BIAS + 1)); train_val_cycle.py:192-215 and train_val_cycle_response.py:186-209 do not (variants.SOLVERS)."""
        from .variants import SOLVERS, LANG_LR_KEYS
        mult = SOLVERS[self.variant]['lang_lr_mult'] if lang_lr_mult is None else float(lang_lr_mult)
        is_bias = 'bias' in k
        if from_frcn:
            # cfg.TRAIN.FROM_FRCN (train_val.py:175-185, the same in all six solvers): fine-tuning from a detector - the mask branch at the full
            # learning rate, everything else at lr x GAMMA; no language-side factor in this branch
            f = 1.0 if 'mask' in k else float(gamma)
        else:
            f = mult if any(t in k for t in LANG_LR_KEYS) else 1.0
        if is_bias and double_bias:
            f *= 2.0
        return f, (1 if (not is_bias or bias_decay) else 0)

    def build_segments(self, double_bias=False, bias_decay=False, lang_lr_mult=None, from_frcn=False, gamma=0.1):
        segs = []
        for k in self.trainable:
            cnt = int(np.prod(self.shapes[k]))
            sg = SgdSeg()
            sg.offset, sg.count = self.offsets[k], cnt
            sg.row_len = cnt // self.shapes[k][0] if k in self.rowscale_off else 1
            sg.lr_mult, sg.weight_decay = self.param_group(k, double_bias, bias_decay, lang_lr_mult, from_frcn, gamma)
            sg.rowscale_off = self.rowscale_off.get(k, -1)
            segs.append(sg)
        self.seg_rule = (bool(double_bias), bool(bias_decay), lang_lr_mult, bool(from_frcn), float(gamma))
        CH = int(O.sgd_chunk())

        def table(seq):
            run = 0
            for g in seq:                                             # chunk0: the update kernel finds a chunk's segment by it
                g.chunk0 = run
                run += -(-int(g.count) // CH)
            arr = (SgdSeg * len(seq))(*seq)
            self._seg_tables.append(arr)
            return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self._seg_tables, self._ow_key = [], frozenset()
        self.nseg = len(segs)
        self.seg_ends = [int(g.offset + g.count) for g in segs]      # host copy: the optimiser's partial updates split here
        self.seg_offs = [int(g.offset) for g in segs]
        self.seg_chunks = [-(-int(g.count) // CH) for g in segs]
        self.seg_chunk0 = [sum(self.seg_chunks[:i]) for i in range(len(segs))]
        self.sgd_chunk = CH
        self.seg_size = C.sizeof(SgdSeg)
        self.segs_dev = table(segs)
        # the same segments with the deferred range moved to the end: [0, n_rest) is updated at the end of the step, [n_rest, nseg) behind
        # the deferred weight gradients (optim.SGD.defer).  Copies: chunk0 differs between the two tables.
        lo, hi = self.defer_range
        cp = lambda g: SgdSeg.from_buffer_copy(bytes(g))
        rest = [cp(g) for g in segs if not (lo <= g.offset < hi)]
        late = [cp(g) for g in segs if lo <= g.offset < hi]
        self.n_rest = len(rest)
        self.segs_split_dev = table(rest + late)
        return self.nseg

    def mark_overwritten(self, ptrs):
        """`ptrs`: data pointers of the gradient tensors that this step's grouped weight gradients WROTE (l2s_wgrad_prob.flags = 1, every
        element, every step of this schedule).  Their segments get l2s_sgd_seg.flags = 1: an update that clears the gradients it consumes
        skips them (232 MB of zero stores per step, and the weight-gradient epilogues no longer read dW: 0.46 GB of HBM traffic together).
        The tables are rewritten in place only when the set changes (once per schedule); a tensor that leaves the set is cleared here."""
        key = frozenset(ptrs)
        if key == self._ow_key:
            return
        base = self.grad.data_ptr()
        torch.cuda.synchronize() if self.grad.is_cuda else None
        for arr, dev in zip(self._seg_tables, (self.segs_dev, self.segs_split_dev)):
            for g in arr:
                was, now = g.flags & 1, int(base + 4 * int(g.offset) in key)
                if was and not now:
                    self.grad[int(g.offset):int(g.offset + g.count)].zero_()
                g.flags = now
            dev.copy_(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8))
        torch.cuda.synchronize() if self.grad.is_cuda else None
        self._ow_key = key

    def stale_marked(self, s0, s1, fresh):
        """gradient pointers of the segments [s0, s1) (segs_dev order) that are marked as overwritten but are NOT in `fresh` (the tensors this
        backward pass has written so far): a partial update about to consume them would read last step's values (optim.SGD.partial)"""
        if not self._ow_key:
            return []
        base = self.grad.data_ptr()
        return [base + 4 * o for o in self.seg_offs[s0:s1] if (base + 4 * o) in self._ow_key and (base + 4 * o) not in fresh]

    def chunk_range(self, lo, hi):
        """[chunk_lo, chunk_hi) of the update kernel's work chunks (segs_dev order) that hold elements of [lo, hi) of the flat buffer"""
        import bisect
        s0 = bisect.bisect_right(self.seg_ends, lo)                  # first segment that ends behind lo
        s1 = bisect.bisect_left(self.seg_offs, hi)                   # first segment that starts at or behind hi
        if s0 >= s1:
            return 0, 0
        c_lo = self.seg_chunk0[s0] + max(0, lo - self.seg_offs[s0]) // self.sgd_chunk
        last = s1 - 1
        c_hi = self.seg_chunk0[last] + min(self.seg_chunks[last], -(-(min(hi, self.seg_ends[last]) - self.seg_offs[last]) // self.sgd_chunk))
        return c_lo, c_hi

    def refresh_shadow_full(self):
        """shadow = dtype(rowscale * param) for every trainable tensor (the SGD kernel keeps it current afterwards)."""
        if not hasattr(self, 'segs_dev'):
            self.build_segments()
        zero = torch.zeros_like(self.grad)
        # lr = 0, momentum = 1 (keeps the momentum buffer), wd = 0: a pure shadow rewrite through the same kernel
        O.sgd_momentum(self.param, zero, self.mom, self.segs_dev, self.nseg, self.rowscale, 0.0, 1.0, 0.0, 0.0, shadow=self.shadow)
