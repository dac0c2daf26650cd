"""resnetv1 — the ResNet-101 C4 lang2seg network with 7 spatial dynamic filters and the att2in2
caption-cycle loss, i.e. the reference's nets/resnet_v1_cycle_res5_2.py ("RES") +
nets/network_cycle_res5_2.py ("NET"), on the MI355X kernels.  Constructor, `create_architecture`,
`train_step` and the state-dict key names are the reference's.

Citations: ENC = lib/layers/lang_encoder.py, ATT = lib/caption_models/AttModel.py,
CRIT = lib/misc/utils.py, PL/ATL/PTL = layer_utils/{proposal,anchor_target,proposal_target}_layer.py."""
import os
import numpy as np
import torch

from .. import ops as O
from .._lib import F32, BF16
from ..model.config import cfg
from .network import Network, ConvOp, Bottleneck
from .params import ParamStore
from .variants import solver_cfg
from . import anchors as ANC

f32 = torch.float32


class resnetv1(Network):
    variant = 'cycle'        # nets/variants.py; the sibling modules resnet_v1_{baseline,7f,7f_response,cycle_response}.py subclass this

    def __init__(self, opt, batch_size=1, num_layers=50, variant=None):
        Network.__init__(self, batch_size=batch_size)
        from .variants import VARIANTS
        if variant is not None:
            self.variant = variant
        self.var = VARIANTS[self.variant]
        self._num_layers = num_layers
        self.opt = dict(opt)
        assert self.opt.get('rnn_type', 'lstm') == 'lstm' and self.opt.get('rnn_num_layers', 1) == 1 and self.opt.get('bidirectional', 1) > 0
        if self.var['cap'] is not None:
            assert self.opt.get('caption_model', 'att2in2') == 'att2in2', 'only the att2in2 captioner is on the hot path'
        self._cap_loss_weight = float(self.opt.get('cap_loss_weight', 0.0)) if self.var['cap'] is not None else 0.0   # RES:253
        self._C4_feat_dim = self.opt['C4_feat_dim']

    # ------------------------------------------------------------------ RES:275-337
    def _init_modules(self):
        assert self._num_layers in (50, 101, 152)
        assert 0 <= cfg.RESNET.FIXED_BLOCKS < 4
        fb = cfg.RESNET.FIXED_BLOCKS
        self.P = ParamStore(self.opt, self._num_layers, self._num_classes, self._num_anchors, fb, self.device, self.dt, self.variant)
        P = self.P
        nb = P.nblocks
        self.layers = {}
        inpl = 64
        for li, planes, stride in [(1, 64, 1), (2, 128, 2), (3, 256, 2), (4, 512, 1)]:
            blocks = []
            for b in range(nb[li - 1]):
                need_dx = li > fb and not (li == fb + 1 and b == 0)          # nothing trainable below the first trainable block
                blocks.append(Bottleneck(self, 'resnet.layer%d.%d' % (li, b), inpl, planes, stride if b == 0 else 1, b == 0, need_dx))
                inpl = planes * 4
            self.layers[li] = blocks
        C4 = self._C4_feat_dim
        A, nc = self._num_anchors, self._num_classes
        self.rpn_conv = ConvOp(self, 'rpn_net.weight', C4, 512, 3, 1, 1, bias_key='rpn_net.bias')
        self.rpn_heads = ConvOp(self, None, 512, 6 * A, group=('rpn_head_w', 'rpn_head_b'), Cout_pad=P.rpn_npad)
        self.rcnn_heads = ConvOp(self, None, 2048, 5 * nc, group=('rcnn_w', 'rcnn_b'), Cout_pad=P.rcnn_npad)
        self.mask_pred = ConvOp(self, 'mask_pred_net.weight', 256, nc, bias_key='mask_pred_net.bias', need_dgrad=False)
        if self.var['cap'] is not None:
            self.att_embed = ConvOp(self, 'caption_model.att_embed.0.weight', self.opt['att_feat_size'], self.opt['rnn_size'],
                                    bias_key='caption_model.att_embed.0.bias')
        # number of dynamic-FC outputs: 7 filters + 7 mixing weights, or the single filter of the baseline network
        self._NFP = 7 * C4 + 7 if self.var['nfilt'] == 7 else C4
        self.up_wT = O.empty((4 * 256 * 2048,), self.dt)     # ConvTranspose forward operand [(dy,dx,co)][ci]
        self.base_anchors = torch.from_numpy(ANC.base_anchors(self._anchor_scales, self._anchor_ratios)).to(self.device)
        self.init_weights()
        sc = solver_cfg(self.variant)           # the param groups of this variant's solver (model/train_val.make_optimizer rebuilds them)
        P.build_segments(double_bias=sc.TRAIN.DOUBLE_BIAS, bias_decay=sc.TRAIN.BIAS_DECAY)
        self.load_state_dict(self._initial_state, strict=False)
        del self._initial_state
        # RES:256 -> caption_models.setup(opt): `--start_from` warm-starts the captioner from <dataset_splitBy>/<start_from>/model-best.pth
        # (caption_models/__init__.py:45-51); missing directory / infos-best.pkl / mismatching keys raise, as the reference asserts
        if self.var['cap'] is not None and self.opt.get('start_from') is not None:
            from ..utils.caption_ckpt import load_caption_weights
            load_caption_weights(self, self.opt, root=self.opt.get('checkpoint_root', '.'))

    def init_weights(self):
        """NET:333-355 / RES:135-141 initialisers (host RNG, once, before the first load).
        The reference never runs its ResNet from these values (the trunk always comes from the pretrained detector); a network that IS
        run from them (bench.py, tools without a checkpoint) has frozen identity BatchNorm and pixel-scale inputs, and its activations
        overflow bf16 / fp32 within one step (layer3 output ~1e6, NaN after the first update).  Three scale factors keep the residual
        stream's variance flat through the 33 blocks, as oracle/weights.make_state_dict does for the parity fixtures: the last BN
        gain of a block 0.15, the shortcut BN gain 0.7, conv1 x 0.05."""
        g = torch.Generator().manual_seed(cfg.RNG_SEED)
        sd = {}
        for k, shp in self.P.shapes.items():
            n = int(np.prod(shp))
            if k.endswith('bn3.weight'):
                t = torch.full((n,), 0.15)
            elif k.endswith('downsample.1.weight'):
                t = torch.full((n,), 0.7)
            elif k.endswith('running_var') or (('.bn' in k or 'downsample.1' in k or k.startswith('resnet.bn1')) and k.endswith('weight')):
                t = torch.ones(n)
            elif k.endswith('running_mean') or '.bn' in k or 'downsample.1' in k or k.startswith('resnet.bn1'):
                t = torch.zeros(n)
            elif k.startswith('resnet.') and len(shp) == 4:
                t = torch.randn(n, generator=g) * float(np.sqrt(2.0 / (shp[2] * shp[3] * shp[0])))
                if k == 'resnet.conv1.weight':
                    t = t * 0.05                                      # inputs are pixel-scale (sigma ~50)
            elif k.startswith(('rpn_', 'cls_score', 'mask_')) and k.endswith('weight'):
                t = torch.randn(n, generator=g) * 0.01
            elif k.startswith('bbox_pred_net') and k.endswith('weight'):
                t = torch.randn(n, generator=g) * 0.001
            elif k.startswith(('rpn_', 'cls_score', 'mask_', 'bbox_pred')) and k.endswith('bias'):
                t = torch.zeros(n)
            elif 'embedding.weight' in k or 'embed.0.weight' in k:
                t = torch.randn(n, generator=g)
            else:
                kk = 1.0 / float(np.sqrt(shp[-1] if len(shp) > 1 else 512))
                t = (torch.rand(n, generator=g) * 2 - 1) * kk
            sd[k] = t.view(shp)
        self._initial_state = sd

    def _make_transposes(self):
        """fp32 transposes [K][N] of the skinny (batch-1) matrices: their data gradients dx = dy . W then run through the same
        one-wave-per-output GEMV kernel as the forward pass (no atomics, full-chip parallelism even at K = 512)."""
        class _T(object):
            pass
        P = self.P
        self.wT, self.extra_transposes = {}, []
        def add(name, src, N, K, dst=None, f32=1):
            e = _T(); e.w_master, e.scale, e.Np, e.k, e.Cin, e.force_f32 = src, None, N, 1, K, f32
            e.wb = torch.zeros(K * N, dtype=torch.float32, device=self.device) if dst is None else dst
            self.extra_transposes.append(e)
            if dst is None:
                self.wT[name] = (e.wb, N, K)
        # only the matrices whose data gradient is taken one row at a time (inside a recurrence); row batches use bwd_x's MFMA path
        for sfx in ['', '_reverse']:
            w = 'rnn_encoder.rnn.weight_hh_l0'
            add(w + sfx, P.view(w + sfx), *P.shapes[w + sfx])
        capk = ['caption_model.core.h2h.weight', 'caption_model.core.attention.h2att.weight']
        if not (self.cap_projected and self.opt['rnn_size'] <= 1024):
            capk.append('caption_model.core.a2c.weight')     # (the projected-attention recurrence never multiplies by a2c^T row by row)
        for k in (capk if self.var['cap'] is not None else []):
            add(k, P.view(k), *P.shapes[k])
        # (the dynamic-filter matrix, 7175 x 1024 fp32 = 29 MB, is read as stored by the split NN kernel: no copy)
        # 2x2 deconv: its forward operand [(dy,dx,co)][ci] is the transpose of the master [ci][(dy,dx,co)] (activation dtype)
        add('mask_up', P.view('mask_up_sampling.weight'), 2048, 4 * 256, dst=self.up_wT, f32=0)

    def bwd_x(self, dy, name, dx, M, accumulate=False, lddy=None, mul=None):
        """dx[M][K] (+)= (dy[M][N] . W[N][K]) (* mul).  One row (inside the recurrences): the transposed copy through the
        one-wave-per-output GEMV; row batches: the MFMA kernel on the weight as stored (no copy)."""
        if M == 1 and name in self.wT:
            assert mul is None
            wT, N, K = self.wT[name]
            O.linear_fwd(dy, wT, None, dx, M, K, N, accumulate=accumulate, ldx=N if lddy is None else lddy, ldw=N)
            return
        P = self.P
        w = P.view(name)
        N, K = P.shapes[name]
        nws = O.linear_bwd_x_ws_floats(M, N, K)
        ws = self.buf('bwdx.ws.' + name, (nws,), f32) if nws else None
        O.linear_bwd_x(dy, w, dx, M, N, K, accumulate=accumulate, lddy=lddy, mul=mul, ws=ws)

    def refresh_weights(self, full=False):
        if not hasattr(self, 'extra_transposes'):
            self._make_transposes()
        if full and getattr(self, '_stem_pack', None) is not None:
            # the matrix-core stem's fragment-ordered conv1 weights: repacked IN PLACE with the other frozen weights, so that a state dict
            # loaded after launch tapes were recorded reaches the replayed steps too (the tapes hold this buffer's address)
            w1 = self.P.frozen['resnet.conv1.weight']
            self._stem_pack, self._stem_pack_ver = O.stem_pack(w1, self._stem_pack), (w1.data_ptr(), w1._version)
        Network.refresh_weights(self, full)

    # ------------------------------------------------------------------ helpers
    def _drop(self, name, shape, p):
        """dropout mask (scaled) or None.  Production: counter-hash RNG on device; parity: injected masks."""
        if not self.training or p <= 0.0:
            return None
        if self.parity is not None:
            m = (self.parity.get('drops') or {}).get(name)
            return m
        m = self.buf('drop.' + name, shape, f32)
        O.dropout_mask(m, p, self.seed_counter(), 1000003 * sum(map(ord, name)) + getattr(self, 'rank_seed', 0))
        return m

    def _keys(self, name, n):
        if self.parity is not None and self.parity.get(name) is not None:
            return self.parity[name]
        k = self.buf('keys.' + name, (n,), torch.int32)
        O.random_keys(k, self.seed_counter(), 7777 + 1000003 * sum(map(ord, name)) + getattr(self, 'rank_seed', 0))
        return k

    # ------------------------------------------------------------------ language encoder (ENC:27-82)
    # hidden/cell states are kept in (T+1)-row arrays with one all-zero row so that "previous state" is always a row of
    # the same array: forward direction h(t) = row t+1, h(t-1) = row t; reverse direction h(t) = row t, previous = row t+1.
    def _encoder_fwd(self, d):
        P, T = self.P, d['T']
        Hh, E = self.opt['rnn_hidden_size'], self.opt['word_embedding_size']
        t = self.t
        emb = self.buf('enc.emb', (T, E), f32)
        t['enc.drop'] = self._drop('word', (T, E), self.opt['word_drop_out'])
        O.embed_fwd(P.view('rnn_encoder.embedding.weight'), d['labels'], t['enc.drop'], emb, T, E, False)
        x = self.buf('enc.x', (T, Hh), f32)
        O.linear_fwd(emb, P.view('rnn_encoder.mlp.0.weight'), P.view('rnn_encoder.mlp.0.bias'), x, T, Hh, E, act=1)
        hidden = self.buf('enc.hidden', (2 * Hh,), f32)
        st = []
        for di, sfx in enumerate(['', '_reverse']):
            g = self.buf('enc.gates' + sfx, (T, 4 * Hh), f32)
            O.linear_fwd(x, P.view('rnn_encoder.rnn.weight_ih_l0' + sfx), P.view('rnn_encoder.rnn.bias_ih_l0' + sfx), g, T, 4 * Hh, Hh)
            st.append(dict(g=g, hs=self.buf('enc.hfull' + sfx, (T + 1, Hh), f32), cs=self.buf('enc.cfull' + sfx, (T + 1, Hh), f32),
                           act=self.buf('enc.act' + sfx, (T, 4 * Hh), f32), whh=P.view('rnn_encoder.rnn.weight_hh_l0' + sfx),
                           bhh=P.view('rnn_encoder.rnn.bias_hh_l0' + sfx)))
        # one launch per time step for both directions (h2h GEMV + cell update fused): T launches instead of 4 T
        for s_ in range(T):
            dirs = []
            for di, q in enumerate(st):
                tt = s_ if di == 0 else T - 1 - s_
                cur, prev = (tt + 1, tt) if di == 0 else (tt, tt + 1)
                dirs.append(dict(w_hh=q['whh'], b_hh=q['bhh'], gates_in=q['g'][tt], h_prev=q['hs'][prev], c_prev=q['cs'][prev],
                                 c=q['cs'][cur], h=q['hs'][cur], act=q['act'][tt], gates_out=q['g'][tt]))
            O.lstm_step_fwd(dirs, Hh)
        O.memcpy(hidden[0:Hh], st[0]['hs'][T]); O.memcpy(hidden[Hh:2 * Hh], st[1]['hs'][0])      # ENC:76-80
        t['enc.emb'], t['enc.x'], t['hidden'] = emb, x, hidden
        return hidden

    def _encoder_bwd(self, d, dhidden):
        P, T, t = self.P, d['T'], self.t
        Hh, E = self.opt['rnn_hidden_size'], self.opt['word_embedding_size']
        dx = self.buf('enc.dx', (T, Hh), f32, zero=True)
        st = []
        for di, sfx in enumerate(['', '_reverse']):
            st.append(dict(hs=self.buf('enc.hfull' + sfx, (T + 1, Hh), f32), cs=self.buf('enc.cfull' + sfx, (T + 1, Hh), f32),
                           act=self.buf('enc.act' + sfx, (T, 4 * Hh), f32), dg=self.buf('enc.dg' + sfx, (T, 4 * Hh), f32),
                           dc=self.buf('enc.dc' + sfx, (2, Hh), f32, zero=True), wT=self.wT['rnn_encoder.rnn.weight_hh_l0' + sfx][0], sfx=sfx))
        # one launch per time step for both directions: dh = W_hh^T dg(next step) fused with this step's cell backward
        k = 0
        for s_ in range(T):
            dirs = []
            for di, q in enumerate(st):
                tt = T - 1 - s_ if di == 0 else s_
                nxt = tt + 1 if di == 0 else tt - 1               # the step processed just before this one
                cur, prev = (tt + 1, tt) if di == 0 else (tt, tt + 1)
                dirs.append(dict(w_hh_T=q['wT'], dgates_next=(q['dg'][nxt] if s_ > 0 else None),
                                 dh_ext=(dhidden[di * Hh:(di + 1) * Hh] if s_ == 0 else None), dc_in=q['dc'][k], act=q['act'][tt],
                                 c_prev=q['cs'][prev], c=q['cs'][cur], dgates=q['dg'][tt], dc_prev=q['dc'][1 - k]))
            O.lstm_step_bwd(dirs, Hh)
            k = 1 - k
        for di, q in enumerate(st):
            sfx, dg, hs = q['sfx'], q['dg'], q['hs']
            hprev = hs[0:T] if di == 0 else hs[1:T + 1]
            O.linear_bwd_w(dg, hprev, P.view('rnn_encoder.rnn.weight_hh_l0' + sfx, P.grad), P.view('rnn_encoder.rnn.bias_hh_l0' + sfx, P.grad), T, 4 * Hh, Hh)
            O.linear_bwd_w(dg, t['enc.x'], P.view('rnn_encoder.rnn.weight_ih_l0' + sfx, P.grad), P.view('rnn_encoder.rnn.bias_ih_l0' + sfx, P.grad), T, 4 * Hh, Hh)
            self.bwd_x(dg, 'rnn_encoder.rnn.weight_ih_l0' + sfx, dx, T, accumulate=True)
        O.act_bwd(dx, t['enc.x'], 1)
        O.linear_bwd_w(dx, t['enc.emb'], P.view('rnn_encoder.mlp.0.weight', P.grad), P.view('rnn_encoder.mlp.0.bias', P.grad), T, Hh, E)
        demb = self.buf('enc.demb', (T, E), f32)
        self.bwd_x(dx, 'rnn_encoder.mlp.0.weight', demb, T)
        O.embed_bwd(demb, t['enc.emb'], d['labels'], t['enc.drop'], P.view('rnn_encoder.embedding.weight', P.grad), T, E, False)

    # ------------------------------------------------------------------ att2in2 captioner (ATT:60-101,406-466; CRIT:43-53)
    # states live in (S+1)-row arrays, row 0 = zeros: h(i) = row i+1, h(i-1) = row i.
    def _caption_pre(self, d):
        """token-only part of _caption_fwd / _caption_bwd, issued early on the language stream"""
        P, S = self.P, d['S']
        R, IE, AH, L = self.opt['rnn_size'], self.opt['input_encoding_size'], self.opt['att_hid_size'], 196
        pv = lambda k: P.view('caption_model.' + k)
        pre = {'drop_att': self._drop('att', (L, R), self.opt['drop_prob_lm']), 'drop_xt': self._drop('xt', (S, IE), self.opt['drop_prob_lm']),
               'drop_out': self._drop('out', (S, R), self.opt['drop_prob_lm'])}
        xt = self.buf('cap.xt', (S, IE), f32)
        O.embed_fwd(pv('embed.0.weight'), d['cap_in'], pre['drop_xt'], xt, S, IE, True)
        sums = self.buf('cap.sums', (S, 5 * R), f32)
        O.linear_fwd(xt, pv('core.i2h.weight'), pv('core.i2h.bias'), sums, S, 5 * R, IE)
        proj = self.cap_projected and R <= 1024
        nz = L * AH + L * R + 4 * R
        self.buf('cap.bwd_zero', (nz + (L * 2 * R if proj else 0),), f32, zero=True)
        pre.update(xt=xt, sums=sums, zeroed=True)
        return pre

    def _caption_fwd(self, d, att_feats, loss):
        P, t, S = self.P, self.t, d['S']
        pre = getattr(self, '_cap_pre', None)
        R, IE, AH, L = self.opt['rnn_size'], self.opt['input_encoding_size'], self.opt['att_hid_size'], 196
        V1 = self.opt['vocab_size'] + 1
        SC = 256                                             # per-row scratch tail of the attention kernels
        pv = lambda k: P.view('caption_model.' + k)
        a = self.buf('cap.a', (L, R), f32)
        self.att_embed.fwd(att_feats, L, 1, 1, a, relu=True, out_f32=True)
        t['cap.a_pre'] = a
        dm = pre['drop_att'] if pre else self._drop('att', (L, R), self.opt['drop_prob_lm'])
        t['cap.drop_att'] = dm
        if dm is not None:
            ad = self.buf('cap.ad', (L, R), f32); O.mul(a, dm, ad)
        else:
            ad = a
        patt = self.buf('cap.patt', (L, AH), f32)
        O.linear_fwd(ad, pv('ctx2att.weight'), pv('ctx2att.bias'), patt, L, AH, R)
        if pre:
            xt, sums, t['cap.drop_xt'] = pre['xt'], pre['sums'], pre['drop_xt']
        else:
            xt = self.buf('cap.xt', (S, IE), f32)
            t['cap.drop_xt'] = self._drop('xt', (S, IE), self.opt['drop_prob_lm'])
            O.embed_fwd(pv('embed.0.weight'), d['cap_in'], t['cap.drop_xt'], xt, S, IE, True)
            sums = self.buf('cap.sums', (S, 5 * R), f32)
            O.linear_fwd(xt, pv('core.i2h.weight'), pv('core.i2h.bias'), sums, S, 5 * R, IE)
        hs = self.buf('cap.hfull', (S + 1, R), f32); cs = self.buf('cap.cfull', (S + 1, R), f32); save = self.buf('cap.save', (S, 6 * R), f32)
        att_h = self.buf('cap.att_h', (S, AH), f32); tanh_ws = self.buf('cap.tanh', (S, L, AH), f32)
        wgt = self.buf('cap.wgt', (S, L), f32); ares = self.buf('cap.ares', (S, R + SC), f32); a2c = self.buf('cap.a2c', (S, 2 * R), f32)
        proj = self.cap_projected and R <= 1024
        if proj:
            # projected attention (csrc/lang.hip): P = att . W_a2c^T once, then 3 launches per token
            Pm = self.buf('cap.P', (L, 2 * R), f32); dots = self.buf('cap.dots', (S, 256), f32)
            O.linear_fwd(ad, pv('core.a2c.weight'), None, Pm, L, 2 * R, R)
            t['cap.P'] = Pm
        t['cap.resident'] = resident = proj and self.cap_persistent and O.cap_recur_supported(S, R, AH, L)
        if resident:
            # the whole recurrence as one resident launch (csrc/cap_recur.hip): 32 workgroups own 16 units each, three granule exchanges per token
            if getattr(self, '_cap_state', None) is None:
                self._cap_state = (O.cap_recur_state(False), O.cap_recur_state(True))
            O.cap_recur_fwd(pv('core.h2h.weight'), pv('core.h2h.bias'), pv('core.attention.h2att.weight'), pv('core.attention.h2att.bias'), patt,
                            pv('core.attention.alpha_net.weight'), pv('core.attention.alpha_net.bias'), Pm, pv('core.a2c.bias'), sums, hs, cs, save,
                            tanh_ws, wgt, self._cap_state[0], S, R, AH, L)
        for i in range(0 if resident else S):
            # h2att(h) and h2h(h) (+= i2h sums) in one launch; attention; a2c Linear fused with the gates (4 launches per step)
            O.linear2_fwd(hs[i], R, pv('core.attention.h2att.weight'), pv('core.attention.h2att.bias'), att_h[i], AH, False,
                          pv('core.h2h.weight'), pv('core.h2h.bias'), sums[i], 5 * R, True)
            if proj:
                O.cap_att_dots_fwd(patt, att_h[i], pv('core.attention.alpha_net.weight'), pv('core.attention.alpha_net.bias'), L, AH, tanh_ws[i], dots[i])
                O.cap_apply_gates_fwd(Pm, dots[i], pv('core.a2c.bias'), sums[i], cs[i], cs[i + 1], hs[i + 1], save[i], wgt[i], L, R)
                continue
            O.cap_attention_fwd(patt, ad, att_h[i], pv('core.attention.alpha_net.weight'), pv('core.attention.alpha_net.bias'), L, AH,
                                tanh_ws[i], wgt[i], ares[i])
            O.cap_a2c_gates_fwd(ares[i], pv('core.a2c.weight'), pv('core.a2c.bias'), R, sums[i], cs[i], cs[i + 1], hs[i + 1], save[i], R)
        t['cap.drop_out'] = pre['drop_out'] if pre else self._drop('out', (S, R), self.opt['drop_prob_lm'])
        if t['cap.drop_out'] is not None:
            ho = self.buf('cap.ho', (S, R), f32); O.mul(hs[1:], t['cap.drop_out'], ho)
        else:
            ho = hs[1:]
        logits = self.buf('cap.logits', (S, V1), f32)
        O.linear_fwd(ho, pv('logit.weight'), pv('logit.bias'), logits, S, V1, R)
        dlogits = self.buf('cap.dlogits', (S, V1), f32)
        lp = self.buf('cap.logp', (S, V1), f32) if self.keep_logprobs else None
        O.logsoftmax_nll(logits, d['cap_tgt'], d['cap_mask'], S, V1, self._cap_loss_weight, loss[5:6], dlogits, lp)
        t.update({'cap.ad': ad, 'cap.patt': patt, 'cap.xt': xt, 'cap.ho': ho, 'cap.dlogits': dlogits, 'cap.logp': lp})

    def _caption_bwd(self, d, att_feats):
        """returns d(att_feats) in the activation dtype [196][att_feat_size]."""
        P, t, S = self.P, self.t, d['S']
        R, IE, AH, L = self.opt['rnn_size'], self.opt['input_encoding_size'], self.opt['att_hid_size'], 196
        V1 = self.opt['vocab_size'] + 1
        SC = 256
        pv = lambda k: P.view('caption_model.' + k)
        gv = lambda k: P.view('caption_model.' + k, P.grad)
        hs = self.buf('cap.hfull', (S + 1, R), f32); cs = self.buf('cap.cfull', (S + 1, R), f32); save = self.buf('cap.save', (S, 6 * R), f32)
        tanh_ws = self.buf('cap.tanh', (S, L, AH), f32); wgt = self.buf('cap.wgt', (S, L), f32)
        ares = self.buf('cap.ares', (S, R + SC), f32)
        dlogits, ad = t['cap.dlogits'], t['cap.ad']
        # Only d(att_feats) is on the step's critical path (the main queue waits for it at the caption join): the parameter
        # gradients that nothing downstream reads are collected in `later` and issued after the branch has produced its result
        # (caption_branch: on the language stream, behind the join point).
        later = self._cap_deferred = []
        later.append(lambda: O.linear_bwd_w(dlogits, t['cap.ho'], gv('logit.weight'), gv('logit.bias'), S, V1, R))
        dho = self.buf('cap.dho', (S, R), f32)
        self.bwd_x(dlogits, 'caption_model.logit.weight', dho, S, mul=t['cap.drop_out'])
        dsums = self.buf('cap.dsums', (S, 5 * R), f32); da2c = self.buf('cap.da2c', (S, 2 * R), f32)
        datt_h = self.buf('cap.datt_h', (S, AH + SC), f32); dares = self.buf('cap.dares', (S, R), f32); ddot = self.buf('cap.ddot', (S, L), f32)
        proj = self.cap_projected and R <= 1024
        # dpatt | dad | dh | dc (| dP): one buffer, one clear
        nz = L * AH + L * R + 4 * R
        zb = self.buf('cap.bwd_zero', (nz + (L * 2 * R if proj else 0),), f32, zero=not (getattr(self, '_cap_pre', None) or {}).get('zeroed'))
        dpatt = zb[:L * AH].view(L, AH); dad = zb[L * AH:L * AH + L * R].view(L, R)
        dh = zb[L * AH + L * R:L * AH + L * R + 2 * R].view(2, R); dc = zb[L * AH + L * R + 2 * R:nz].view(2, R)
        wT_h2h = self.wT['caption_model.core.h2h.weight'][0]; wT_h2att = self.wT['caption_model.core.attention.h2att.weight'][0]
        k = 0
        if proj:
            Pm = t['cap.P']; dwl = self.buf('cap.dwl', (S, 256), f32); dP = zb[nz:].view(L, 2 * R)
        if t.get('cap.resident'):
            O.cap_recur_bwd(pv('core.h2h.weight'), pv('core.attention.h2att.weight'), Pm, pv('core.attention.alpha_net.weight'), save, cs, wgt, tanh_ws, dho,
                            dsums, da2c, ddot, datt_h, self._cap_state[1], S, R, AH, L)
        for i in range(-1 if t.get('cap.resident') else S - 1, -1, -1):
            if proj:
                # 3 launches per step: gates + d(weight) = P . d(a2c); softmax backward + datt_h; dh(i-1)
                O.cap_gates_bwd_dw(dh[k], dc[k], save[i], cs[i], Pm, dsums[i], da2c[i], dc[1 - k], dwl[i], L, R, dh2=dho[i])
                O.cap_attention_bwd_step2(dwl[i], tanh_ws[i], wgt[i], pv('core.attention.alpha_net.weight'), L, AH, ddot[i], datt_h[i])
            else:
                # 4 launches per step: gates (dh = recurrent part + this step's output gradient), a2c^T, the attention pieces the
                # recurrence needs, and dh(i-1) = W_h2h^T dsums + W_h2att^T datt_h
                O.cap_gates_bwd(dh[k], dc[k], save[i], cs[i], dsums[i], da2c[i], dc[1 - k], R, dh2=dho[i])
                self.bwd_x(da2c[i], 'caption_model.core.a2c.weight', dares[i], 1)
                O.cap_attention_bwd_step(dares[i], ad, tanh_ws[i], wgt[i], pv('core.attention.alpha_net.weight'), L, AH, ddot[i], datt_h[i])
            O.linear_sum2_fwd(dsums[i], wT_h2h, 5 * R, datt_h[i], wT_h2att, AH, dh[1 - k], R)
            k = 1 - k
        O.cap_attention_bwd_batched(ddot, wgt, None if proj else dares, R, tanh_ws, pv('core.attention.alpha_net.weight'), S, L, AH, dpatt, dad,
                                    gv('core.attention.alpha_net.weight'), gv('core.attention.alpha_net.bias'))
        if proj:
            # d(P) = sum_t weight_t (x) d(a2c)_t  [L][2R];  d(att) += d(P) . W_a2c
            O.linear_bwd_w(wgt, da2c, dP, None, S, L, 2 * R)
            self.bwd_x(dP, 'caption_model.core.a2c.weight', dad, L, accumulate=True)
        hprev = hs[0:S]

        def recurrence_grads():
            if proj:
                O.linear_bwd_w(dP, ad, gv('core.a2c.weight'), gv('core.a2c.bias'), L, 2 * R, R)
            else:
                O.linear_bwd_w(da2c, ares, gv('core.a2c.weight'), gv('core.a2c.bias'), S, 2 * R, R, ldx=R + SC)
            O.linear_bwd_w(dsums, hprev, gv('core.h2h.weight'), gv('core.h2h.bias'), S, 5 * R, R)
            O.linear_bwd_w(datt_h, hprev, gv('core.attention.h2att.weight'), gv('core.attention.h2att.bias'), S, AH, R, lddy=AH + SC)
            O.linear_bwd_w(dsums, t['cap.xt'], gv('core.i2h.weight'), gv('core.i2h.bias'), S, 5 * R, IE)
            dxt = self.buf('cap.dxt', (S, IE), f32)
            self.bwd_x(dsums, 'caption_model.core.i2h.weight', dxt, S)
            O.embed_bwd(dxt, t['cap.xt'], d['cap_in'], t['cap.drop_xt'], gv('embed.0.weight'), S, IE, True)
            O.linear_bwd_w(dpatt, ad, gv('ctx2att.weight'), gv('ctx2att.bias'), L, AH, R)
        later.append(recurrence_grads)
        # ctx2att
        self.bwd_x(dpatt, 'caption_model.ctx2att.weight', dad, L, accumulate=True)
        dadT = self.buf('cap.dadT', (L, R))
        # dropout mask (it multiplies the accumulated sum, not only this term), ReLU backward and the cast to the activation dtype: one launch
        O.mask_relu_cast(dad, t['cap.drop_att'], t['cap.a_pre'], dadT)
        self.att_embed.wgrad(dadT, att_feats, L, 1, 1)
        datt = self.buf('cap.datt_feats', (L, self.opt['att_feat_size']))
        self.att_embed.dgrad(dadT, L, 1, 1, datt)
        return datt

    # ------------------------------------------------------------------ backbone / RoI-head hooks (overridden by the VGG variant)
    def _backbone_fwd(self, d, saved):
        """conv1/bn1/relu/maxpool/layer1-3 (RES:261-265,309-310) -> (C4 map [H*W][1024], Hc, Wc)."""
        P = self.P
        H, W = int(d['data'].shape[1]), int(d['data'].shape[2])
        OH1, OW1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        h, w = (OH1 + 2 - 3) // 2 + 1, (OW1 + 2 - 3) // 2 + 1
        x = self.buf('stem.pool', (h * w, 64))
        w1 = P.frozen['resnet.conv1.weight']
        if self.dt == BF16 and self.stem_mfma:
            # conv + affine + ReLU + pooling in one launch on the matrix cores (stem_mfma.hip); the weights in fragment order, repacked when they change
            if getattr(self, '_stem_pack_ver', None) != (w1.data_ptr(), w1._version):
                self._stem_pack, self._stem_pack_ver = O.stem_pack(w1, getattr(self, '_stem_pack', None)), (w1.data_ptr(), w1._version)
            O.stem_pool_bf16(d['data'], self._stem_pack, P.bn_scale['resnet.conv1.weight'], P.bn_bias['resnet.conv1.weight'], x, H, W, OH1, OW1, h, w)
        else:
            c1 = self.buf('stem.c1', (OH1 * OW1, 64))
            O.stem_conv(d['data'], w1, P.bn_scale['resnet.conv1.weight'], P.bn_bias['resnet.conv1.weight'], c1, H, W, OH1, OW1)
            O.maxpool(c1, x, OH1, OW1, 64, h, w)
        self._mark('stem')
        first = 1 if self.join_before_layer1 else cfg.RESNET.FIXED_BLOCKS + 1        # first trainable layer
        # The previous step's tail (its last weight gradients, the update, the transposes) runs beside this step's frozen prefix, and the last
        # weight gradients READ the activation buffers this pass is about to overwrite: layer2[0]'s read the output of layer1, layer3[0]'s the
        # output of layer2.  So the join sits in front of the launch that writes layer1's output (a latent race until round 5, when the fused
        # layer1 became fast enough to lose it): with the split update it waits for [layer2's weight gradients + update] and for the
        # weight-gradient streams' launches only (join_update(layer2_only=True)); layer3 then joins the rest (partial updates, transposes).
        def join_tail():
            self.join_update(full=False, layer2_only=(first == 2))
            self._mark('layer1 done, update joined')
        for li in (1, 2, 3):
            if li == first and not (first == 2 and cfg.RESNET.FIXED_BLOCKS == 1):
                join_tail()
            if li == 3 and self.update_split and first == 2:
                self.join_update(full=False)                   # everything else of the previous step's tail (layer3's partial update, the transposes)
            before_last = join_tail if (li == 1 and first == 2) else None
            if li == 1 and self._layer1_is_fusable():
                x = self._layer1_fused_fwd(x, h, w, before_last)
                continue
            for b, blk in enumerate(self.layers[li]):
                if before_last is not None and b == len(self.layers[li]) - 1:
                    before_last()
                if li == 2 and first == 2 and b == len(self.layers[li]) - 1:
                    self.join_backlog()                            # layer3[0]'s weight gradients of the previous step read this block's output buffer
                x, h, w, sv = blk.fwd(x, 1, h, w, 'l%d.%d' % (li, b))
                saved[(li, b)] = sv
        if cfg.RESNET.FIXED_BLOCKS >= 3:
            self.join_update(full=False)
        return x, h, w

    def _layer1_is_fusable(self):
        """the frozen layer1 (RES:291-299) in bf16: every block behind its conv1 as one launch (csrc/bottleneck_fused.hip)"""
        blks = self.layers[1]
        return (self.layer1_fused and self.dt == BF16 and cfg.RESNET.FIXED_BLOCKS >= 1 and blks[0].planes == 64 and blks[0].stride == 1
                and blks[0].down is not None and all(not c.trainable for b in blks for c in (b.c1, b.c2, b.c3)))

    def _layer1_fused_fwd(self, x, h, w, before_last=None):
        """4 launches instead of 10: conv1 of the first block, then per block [3x3 -> 1x1 + shortcut (+ the next block's conv1)]; nothing of a
        frozen block is kept for backward"""
        blks = self.layers[1]
        a = self.buf('l1f.a0', (h * w, 64))
        blks[0].c1.fwd(x, 1, h, w, a, relu=True)
        for b, blk in enumerate(blks):
            nxt = blks[b + 1] if b + 1 < len(blks) else None
            if nxt is None and before_last is not None:
                before_last()                                  # the previous step's last weight gradients still read this block's output buffer
            y = self.buf('l1f.y%d' % b, (h * w, 256))
            an = self.buf('l1f.a%d' % (b + 1), (h * w, 64)) if nxt is not None else None
            O.bottleneck64_fwd(a, x, blk.c2.wf, blk.c2.bias, blk.c3.wf, blk.c3.bias, y, h, w,
                               wd=blk.down.wf if b == 0 else None, bd=blk.down.bias if b == 0 else None,
                               w1n=nxt.c1.wf if nxt is not None else None, b1n=nxt.c1.bias if nxt is not None else None, a_next=an)
            x, a = y, an
        return x

    def _backbone_bwd(self, dbase, saved, S, main, dp):
        """layer3, layer2 (layer1 and the stem are frozen: RES:290-299)"""
        g = dbase
        fb = cfg.RESNET.FIXED_BLOCKS
        for li in (3, 2, 1):
            if li <= fb:
                break
            for b in reversed(range(len(self.layers[li]))):
                g = self.layers[li][b].bwd(g, saved[(li, b)], 'l%d.%d' % (li, b), x_is_relu_out=True)
                if li == 3 and b in (16, 8) and len(self.layers[3]) > 16:
                    self.flush_wgrads('layer3:%d' % b)         # the weight gradients of the finished third of layer3, one grouped launch
                    if dp is not None:
                        self.dp_ready('layer3:%d' % b)         # ... and its gradients to the reducer
            self.flush_wgrads('layer%d' % li)
            if dp is not None and li == 3:
                self.dp_ready('layer3')                        # everything except layer2 is final (the reducer's stream waits for
                                                               # the language / weight-gradient streams itself)

    # ------------------------------------------------------------------ RoI feature extraction (NET:566-574, 607-615)
    def _crop_max_pool(self):
        return bool(cfg.RESNET.MAX_POOL)                 # RES:252-253 (the VGG network keeps Network._crop_pool_layer's default, True)

    def _pool_mode(self):
        """(kind, crop size, 2x2 max pool after the crop) for cfg.POOLING_MODE / POOLING_ALIGN / RESNET.MAX_POOL"""
        PS = int(cfg.POOLING_SIZE)
        if cfg.POOLING_MODE != 'crop':
            return 'pool', PS, False                     # NET:104-105 RoIPoolFunction
        if cfg.POOLING_ALIGN:
            return 'align', 2 * PS, True                 # NET:569-570: _crop_pool_layer_align(net_conv, rois, im_info), max_pool defaults to True
        mp = self._crop_max_pool()
        return 'crop', (2 * PS if mp else PS), mp        # NET:107-149

    def _rois_pool_fwd(self, net_conv, Hc, Wc, rois, R, saved):
        C4, PS = self._C4_feat_dim, int(cfg.POOLING_SIZE)
        kind, CS, mp = self._pool_mode()
        pool5 = self.buf('roi.pool5', (R * PS * PS, C4))
        if kind == 'pool':
            saved['roi_argmax'] = self.buf('roi.argmax', (R * PS * PS, C4), torch.int32)
            O.roipool_fwd(net_conv, Hc, Wc, C4, rois, R, PS, 1.0 / 16.0, pool5, saved['roi_argmax'])
            return pool5
        crop = self.buf('roi.crop', (R * CS * CS, C4)) if mp else pool5
        if kind == 'align':
            O.cropalign_fwd(net_conv, Hc, Wc, C4, rois, R, CS, self._im_hw[0], self._im_hw[1], crop)
        else:
            O.roialign_fwd(net_conv, Hc, Wc, C4, rois, R, CS, 1.0 / 16.0, crop)
        if mp:
            O.maxpool2x2_fwd(crop, pool5, R, CS, CS, C4)
            saved['roi_crop'] = crop
        return pool5

    def _rois_pool_bwd(self, g, Hc, Wc, rois, R, saved):
        """g = d(pool5) [R*PS*PS][C4] -> d(net_conv) float [H*W][C4]"""
        C4, PS = self._C4_feat_dim, int(cfg.POOLING_SIZE)
        kind, CS, mp = self._pool_mode()
        d_nc_roi = self.buf('roi.dfeat', (Hc * Wc, C4), f32, zero=(kind == 'pool'))    # (the RoIAlign backward writes every element itself, or clears first)
        if kind == 'pool':
            O.roipool_bwd(g, saved['roi_argmax'], R, PS, C4, d_nc_roi)
            return d_nc_roi
        if mp:
            dcrop = self.buf('roi.dcrop', (R * CS * CS, C4))
            O.maxpool2x2_bwd(g, saved['roi_crop'], dcrop, R, CS, CS, C4, False)
            g = dcrop
        if kind == 'align':
            O.cropalign_bwd(g, Hc, Wc, C4, rois, R, CS, self._im_hw[0], self._im_hw[1], d_nc_roi)
        else:
            O.roialign_bwd(g, Hc, Wc, C4, rois, R, CS, 1.0 / 16.0, d_nc_roi)
        return d_nc_roi

    def _roi_head_fwd(self, net_conv, Hc, Wc, rois, R, FGM, saved):
        """RoI head (NET:572-586): crop-pool -> layer4 -> average -> (cls | bbox) heads, mask head on the first FGM RoI slots.
        Returns (heads [R][NPC] f32, NPC, mask scores or None)."""
        P, dt, t = self.P, self.dt, self.t
        C4, nc = self._C4_feat_dim, self._num_classes
        PS, MS = int(cfg.POOLING_SIZE), int(cfg.MASK_SIZE)
        blk0 = self.layers[4][0]
        kind, CS, mp = self._pool_mode()
        fused = (self.fuse_roialign and dt == BF16 and kind == 'crop' and not mp and blk0.down is not None and blk0.stride == 1 and
                 O.roialign_block0_ok(C4, PS, blk0.planes, blk0.planes * 4))
        if fused:
            # crop-and-resize + layer4[0].conv1 + layer4[0].downsample in one launch, one workgroup per RoI; the crop is still written
            # (once) for the weight gradients and the backward pass
            pool5 = self.buf('roi.pool5', (R * PS * PS, C4))
            a1 = self.buf('l4r.0.a1', (R * PS * PS, blk0.planes)); sc = self.buf('l4r.0.sc', (R * PS * PS, blk0.planes * 4))
            O.roialign_block0_fwd(net_conv, Hc, Wc, C4, rois, R, PS, 1.0 / 16.0, blk0.c1.wf, blk0.c1.bias, blk0.planes,
                                  blk0.down.wf, blk0.down.bias, blk0.planes * 4, pool5, a1, sc)
            x, hh, ww, sv = blk0.fwd_rest(pool5, a1, sc, R, PS, PS, 'l4r.0')
            saved[('4r', 0)] = sv
        else:
            pool5 = self._rois_pool_fwd(net_conv, Hc, Wc, rois, R, saved)
            x, hh, ww = pool5, PS, PS
        for b, blk in enumerate(self.layers[4]):
            if fused and b == 0:
                continue
            x, hh, ww, sv = blk.fwd(x, R, hh, ww, 'l4r.%d' % b)
            saved[('4r', b)] = sv
        fc7s = x
        fc7 = self.buf('roi.fc7', (R, 2048))
        O.avgpool_fwd(fc7s, fc7, R, PS * PS, 2048)
        NPC = P.rcnn_npad
        cheads = self.buf('roi.heads', (R, NPC), f32)
        self.rcnn_heads.fwd(fc7, R, 1, 1, cheads, out_f32=True)
        up = self.buf('mask.up', (FGM * MS * MS, 256))
        O.conv_igemm(fc7s, self.up_wT, up, FGM, PS, PS, 2048, PS, PS, 4 * 256, bias=P.view('mask_up_sampling.bias'), relu=True, deconv=True, dt=dt)
        mscore = self.buf('mask.score', (FGM * MS * MS, nc), f32)
        self.mask_pred.fwd(up, FGM, MS, MS, mscore, out_f32=True)
        t.update({'pool5': pool5, 'spatial_fc7': fc7s, 'rcnn_heads': cheads, 'mask_score': mscore})
        saved['roi'] = (fc7s, fc7, up)
        return cheads, NPC, mscore

    def _roi_head_bwd(self, d_cheads, dscore, labels, counts, rois, Hc, Wc, R, FGM, saved):
        """adjoint of _roi_head_fwd -> d(net_conv) [H*W][C4] f32 (RoIAlign scatter)."""
        P, dt = self.P, self.dt
        C4 = self._C4_feat_dim
        PS, MS = int(cfg.POOLING_SIZE), int(cfg.MASK_SIZE)
        fc7s, fc7, up = saved['roi']
        HW = Hc * Wc
        # rcnn heads -> fc7 -> spatial_fc7
        self.rcnn_heads.wgrad(d_cheads, fc7, R, 1, 1)
        dfc7 = self.buf('roi.dfc7', (R, 2048))
        self.rcnn_heads.dgrad(d_cheads, R, 1, 1, dfc7)
        # mask head
        dup = self.buf('mask.dup', (FGM * MS * MS, 256))
        # (the data gradient on the main path; the ordered reduction of its partial sums into dW / db - 27 us that nothing of the chain waits for - on a
        # weight-gradient stream with the deconvolution's bias gradient)
        mws = self.buf('mask.pred_ws', (FGM * 14 * 257,), f32)
        O.maskpred_bwd_dx(dscore, labels, counts, FGM, MS * MS, 256, P.view('mask_pred_net.weight'), up, up, dup, mws)
        with self.fork_wgrad():
            O.maskpred_bwd_reduce(mws, labels, counts, FGM, 256, P.view('mask_pred_net.weight', P.grad), P.view('mask_pred_net.bias', P.grad))
            O.colsum(dup, FGM * MS * MS, 256, 256, P.view('mask_up_sampling.bias', P.grad), ws=self.buf('mask.up_bias_ws', (32 * 256,), f32))
        # the 2x2 stride-2 transposed convolution's weight gradient = a convolution weight gradient with the roles of input and output swapped
        self.wgq.add(P.view('mask_up_sampling.weight', P.grad), fc7s, dup, FGM, MS, MS, 256, PS, PS, 2048, 2, 2, 0)
        dmask_fc7 = self.buf('mask.dfc7s', (FGM * PS * PS, 2048))
        O.conv_igemm(dup, P.view('mask_up_sampling.weight', P.shadow), dmask_fc7, FGM, MS, MS, 256, PS, PS, 2048, 2, 2, 2, 0, dt=dt)
        g = self.buf('l4r.g', (R * PS * PS, 2048))
        O.avgpool_bwd(dfc7, g, dmask_fc7, fc7s, FGM, PS * PS, 2048)
        if R > FGM:
            off = FGM * PS * PS
            O.avgpool_bwd(dfc7[FGM:], g[off:], None, fc7s[off:], R - FGM, PS * PS, 2048)
        for b in reversed(range(len(self.layers[4]))):
            g = self.layers[4][b].bwd(g, saved[('4r', b)], 'l4r.%d' % b, x_is_relu_out=(b > 0))
        self._mark('roi head bwd')
        return self._rois_pool_bwd(g, Hc, Wc, rois, R, saved)

    # ------------------------------------------------------------------ the step
    keep_logprobs = False

    def forward_backward(self, d, backward=True):
        P, dt = self.P, self.dt
        self.t = t = {}
        TR = cfg.TRAIN
        A, nc = self._num_anchors, self._num_classes
        R = int(TR.BATCH_SIZE); FGS = int(round(TR.FG_FRACTION * R))       # FGS: foreground RoIs sampled when background candidates exist
        # FGM: RoI slots the mask head runs on.  Normally FGS; TRAIN.MASK_SLOTS_ALL sizes it for the no-background case of
        # proposal_target_layer.py:155-158, where all R sampled RoIs are foreground (4x the mask-head work for a case real data never hits)
        FGM = R if getattr(TR, 'MASK_SLOTS_ALL', False) else FGS
        PS, MS = int(cfg.POOLING_SIZE), int(cfg.MASK_SIZE)
        C4 = self._C4_feat_dim
        H, W = int(d['data'].shape[1]), int(d['data'].shape[2])
        im_h, im_w = float(d['im_info'][0]), float(d['im_info'][1])
        self._im_hw = (im_h, im_w)
        self._mark('step start')
        main = torch.cuda.current_stream()
        S = self.streams() if self.use_streams else None
        if backward:                                        # (a forward-only pass - get_summary() between two replayed steps - leaves the last backward pass's set in place:
            self._fresh = set()                             # optim.update_range judges staleness from it)  gradients written (not added to) by this pass's grouped weight gradients
        if self.update_clears_grad:
            # optimizer.zero_grad() (TV:383) is folded into the update kernel: the buffer is zero here unless a backward pass went by without an update
            if backward:
                if getattr(self, '_bwd_pending', 0):
                    raise RuntimeError('two backward passes without an optimiser update in between: the gradients would add up '
                                       '(the update clears them; construct SGD(keep_grad=True) to have every step clear them itself)')
                self._bwd_pending = 1
        elif S is not None and self.update_on_wg:
            self.sfork(main, S['wg'])                       # (an update that ran on this queue after all, e.g. early partial updates)
            with torch.cuda.stream(S['wg']):                # behind the previous step's update, which read the gradients
                O.memset_zero(P.grad)
        else:
            O.memset_zero(P.grad)
        import contextlib
        def on(name):
            return torch.cuda.stream(S[name]) if S is not None else contextlib.nullcontext()
        # ---- expression encoding (ENC:27-82) forked onto the language stream: 80 dependent GEMV launches that overlap with the backbone
        if S is not None:
            self.sfork(main, S['lang'])
        # Round 5: every launch that does not depend on the image leaves the main queue, where a launch boundary is 2.4 - 3.2 us of the step
        # (DESIGN.md 4.7l).  The step counter (fresh RNG on every replay: every reader - dropout masks, sampling keys - runs on the language
        # stream, behind this launch), the clear of the loss accumulators and of the dynamic-filter gradient, the RoI sampling keys: all at
        # the head of the language stream, which the main queue joins before the dynamic filters.
        post_ = int(cfg['TRAIN' if self._mode == 'TRAIN' else 'TEST'].RPN_POST_NMS_TOP_N)
        with on('lang'):
            O.counter_inc(self.seed_counter())
            loss = self.buf('loss', (8,), f32, zero=True)
            NF_ = 7 * C4 + 7
            dfilt_ = self.buf('dyn.dfilt', (NF_,), f32, zero=True) if backward else None
            n_gt_ = int(d['gt_boxes'].shape[0])
            roi_keys = (self._keys('roi_fg_keys', post_ + n_gt_), self._keys('roi_bg_keys', post_ + n_gt_), self._keys('roi_bg_rand', R))
        if S is not None and self.update_on_wg:
            with torch.cuda.stream(S['lang']):
                self.join_update(full=False)                # the encoder reads updated weights
        with on('lang'):
            hidden = self._encoder_fwd(d)
            HD = hidden.numel()
            NF = 7 * C4 + 7                      # layout the correlation kernels read: 7 filters [C4] + 7 mixing weights
            NFP = self._NFP
            filt = self.buf('dyn.filt', (NF,), f32, zero=(NFP != NF))
            O.linear_fwd(hidden, P.gview('dyn_w', NFP * HD), P.gview('dyn_b', NFP), filt, 1, NFP, HD, act=2)
            if NFP != NF:
                # baseline network (network.py:475-479): one filter, response = its correlation -> mixing weights (1,0,..,0)
                if not hasattr(self, '_r_one'):
                    self._r_one = torch.tensor([1.0, 0, 0, 0, 0, 0, 0], dtype=f32, device=self.device)
                O.memcpy(filt[7 * C4:], self._r_one)
            # the pieces of the captioner that depend only on the tokens and the RNG counter (dropout masks, word embedding, i2h sums, the
            # zeroed backward buffer): off the caption branch's dependent chain, which is the step's critical path
            self._cap_pre = self._caption_pre(d) if (self.var['cap'] is not None and 'cap' not in self.knockout) else None
        self._mark('encoder_fwd(lang)')
        saved = {}
        base, Hc, Wc = self._backbone_fwd(d, saved)
        self._mark('layer1-3 fwd')
        HW = Hc * Wc
        t['net_conv_base'] = base
        # ---- dynamic filters (NET:504-562) ----
        self.join_deferred()         # last step's deferred weight gradients (they read the activations overwritten from here on) + their update
        self.join_transposes()       # last step's transposed weight copies (first readers: caption / RoI branches below)
        if S is not None:
            self.sfork(S['lang'], main)
        # the scatter targets of the backbone's stride-2 blocks (layer3[0]'s dx) are cleared here on the language stream, idle until the RPN
        # losses, instead of by a launch inside the backward chain on the main queue (which joins this stream again before the losses)
        self._precleared = set()
        if backward and S is not None:
            with on('lang'):
                for li in (3, 2):
                    if (li, 0) not in saved:                    # (another backbone: vgg16)
                        continue
                    blk = self.layers[li][0]
                    if li > cfg.RESNET.FIXED_BLOCKS and blk.down is not None and blk.stride != 1 and blk.need_dx:
                        x_, a1_, a2_, IH_, IW_, OH_, OW_, n_ = saved[(li, 0)]
                        O.memset_zero(self.buf('l%d.0.dx' % li, (n_ * IH_ * IW_, blk.inpl)))
                        self._precleared.add('l%d.0' % li)
        net_conv = self.buf('dyn.y', (HW, C4)); resp = self.buf('dyn.resp', (HW,), f32); respk = self.buf('dyn.respk', (HW, 7), f32)
        gate = 1 if self.var['gate'] == 'sigmoid' else 0
        O.dynfilter_fwd(base, filt, filt[7 * C4:], net_conv, resp, respk, Hc, Wc, C4, gate=gate)
        t['net_conv'], t['response'] = net_conv, resp
        dresp_extra = None
        if gate:
            # response loss (network_cycle_response.py:415-423) and its gradient w.r.t. the raw response
            dresp_extra = self.buf('dyn.dresp_loss', (HW,), f32)
            O.response_loss(resp, d['gt_masks'], H, W, Hc, Wc, 1.0, loss, dresp_extra)
        # ---- caption-cycle branch (NET:415-439), forward AND backward, forked onto the caption stream: it needs only net_conv
        # and rejoins at d(net_conv); layer4's weight gradients from both branches accumulate atomically.
        AF = self.opt['att_feat_size']

        def l4_on_map(xin, tag):
            x, hh, ww = xin, Hc, Wc
            for b, blk in enumerate(self.layers[4]):
                x, hh, ww, sv = blk.fwd(x, 1, hh, ww, '%s.%d' % (tag, b))
                saved[(tag, b)] = sv
            return x

        def l4_on_map_bwd(g, tag, in_relu=False):
            # in_relu: the map that was fed in is itself a ReLU output (layer3's), so its gradient is masked here
            self.prio_floor = self.cap_map_prio
            for b in reversed(range(len(self.layers[4]))):
                g = self.layers[4][b].bwd(g, saved[(tag, b)], '%s.%d' % (tag, b), x_is_relu_out=(b > 0 or in_relu))
            self.prio_floor = 0
            return g

        def caption_branch():
            """generator; returns (d net_conv, d base or None) contributed by the caption loss.  It yields between its four pieces so that
            the launches of the main path are ISSUED in between: one host thread feeds every stream, and the ~220 launches of this branch
            in one burst would leave the main queue without work for their whole issue time."""
            if 'cap' in self.knockout:
                return self.buf('l4m.skip', (HW, C4)), None
            self._mark('dyn fwd')
            feats = l4_on_map(net_conv, 'l4m')
            self._mark('cap: layer4 on map fwd')
            att = self.buf('cap.att', (196, AF))
            if self.var['cap'] == 'mask':
                gm = self.buf('cap.gm', (HW,), f32)
                O.mask_downsample(d['gt_masks'], gm, H, W, Hc, Wc)
                O.adaptive_pool_fwd(feats, None, att, Hc, Wc, 2048, 14, 14, AF)
                O.adaptive_pool_fwd(feats, gm, att[:, 2048:], Hc, Wc, 2048, 14, 14, AF)
                t.update({'feats_all': feats, 'att_feats': att, 'gt_mask_small': gm})
            else:
                # cycle_response: [layer4(map before the gating) ; layer4(gated map)] (network_cycle_response.py:425-439)
                feats_b = l4_on_map(base, 'l4b')
                O.adaptive_pool_fwd(feats_b, None, att, Hc, Wc, 2048, 14, 14, AF)
                O.adaptive_pool_fwd(feats, None, att[:, 2048:], Hc, Wc, 2048, 14, 14, AF)
                t.update({'feats_all': feats, 'feats_before_all': feats_b, 'att_feats': att})
            self._mark('cap: pools')
            yield
            self._caption_fwd(d, att, loss)
            self._mark('cap: captioner fwd')
            if not backward:
                return None, None
            yield
            datt = self._caption_bwd(d, att)
            self._mark('cap: captioner bwd')
            yield
            def deferred_grads():
                # captioner parameter gradients off the critical path: on the language stream (idle here), ordered after this branch
                if S is not None:
                    self.sfork(S['cap'], S['lang'])
                with on('lang'):
                    for f in self._cap_deferred:
                        f()
                    self._mark('cap: deferred parameter gradients (lang)')
                self._cap_deferred = []
            g = self.buf('l4m.g', (HW, 2048))
            if self.var['cap'] == 'mask':
                O.adaptive_pool_bwd(datt, AF, 0, 2048, gm, g, feats, Hc, Wc, 2048, 14, 14)
                r = l4_on_map_bwd(g, 'l4m')
                self._mark('cap: pool bwd + layer4 on map dgrad')
                deferred_grads()
                return r, None
            gb = self.buf('l4b.g', (HW, 2048))
            O.adaptive_pool_bwd(datt, AF, 0, 0, None, gb, feats_b, Hc, Wc, 2048, 14, 14)
            O.adaptive_pool_bwd(datt, AF, 2048, 0, None, g, feats, Hc, Wc, 2048, 14, 14)
            d_base_cap = l4_on_map_bwd(gb, 'l4b', in_relu=True)
            r = l4_on_map_bwd(g, 'l4m')
            deferred_grads()
            return r, d_base_cap
        cap_state = dict(gen=None, out=(None, None))
        if self.var['cap'] is not None:
            if S is not None:
                self.sfork(main, S['cap'])                  # the branch needs net_conv only: ordered after the dynamic filters, whenever issued
            cap_state['gen'] = caption_branch()

        def cap_advance(finish=False):
            """issue the next piece (or all remaining pieces) of the caption branch on its stream"""
            while cap_state['gen'] is not None:
                with on('cap'):
                    try:
                        next(cap_state['gen'])
                    except StopIteration as e:
                        cap_state['out'] = e.value if e.value is not None else (None, None)
                        cap_state['gen'] = None
                if not finish:
                    break
        cap_advance()                                         # layer4 on the map + pooled caption features
        self._mark('dyn + caption branch (cap)')
        # ---- RPN (NET:235-275) ----
        rpn = self.buf('rpn.a', (HW, 512))
        self.rpn_conv.fwd(net_conv, 1, Hc, Wc, rpn, relu=True)
        NPR = P.rpn_npad
        rheads = self.buf('rpn.heads', (HW, NPR), f32)
        self.rpn_heads.fwd(rpn, 1, Hc, Wc, rheads, out_f32=True)
        nA = HW * A
        prob = self.buf('rpn.prob', (HW, 2 * A), f32); boxes = self.buf('rpn.boxes', (nA, 4), f32); scores = self.buf('rpn.scores', (nA,), f32)
        O.rpn_decode(rheads, NPR, self.base_anchors, Hc, Wc, A, 16, im_h, im_w, prob, boxes, scores)
        t['rpn_heads'], t['rpn_cls_prob'] = rheads, prob
        self._mark('rpn conv+heads+decode')
        key = 'TRAIN' if self._mode == 'TRAIN' else 'TEST'
        pre = int(cfg[key].RPN_PRE_NMS_TOP_N); post = int(cfg[key].RPN_POST_NMS_TOP_N)
        pre = nA if pre <= 0 else min(pre, nA)
        sb = self.buf('prop.sb', (pre, 4), f32); ss = self.buf('prop.ss', (pre,), f32); si = self.buf('prop.si', (pre,), torch.int32)
        O.sort_topk(scores, boxes, nA, pre, self.buf('prop.sortws', (O.sort_ws_ints(nA),), torch.int32), sb, ss, si)                                      # PL:49-53
        nms_ws = self.buf('prop.nmsws', (O.nms_workspace_bytes(pre) // 8 + 8,), torch.int64)
        keep = self.buf('prop.keep', (post,), torch.int32); nkeep = self.buf('prop.nkeep', (1,), torch.int32)
        O.nms(sb, pre, float(cfg[key].RPN_NMS_THRESH), 0 if cfg.NMS_CMP == 'ge' else 1, post, nms_ws, keep, nkeep)   # PL:56-60
        rois_all = self.buf('prop.rois', (post, 5), f32); rsc_all = self.buf('prop.rsc', (post,), f32)
        O.gather_rois(sb, ss, keep, nkeep, post, rois_all, rsc_all)
        t['proposal_rois'], t['proposal_n'], t['proposal_scores'] = rois_all, nkeep, rsc_all
        cap_advance()                                         # captioner forward
        if self.parity is not None and self.parity.get('forced_proposals') is not None:
            fr, fs = self.parity['forced_proposals']
            rois_all = self.buf('prop.rois_forced', (post, 5), f32, zero=True); rsc_all = self.buf('prop.rsc_forced', (post,), f32, zero=True)
            rois_all[:fr.shape[0]].copy_(fr); rsc_all[:fs.shape[0]].copy_(fs)
            nkeep = torch.tensor([fr.shape[0]], dtype=torch.int32, device=self.device)
        self._mark('sort+nms+gather')
        # ---- targets (ATL:19-153, PTL:22-204) ----
        # anchor targets only need the gt box: they run on the language stream, beside the proposal chain
        rl = self.buf('atl.labels', (nA,), torch.int32); rt = self.buf('atl.t', (HW, 4 * A), f32)
        ri = self.buf('atl.i', (HW, 4 * A), f32); ro = self.buf('atl.o', (HW, 4 * A), f32)
        aws = self.buf('atl.ws', (O.anchor_target_ws_ints(nA),), torch.int32)
        n_gt = int(d['gt_boxes'].shape[0])
        if S is not None:
            self.sfork(main, S['lang'])
        with on('lang'):
            O.anchor_target(d['gt_boxes'], n_gt, self.base_anchors, Hc, Wc, A, 16, im_h, im_w, self._keys('rpn_fg_keys', nA),
                            self._keys('rpn_bg_keys', nA), TR.RPN_NEGATIVE_OVERLAP, TR.RPN_POSITIVE_OVERLAP, int(TR.RPN_BATCHSIZE),
                            TR.RPN_FG_FRACTION, rl, rt, ri, ro, aws)
            # ... and so do the RPN losses and the RPN's own backward pass (NET:375-390): nothing of them depends on the proposals, the proposal
            # chain keeps ONE compute unit busy for 0.3 ms, and at the end of the window these four launches were 0.11 ms of the main path
            d_rheads = self.buf('rpn.dheads', (HW, NPR))
            d_nc_rpn = self.buf('rpn.dnc', (HW, C4))

            def rpn_bwd():
                O.rpn_loss(rheads, NPR, rl, rt, ri, ro, Hc, Wc, A, 3.0, 1.0, loss, d_rheads, NPR, atl_ws=aws)
                if backward:
                    self.rpn_heads.wgrad(d_rheads, rpn, 1, Hc, Wc)
                    drpn = self.buf('rpn.da', (HW, 512))
                    self.rpn_heads.dgrad(d_rheads, 1, Hc, Wc, drpn, ref=rpn)
                    self.rpn_conv.wgrad(drpn, net_conv, 1, Hc, Wc)
                    self.rpn_conv.dgrad(drpn, 1, Hc, Wc, d_nc_rpn)
                    if self.rpn_bwd_early and self.rpn_wgrad_early:
                        self.flush_wgrads('rpn')                # the weight-gradient stream is idle until the caption join
                    self._mark('rpn loss + bwd')
            if self.rpn_bwd_early:
                rpn_bwd()
        t.update({'rpn_labels': rl, 'rpn_bbox_targets': rt, 'rpn_bbox_inside': ri, 'rpn_bbox_outside': ro})
        rois = self.buf('ptl.rois', (R, 5), f32); labels = self.buf('ptl.labels', (R,), torch.int32)
        btio = self.buf('ptl.btio', (3, R, 4 * nc), f32); bt, bi, bo = btio[0], btio[1], btio[2]     # (one allocation: l2s_proposal_target clears it in one launch)
        mt = self.buf('ptl.mt', (FGM, MS * MS), f32); counts = self.buf('ptl.counts', (4,), torch.int32)
        pws = self.buf('ptl.ws', (4 * (post + n_gt) + R + 16,), torch.int32)
        cst = self._consts()
        assert post == post_ and n_gt == n_gt_
        O.proposal_target(rois_all, rsc_all, nkeep, post, d['gt_boxes'], n_gt, d['gt_masks'], H, W, roi_keys[0],
                          roi_keys[1], roi_keys[2], R, FGS, FGM, TR.FG_THRESH, TR.BG_THRESH_HI,
                          TR.BG_THRESH_LO, cst['means'], cst['stds'], cst['inw'], nc, MS, rois, labels, bt, bi, bo, mt, counts, pws)
        t.update({'rois': rois, 'labels': labels, 'bbox_targets': bt, 'bbox_inside': bi, 'bbox_outside': bo, 'mask_targets': mt, 'counts': counts})
        self._mark('targets')
        self.conv_algo = 7 if self.roi_pdma else None       # (L2S_ALGO_PDMA)
        cheads, NPC, mscore = self._roi_head_fwd(net_conv, Hc, Wc, rois, R, FGM, saved)
        self.conv_algo = None
        self._mark('roi head fwd')
        cap_advance()                                         # captioner backward
        # ---- detection losses + head gradients (NET:375-413) ----
        d_cheads = self.buf('roi.dheads', (R, NPC)); dscore = self.buf('mask.dscore', (FGM * MS * MS,), f32)
        if S is not None:
            self.sfork(S['lang'], main)                    # anchor targets, RPN losses, RPN backward
        if not self.rpn_bwd_early:
            rpn_bwd()
        O.rcnn_loss(cheads, NPC, labels, bt, bi, bo, R, nc, 1.0, loss, d_cheads, NPC)
        if mscore is not None:
            O.mask_loss(mscore, nc, labels, mt, counts, FGM, MS * MS, 1.0, loss, dscore)
        # =================================== backward (detection side, main stream) ===================================
        dp = self.dp if self.dp is not None else self._early_op     # either one takes the finished gradient prefixes
        if not backward:
            cap_advance(finish=True)
            if S is not None and self.var['cap'] is not None:
                self.sfork(S['cap'], main)
            O.total_loss(loss, self._cap_loss_weight)
            t['loss'] = loss
            return loss
        self._mark('losses')
        self.conv_algo = 7 if self.roi_pdma else None
        d_nc_roi = self._roi_head_bwd(d_cheads, dscore, labels, counts, rois, Hc, Wc, R, FGM, saved)
        self.conv_algo = None
        self._mark('roialign bwd')
        cap_advance(finish=True)                              # layer4 on the map backward
        d_nc_cap, d_base_cap = cap_state['out']
        self._mark('main reaches the caption join')
        if S is not None and self.var['cap'] is not None:
            self.sfork(S['cap'], main)                     # join the caption branch
        t['loss'] = loss                                    # (summed on the language stream below, behind every loss launch)
        # weight gradients of the caption branch, the RoI head (layer4: RoI pass + caption pass = two pixel segments of one problem)
        # and the RPN, as grouped launches on the weight-gradient stream
        if self.defer_heads:
            self.wgq.defer()                                    # launched by the optimiser, behind the first part of the update (optim.SGD.defer)
            if dp is not None:
                self.dp_ready('caption')                        # the captioner's own matrices are final; att_embed .. RPN follow with the deferred launches
        else:
            self.flush_wgrads('heads')
            if dp is not None:
                self.dp_ready('heads')                          # caption + layer4 + RoI/mask heads are final here
        d_nc = self.buf('dyn.dy', (HW, C4))
        if d_nc_cap is not None:
            O.add3(d_nc_cap, d_nc_rpn, d_nc_roi, d_nc)
        else:
            O.add3(d_nc_rpn, None, d_nc_roi, d_nc)        # (dtype, -, fp32) operands
        self._mark('caption join + add3')
        # dynamic filters (NET:504-562)
        dbase = self.buf('dyn.dx', (HW, C4)); dfilt = dfilt_; dresp_ws = self.buf('dyn.dresp', (O.dynfilter_ws_floats(Hc, Wc, C4),), f32)
        # (only dbase is needed on this queue: the filter / mixing-weight gradients are finished on the language stream below)
        O.dynfilter_bwd(d_nc, base, filt, filt[7 * C4:], resp, respk, dbase, base, None, None, dresp_ws, Hc, Wc, C4,
                        gate=gate, dresp_extra=dresp_extra)
        if d_base_cap is not None:
            O.add3(dbase, d_base_cap, None, dbase)         # cycle_response: layer4 also ran on the map before the gating
        # language-side backward (dynamic FCs, bi-LSTM, embedding: ~170 small dependent launches) forked onto the language
        # stream; the backbone backward below does not depend on it.  Joined by the optimiser (join_side()).
        if S is not None:
            self.sfork(main, S['lang'])
        with on('lang'):
            O.total_loss(loss, self._cap_loss_weight)
            O.dynfilter_bwd_finish(dresp_ws, respk, dfilt, dfilt[7 * C4:], Hc, Wc, C4)
            O.act_bwd(dfilt, filt, 2)
            O.linear_bwd_w(dfilt, hidden, P.gview('dyn_w', NFP * HD, P.grad), P.gview('dyn_b', NFP, P.grad), 1, NFP, HD)
            dhidden = self.buf('enc.dhidden', (HD,), f32)
            nws = O.linear_bwd_x_ws_floats(1, NFP, HD)
            O.linear_bwd_x(dfilt, P.gview('dyn_w', NFP * HD), dhidden, 1, NFP, HD, ws=self.buf('bwdx.ws.dyn_w', (max(nws, 1),), f32))
            self._encoder_bwd(d, dhidden)
            self._mark('language bwd done (lang)')
        self._mark('dyn bwd + language bwd(lang)')
        self._backbone_bwd(dbase, saved, S, main, dp)
        self.flush_wgrads('backbone')                           # whatever a backbone variant left queued
        self._mark('layer3-2 bwd')
        if S is not None:
            self.sfork(S['lang'], main)
        self._pass_without_step = True                          # until optim.SGD.step records the tail's event slots (Network.join_update)
        return loss


    # ------------------------------------------------------------------ TEST mode (NET:488-593 mode == 'TEST', 595-626, 650-658)
    def _backbone_and_filter(self, d):
        """conv1..layer3, expression encoding, dynamic filters -> (net_conv, base, Hc, Wc); single stream (inference)."""
        P = self.P
        C4 = self._C4_feat_dim
        H, W = int(d['data'].shape[1]), int(d['data'].shape[2])
        self.join_update()
        hidden = self._encoder_fwd(d)
        HD = hidden.numel()
        NF, NFP = 7 * C4 + 7, self._NFP
        filt = self.buf('dyn.filt', (NF,), f32, zero=(NFP != NF))
        O.linear_fwd(hidden, P.gview('dyn_w', NFP * HD), P.gview('dyn_b', NFP), filt, 1, NFP, HD, act=2)
        if NFP != NF:
            if not hasattr(self, '_r_one'):
                self._r_one = torch.tensor([1.0, 0, 0, 0, 0, 0, 0], dtype=f32, device=self.device)
            O.memcpy(filt[7 * C4:], self._r_one)
        base, Hc, Wc = self._backbone_fwd(d, {})
        net_conv = self.buf('dyn.y', (Hc * Wc, C4)); resp = self.buf('dyn.resp', (Hc * Wc,), f32); respk = self.buf('dyn.respk', (Hc * Wc, 7), f32)
        O.dynfilter_fwd(base, filt, filt[7 * C4:], net_conv, resp, respk, Hc, Wc, C4, gate=1 if self.var['gate'] == 'sigmoid' else 0)
        return net_conv, base, resp, Hc, Wc

    def _roi_heads_test(self, net_conv, Hc, Wc, rois, n, labels=None):
        """crop-pool -> layer4 -> (cls scores, cls prob, de-normalised deltas) and mask probabilities for `n` rois [n][5]."""
        P, dt = self.P, self.dt
        C4, nc = self._C4_feat_dim, self._num_classes
        PS, MS = int(cfg.POOLING_SIZE), int(cfg.MASK_SIZE)
        pool5 = self._rois_pool_fwd(net_conv, Hc, Wc, rois, n, {})
        x, hh, ww = pool5, PS, PS
        for b, blk in enumerate(self.layers[4]):
            x, hh, ww, _ = blk.fwd(x, n, hh, ww, 'l4t.%d' % b)
        fc7s = x
        fc7 = self.buf('roi.fc7', (n, 2048))
        O.avgpool_fwd(fc7s, fc7, n, PS * PS, 2048)
        NPC = P.rcnn_npad
        cheads = self.buf('roi.heads', (n, NPC), f32)
        self.rcnn_heads.fwd(fc7, n, 1, 1, cheads, out_f32=True)
        cst = self._consts()
        cls_prob = self.buf('test.cls_prob', (n, nc), f32); bbox_pred = self.buf('test.bbox_pred', (n, 4 * nc), f32)
        O.rcnn_predict(cheads, NPC, n, nc, cst['stds'], cst['means'], cls_prob, bbox_pred)
        up = self.buf('mask.up', (n * MS * MS, 256))
        O.conv_igemm(fc7s, self.up_wT, up, n, PS, PS, 2048, PS, PS, 4 * 256, bias=P.view('mask_up_sampling.bias'), relu=True, deconv=True, dt=dt)
        mscore = self.buf('mask.score', (n * MS * MS, nc), f32)
        self.mask_pred.fwd(up, n, MS, MS, mscore, out_f32=True)
        if labels is None:
            mprob = self.buf('test.mask_prob', (n * MS * MS, nc), f32)
            O.mask_prob(mscore, nc, nc, None, MS * MS, n * MS * MS, mprob)
        else:
            mprob = self.buf('test.mask_prob_l', (n, MS, MS), f32)
            O.mask_prob(mscore, nc, nc, labels, MS * MS, n * MS * MS, mprob)
        return cheads, cls_prob, bbox_pred, mprob

    def forward_test(self, d):
        """NET:628-662 with mode == 'TEST'.  Fills self._predictions like the reference (device tensors, NHWC / row-major)."""
        self.join_transposes()
        self.t = {}
        A = self._num_anchors
        im_h, im_w = float(d['im_info'][0]), float(d['im_info'][1])
        self._im_hw = (im_h, im_w)
        net_conv, base, resp, Hc, Wc = self._backbone_and_filter(d)
        HW = Hc * Wc
        P = self.P
        rpn = self.buf('rpn.a', (HW, 512))
        self.rpn_conv.fwd(net_conv, 1, Hc, Wc, rpn, relu=True)
        NPR = P.rpn_npad
        rheads = self.buf('rpn.heads', (HW, NPR), f32)
        self.rpn_heads.fwd(rpn, 1, Hc, Wc, rheads, out_f32=True)
        nA = HW * A
        prob = self.buf('rpn.prob', (HW, 2 * A), f32); boxes = self.buf('rpn.boxes', (nA, 4), f32); scores = self.buf('rpn.scores', (nA,), f32)
        O.rpn_decode(rheads, NPR, self.base_anchors, Hc, Wc, A, 16, im_h, im_w, prob, boxes, scores)
        if str(cfg.TEST.MODE) == 'top':
            # NET:263-264 -> proposal_top_layer.py:18-67: the RPN_TOP_N best anchors, decoded + clipped (rpn_decode did both), no NMS
            post = int(cfg.TEST.RPN_TOP_N)
            rois = self.buf('tprop.rois_top', (post, 5), f32, zero=True)
            if nA >= post:
                sb = self.buf('tprop.sb_top', (post, 4), f32); ss = self.buf('tprop.ss_top', (post,), f32); si = self.buf('tprop.si_top', (post,), torch.int32)
                O.sort_topk(scores, boxes, nA, post, self.buf('prop.sortws', (O.sort_ws_ints(nA),), torch.int32), sb, ss, si)
                rois[:, 1:].copy_(sb)
            else:
                # fewer anchors than RPN_TOP_N (:44-49): drawn with replacement from numpy's generator, as the reference does
                idx = torch.from_numpy(np.random.choice(nA, size=post, replace=True)).to(self.device)
                rois[:, 1:].copy_(boxes[idx])
            n = post
            nkeep = None
        elif str(cfg.TEST.MODE) == 'nms':
            pre = int(cfg.TEST.RPN_PRE_NMS_TOP_N); post = int(cfg.TEST.RPN_POST_NMS_TOP_N)
            pre = nA if pre <= 0 else min(pre, nA)
            sb = self.buf('tprop.sb', (pre, 4), f32); ss = self.buf('tprop.ss', (pre,), f32); si = self.buf('tprop.si', (pre,), torch.int32)
            O.sort_topk(scores, boxes, nA, pre, self.buf('prop.sortws', (O.sort_ws_ints(nA),), torch.int32), sb, ss, si)
            nms_ws = self.buf('tprop.nmsws', (O.nms_workspace_bytes(pre) // 8 + 8,), torch.int64)
            keep = self.buf('tprop.keep', (post,), torch.int32); nkeep = self.buf('tprop.nkeep', (1,), torch.int32)
            O.nms(sb, pre, float(cfg.TEST.RPN_NMS_THRESH), 0 if cfg.NMS_CMP == 'ge' else 1, post, nms_ws, keep, nkeep)
            rois = self.buf('tprop.rois', (post, 5), f32, zero=True); rsc = self.buf('tprop.rsc', (post,), f32)
            O.gather_rois(sb, ss, keep, nkeep, post, rois, rsc)
            n = int(nkeep.item())                               # TEST mode returns host arrays anyway (NET:691-697)
        else:
            raise NotImplementedError(cfg.TEST.MODE)            # NET:265-266
        n_own = n
        own = rois
        if self.parity is not None and self.parity.get('forced_proposals') is not None:
            fr, _ = self.parity['forced_proposals']
            rois = self.buf('tprop.rois_forced', (post, 5), f32, zero=True)
            rois[:fr.shape[0]].copy_(fr); n = int(fr.shape[0])
        # heads run on all `post` slots (static shapes); rows >= n are padding and sliced off
        cheads, cls_prob, bbox_pred, mprob = self._roi_heads_test(net_conv, Hc, Wc, rois, post)
        nc = self._num_classes; MS = int(cfg.MASK_SIZE)
        if mprob is None:                                   # VGG16 / Faster R-CNN network: no mask branch (network_vgg.py:614)
            self._predictions = dict(net_conv=net_conv, net_conv_hw=(Hc, Wc), response=resp, rois=rois[:n], own_rois=own[:n_own],
                                     cls_score=cheads[:n, :nc], cls_prob=cls_prob[:n], bbox_pred=bbox_pred[:n], rpn_cls_prob=prob)
            return self._predictions
        self._predictions = dict(net_conv=net_conv, net_conv_hw=(Hc, Wc), response=resp, rois=rois[:n], own_rois=own[:n_own],
                                 cls_score=cheads[:n, :nc], cls_prob=cls_prob[:n], bbox_pred=bbox_pred[:n],
                                 mask_prob=mprob.view(post, MS, MS, nc)[:n], rpn_cls_prob=prob)
        return self._predictions

    def test_image(self, blobs):
        """NET:684-699: (cls_score, cls_prob, bbox_pred, rois) as float32 ndarrays + net_conv (device tensor [H*W][1024], see
        self._predictions['net_conv_hw']), for the single expression in `blobs`."""
        self.eval()
        p = self.forward_test(self.upload_blob(blobs, 0))
        return (p['cls_score'].float().cpu().numpy().copy(), p['cls_prob'].cpu().numpy().copy(), p['bbox_pred'].cpu().numpy().copy(),
                p['rois'].cpu().numpy().copy(), p['net_conv'])

    def _predict_masks_from_boxes_and_labels(self, net_conv, boxes, labels):
        """NET:595-626: boxes ndarray (n,4) in the scaled image, labels ndarray (n,) -> device tensor (n,14,14) in [0,1]."""
        assert not self.training, 'only support testing mode'
        Hc, Wc = self._predictions['net_conv_hw']
        n = int(boxes.shape[0])
        rois = torch.from_numpy(np.hstack([np.zeros((n, 1)), boxes]).astype(np.float32)).to(self.device)
        lab = torch.from_numpy(np.asarray(labels).astype(np.int32)).to(self.device)
        return self._roi_heads_test(net_conv, Hc, Wc, rois, n, labels=lab)[3]

    def _consts(self):
        if not hasattr(self, '_cst'):
            TR = cfg.TRAIN
            mk = lambda v: torch.tensor(list(v), dtype=f32, device=self.device)
            self._cst = dict(means=mk(TR.BBOX_NORMALIZE_MEANS), stds=mk(TR.BBOX_NORMALIZE_STDS), inw=mk(TR.BBOX_INSIDE_WEIGHTS))
        return self._cst
