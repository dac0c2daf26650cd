"""Network — host-side mirror of the reference's model-graph layer
(pyutils/mask-faster-rcnn/lib/nets/network_cycle_res5_2.py, "NET"): same public methods
(`create_architecture`, `train_step`, `train_step_with_summary`, `get_summary`, `test_image`,
`state_dict`/`load_state_dict`, `train`/`eval`/`cuda`), same losses tuple.  Every arithmetic op is a
HIP kernel behind the C ABI (lang2seg_amd/ops.py); this file only sequences launches on one stream
with an explicit, static activation plan (no autograd, no host round trips inside the step:
NMS, target assignment and sampling stay on device, see csrc/roi.hip).

Forward/backward structure follows NET:488-593 (_predict), NET:375-454 (_add_losses) and
NET:702-719 (train_step); the manual backward is the adjoint of exactly those ops."""
import numpy as np
import os
import torch

from .. import ops as O
from .._lib import F32, BF16, LOSS_CAP
from ..model.config import cfg
from .params import ParamStore
from . import anchors as ANC


class ConvOp(object):
    """One convolution / linear layer on the MFMA implicit-GEMM kernels (forward, data-grad, weight-grad)."""

    def __init__(self, net, wkey, Cin, Cout, k=1, stride=1, pad=0, bias_key=None, need_dgrad=True, group=None, Cout_pad=None, full_map=False):
        self.net, self.wkey, self.Cin, self.Cout, self.k, self.stride, self.pad = net, wkey, Cin, Cout, k, stride, pad
        # full_map: a 'valid' k x k convolution that is only ever applied to k x k maps (VGG's fc6 on the 7x7 pooled RoIs: one output pixel).  Its
        # data gradient is then a plain GEMM dx[n][(y, x, ci)] = sum_co g[n][co] W[co][y][x][ci]: the data-gradient weight copy is the transpose of
        # the [Cout][k k Cin] matrix ("one tap, k k Cin channels"), not the flipped-tap layout of a general convolution, which would run it as a
        # k x k convolution with pad k - 1 over 1x1 maps - 48 of every 49 taps on padding (fc6 at 600x1000: 2.42 ms = 0.009 of peak, a third of the
        # VGG step, found when bench.py --variant vgg first ran at the BASELINE size in round 5).
        self.full_map = bool(full_map and k > 1 and pad == 0 and stride == 1)
        self.t_taps, self.t_cin = (1, k * k * Cin) if self.full_map else (k * k, Cin)     # shape of the transposed (data-gradient) copy
        self.bias_key, self.need_dgrad, self.group = bias_key, need_dgrad, group
        self.Np = Cout if Cout_pad is None else Cout_pad      # padded output width (grouped heads)
        # the backbone's layer1-3 launches form latency-bound dependent chains that share their CUs with the weight-gradient stream: their
        # waves issue first (l2s_conv_desc.prio)
        self.prio = 1 if str(wkey).startswith(('resnet.layer1.', 'resnet.layer2.', 'resnet.layer3.')) and getattr(net, 'chain_prio', True) else 0
        P = net.P
        taps = k * k
        cnt = self.Np * taps * Cin
        if group is not None:
            self.trainable = True
            self.w_master = P.gview(group[0], cnt)
            self.w_grad = P.gview(group[0], cnt, P.grad)
            self.wf = P.gview(group[0], cnt, P.shadow)
            self.bias = P.gview(group[1], self.Np)
            self.bias_grad = P.gview(group[1], self.Np, P.grad)
            self.scale = None
        else:
            self.trainable = wkey in P.offsets
            self.scale = P.bn_scale.get(wkey)
            if self.trainable:
                self.w_master = P.view(wkey); self.w_grad = P.view(wkey, P.grad); self.wf = P.view(wkey, P.shadow)
            else:
                self.w_master = P.frozen[wkey].view(-1)
                self.w_grad = None
                self.wf = O.empty((cnt,), net.dt)
            if bias_key is not None and bias_key in P.offsets:
                self.bias = P.view(bias_key); self.bias_grad = P.view(bias_key, P.grad)
            elif bias_key is not None:                      # frozen layer with a plain bias (VGG conv1_1 .. conv2_2)
                self.bias = P.frozen[bias_key]; self.bias_grad = None
            else:
                self.bias = P.bn_bias.get(wkey); self.bias_grad = None
        self.wb = O.empty((cnt,), net.dt) if (need_dgrad and self.trainable) else None
        net.convs.append(self)

    def refresh(self, full=False):
        """(re)build the dtype copies the kernels read: forward [Cout][taps][Cin] (frozen layers only; trainable
        ones are written by the SGD kernel) and the data-gradient layout [Cin][taps flipped][Cout]."""
        taps = self.k * self.k
        if not self.trainable and full:
            O.weight_cast(self.w_master, self.scale, self.wf, self.Np, taps, self.Cin)
        if self.wb is not None:
            O.weight_transpose(self.w_master, self.scale, self.wb, self.Np, self.t_taps, self.t_cin)

    def out_hw(self, IH, IW):
        return (IH + 2 * self.pad - self.k) // self.stride + 1, (IW + 2 * self.pad - self.k) // self.stride + 1

    def fwd(self, x, n, IH, IW, y, add=None, relu=False, out_f32=False, tile=0):
        OH, OW = self.out_hw(IH, IW)
        O.conv_igemm(x, self.wf, y, n, IH, IW, self.Cin, OH, OW, self.Np, self.k, self.k, self.stride, self.pad,
                     bias=self.bias, add=add, relu=relu, out_f32=out_f32, tile=tile, dt=self.net.dt, ws=self.net.splitk_ws(n * OH * OW * self.Np), prio=max(self.prio, self.net.prio_floor),
                     algo=(self.net.conv_algo if self.k == 1 and self.stride == 1 and self.Np >= 1024 else None))
        return y

    def dgrad(self, g, n, IH, IW, dx, add=None, ref=None):
        """dx[n,IH,IW,Cin] = conv^T(g); epilogue: (+ add) then ReLU mask by ref > 0."""
        OH, OW = self.out_hw(IH, IW)
        if self.full_map:
            assert IH == self.k and IW == self.k and OH == 1 and OW == 1
            O.conv_igemm(g, self.wb, dx, n, 1, 1, self.Np, 1, 1, self.t_cin, 1, 1, 1, 0, add=add, ref=ref, dt=self.net.dt,
                         prio=max(self.prio, self.net.prio_floor))
            return dx
        if self.stride == 1:
            O.conv_igemm(g, self.wb, dx, n, OH, OW, self.Np, IH, IW, self.Cin, self.k, self.k, 1, self.k - 1 - self.pad,
                         add=add, ref=ref, dt=self.net.dt, ws=self.net.splitk_ws(n * IH * IW * self.Cin), prio=max(self.prio, self.net.prio_floor),
                         algo=(self.net.conv_algo if self.k == 1 and self.Cin >= 1024 else None))
        else:
            assert self.k == 1
            O.conv_igemm(g, self.wb, dx, n, OH, OW, self.Np, OH, OW, self.Cin, 1, 1, 1, 0, add=add, ref=ref,
                         scatter=(IH, IW, self.stride), dt=self.net.dt, prio=max(self.prio, self.net.prio_floor))
        return dx

    def wgrad(self, g, x, n, IH, IW):
        """weight (+bias) gradient.  Nothing needs it before the optimiser (or the gradient all-reduce of its stage), so it is only
        QUEUED here: Network.flush_wgrads launches the queued problems of a whole backward stage as one grouped launch on the
        weight-gradient stream (csrc/conv_wgrad.hip: no split-K, no atomics, two uses of one tensor = one problem with two pixel
        segments).  The bias gradient (a column sum) is launched right away on a weight-gradient stream."""
        ko = self.net.knockout                            # experiment only (bench.py --knockout); train_net refuses it
        if 'wgrad' in ko or ('wgrad4' in ko and str(self.wkey).startswith('resnet.layer4.')) or \
                ('wgrad3' in ko and str(self.wkey).startswith(('resnet.layer3.', 'resnet.layer2.'))):
            return
        OH, OW = self.out_hw(IH, IW)
        self.net.wgq.add(self.w_grad, g, x, n, IH, IW, self.Cin, OH, OW, self.Np, self.k, self.stride, self.pad)
        if self.bias_grad is not None:
            with self.net.fork_wgrad():
                O.colsum(g, n * OH * OW, self.Np, self.Np, self.bias_grad, ws=self.net.buf('colsum.ws.' + str(self.bias_key or self.group[1]), (32 * self.Np,), torch.float32))


class WgradQueue(object):
    """weight gradients waiting for the grouped launch of their backward stage (Network.flush_wgrads)"""

    def __init__(self, net):
        self.net, self.items, self.tables = net, [], {}
        self.deferred = []               # problems of the heads stage held back until the end of the step (Network.defer_heads)
        self.on_launch = None            # optional hook(tag, variant, flop, k) -> context manager (bench.py times the launches with it)

    def defer(self):
        """hold the queued problems back: they are launched by flush_deferred(), behind the first part of the optimiser update"""
        self.deferred.extend(self.items)
        self.items = []

    def flush_deferred(self, tag):
        assert not self.items, 'weight gradients queued behind the deferred ones'
        self.items, self.deferred = self.deferred, []
        self.flush(tag)

    def add(self, dw, g, x, n, IH, IW, Cin, OH, OW, Cout, k, stride, pad, lddy=None, ldx=None):
        self.items.append((dw, g, x, n, IH, IW, Cin, OH, OW, Cout, k, stride, pad, Cout if lddy is None else lddy, Cin if ldx is None else ldx))

    def flush(self, tag):
        from .._lib import WgradProb, ptr, load
        import ctypes as C
        if not self.items:
            return False
        lib = load()
        net = self.net
        items, self.items = self.items, []
        # uses of one tensor -> pixel segments of one problem (two per problem; a third use goes to a later launch, in order)
        by_dw = {}
        for it in items:
            by_dw.setdefault((it[0].data_ptr(), it[6], it[9], it[10], it[11], it[12]), []).append(it)
        rounds = {}
        for key, uses in by_dw.items():
            for r in range(0, len(uses), 2):
                seg = uses[r:r + 2]
                dw, g, x, n, IH, IW, Cin, OH, OW, Cout, k, stride, pad, lddy, ldx = seg[0]
                M = max(u[3] * u[7] * u[8] for u in seg)
                same = all(u[7] == u[4] and u[8] == u[5] for u in seg)
                tl = 256 if net.dt == BF16 else 0                   # (256: the 8-wave 256x256 tile / the LDS-DMA filter-row tile may be chosen)
                if self.SMALL_M_TILE and net.dt == BF16 and M < 8192 and Cin >= 256 and Cout >= 256 and Cin % 128 == 0 and Cout % 128 == 0:
                    tl = self.SMALL_M_TILE                          # A/B: 128-wide tiles for the backbone's problems (layer3)
                v = int(lib.l2s_wgrad_variant(Cin, Cout, k, k, stride, pad, int(same), M, tl))
                rounds.setdefault((r // 2, v), []).append(seg)
        bkp = 32 if net.dt == BF16 else 16
        # a tensor's first problem of the step WRITES its gradient (l2s_wgrad_prob.flags = 1: no read of dW, and the update need not clear it,
        # ParamStore.mark_overwritten); later contributions (a third use) add.  Network._fresh: the tensors written so far in this backward pass.
        fresh = getattr(net, '_fresh', None) if getattr(net, 'wgrad_overwrite', False) else None
        order = sorted(rounds)
        # A/B (V5_STREAM): the LDS-DMA filter-row launch (128 workgroups) on another stream, beside the 120-workgroup 256x256 launch
        plan = [('wg', order)]
        if tag == 'layer2' and net.use_streams and getattr(net, 'layer2_side', False):
            # the LAST stage of the backward pass on the other weight-gradient stream: 'wg' is still ~0.25 ms behind (a serial queue of grouped
            # launches since the caption join), and the next step's layer2 waits for exactly these gradients and their update, not for that backlog
            plan = [('wg2', order)]
            net._layer2_on_side = True
        if self.V5_STREAM != 'wg' and net.use_streams and getattr(net, 'dp', None) is None and (0, 5) in rounds and (0, 4) in rounds:
            plan = [(self.V5_STREAM, [(0, 5)]), ('wg', [o for o in order if o != (0, 5)])]
        for sname, keys in plan:
          with net.fork_wgrad(fixed=sname):
            ws = net.wgrad_ws() if sname == 'wg' else net.wgrad_ws(alt=True)
            for (rnd, v) in keys:
                probs = rounds[(rnd, v)]
                for c0 in range(0, len(probs), 64):
                    chunk = probs[c0:c0 + 64]
                    # few output tiles and many pixels (layer2: 128 channels, 9375 pixels): cut the pixels so that the launch still has
                    # ~1000 workgroups, at least 16 slices each; the partial tiles go to slabs of the workspace (no atomics)
                    tiles = [int(lib.l2s_wgrad_tiles(v, seg[0][6], seg[0][9], seg[0][10], seg[0][10])) for seg in chunk]
                    total = sum(tiles)
                    want = max(1, -(-self.MIN_WG // max(total, 1)))
                    if v == 4:
                        want = max(1, self.V4_FILL // max(total, 1))   # one 8-wave workgroup per CU: fill one round, not more
                    arr = (WgradProb * len(chunk))()
                    flop, off = 0.0, 0
                    for i, seg in enumerate(chunk):
                        dw, g, x, n, IH, IW, Cin, OH, OW, Cout, k, stride, pad, lddy, ldx = seg[0]
                        q = arr[i]
                        q.dw, q.nseg, q.Cin, q.Cout, q.KH, q.KW, q.stride, q.pad = dw.data_ptr(), len(seg), Cin, Cout, k, k, stride, pad
                        q.flags = 0
                        if fresh is not None and rnd == 0 and dw.data_ptr() not in fresh and dw.numel() == Cout * k * k * Cin:
                            q.flags = 1
                            fresh.add(dw.data_ptr())
                        slices = min(u[3] * u[7] * (u[8] + (1 if v in (2, 3, 5) else 0)) // bkp for u in seg)
                        split = max(1, min(want, slices // 16, 16))
                        if v in (5, 6):
                            split = 1                                   # the LDS-DMA launches: stream-K balanced (5) / one workgroup per 256x256 tile on purpose (6)
                        slab = Cout * k * k * Cin
                        if split > 1 and (off + split * slab) * 4 <= ws.numel() * 4:
                            q.split, q.ws_off = split, off
                            off += split * slab
                        else:
                            q.split, q.ws_off = 1, 0
                        for s_, u in enumerate(seg):
                            q.dy[s_], q.x[s_] = u[1].data_ptr(), u[2].data_ptr()
                            q.n_img[s_], q.IH[s_], q.IW[s_], q.OH[s_], q.OW[s_], q.lddy[s_], q.ldx[s_] = u[3], u[4], u[5], u[7], u[8], u[13], u[14]
                            flop += 2.0 * u[3] * u[7] * u[8] * Cout * k * k * Cin
                    raw = bytes(arr)
                    ck = (tag, rnd, v, c0, getattr(net, '_rec_key', None))
                    ent = self.tables.get(ck)
                    if ent is None or ent[0] != raw:
                        # pinned staging buffer + asynchronous copy on the weight-gradient stream (a pageable .to(device) is a host stall
                        # behind everything queued there); the pinned tensor lives as long as the table entry
                        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
                        if net.device != 'cpu' and torch.cuda.is_available():
                            host = host.pin_memory()
                        ent = (raw, host.to(net.device, non_blocking=True), host)
                        self.tables[ck] = ent
                    dt = O.dt_of(chunk[0][0][2])
                    ctx = self.on_launch(tag, v, flop, chunk[0][0][10]) if self.on_launch is not None else None
                    if ctx is not None:
                        ctx.__enter__()
                    O.call('l2s_conv_wgrad_grouped', ent[1].data_ptr(), C.cast(arr, C.c_void_p), len(chunk), v, dt, ws.data_ptr(),
                           ws.numel() * 4, O.stream())
                    if ctx is not None:
                        ctx.__exit__(None, None, None)
        return True

    V5_STREAM = 'wg'   # A/B: 'tr' = the filter-row launch of a stage beside the stage's other launches instead of behind them
    SMALL_M_TILE = 0   # A/B: tile argument for problems with fewer than 8192 pixels (128: 128x128 / 128x64-row tiles instead of 64x64)
    MIN_WG = 256   # workgroups a grouped launch should have before its problems stop splitting their pixels (384 until round 4: 208.5 -> 209.1 img/s, x4)
    # workgroups the 256x256 launch splits its pixels up to.  Round 4: 128, i.e. layer4's 120 tiles are NOT split any more - 120 workgroups on 120 CUs
    # leave the other CUs to the data-gradient chain (as the LDS-DMA filter-row launch does), and the slabs + reduce of the split are gone:
    # 202.0 -> 205.6 img/s, same box x 3
    V4_FILL = 128


class Bottleneck(object):
    """RES:78-114 (stride on the first 1x1)."""

    def __init__(self, net, prefix, inplanes, planes, stride, has_down, need_dx=True):
        self.net, self.prefix, self.stride, self.need_dx = net, prefix, stride, need_dx
        self.inpl, self.planes = inplanes, planes
        self.c1 = ConvOp(net, prefix + '.conv1.weight', inplanes, planes, 1, stride, 0, need_dgrad=need_dx)
        self.c2 = ConvOp(net, prefix + '.conv2.weight', planes, planes, 3, 1, 1)
        self.c3 = ConvOp(net, prefix + '.conv3.weight', planes, planes * 4, 1, 1, 0)
        self.down = ConvOp(net, prefix + '.downsample.0.weight', inplanes, planes * 4, 1, stride, 0, need_dgrad=need_dx) if has_down else None

    def fwd(self, x, n, IH, IW, tag):
        net = self.net
        OH, OW = self.c1.out_hw(IH, IW)
        a1 = net.buf(tag + '.a1', (n * OH * OW, self.planes))
        a2 = net.buf(tag + '.a2', (n * OH * OW, self.planes))
        y = net.buf(tag + '.y', (n * OH * OW, self.planes * 4))
        self.c1.fwd(x, n, IH, IW, a1, relu=True)
        self.c2.fwd(a1, n, OH, OW, a2, relu=True)
        if self.down is not None:
            sc = net.buf(tag + '.sc', (n * OH * OW, self.planes * 4))
            self.down.fwd(x, n, IH, IW, sc)
            self.c3.fwd(a2, n, OH, OW, y, add=sc, relu=True)
        else:
            self.c3.fwd(a2, n, OH, OW, y, add=x, relu=True)
        return y, OH, OW, (x, a1, a2, IH, IW, OH, OW, n)

    def fwd_rest(self, x, a1, sc, n, IH, IW, tag):
        """the block behind its first two 1x1 convolutions: a1 = relu(conv1(x)) and sc = downsample(x) were produced by the caller
        (the fused RoIAlign kernel, l2s_roialign_block0_fwd); stride 1 only"""
        net = self.net
        a2 = net.buf(tag + '.a2', (n * IH * IW, self.planes))
        y = net.buf(tag + '.y', (n * IH * IW, self.planes * 4))
        self.c2.fwd(a1, n, IH, IW, a2, relu=True)
        self.c3.fwd(a2, n, IH, IW, y, add=sc, relu=True)
        return y, IH, IW, (x, a1, a2, IH, IW, IH, IW, n)

    def bwd(self, g, saved, tag, x_is_relu_out=True):
        """g = dL/d(pre-ReLU sum) (already masked by y > 0).  Returns dL/dx masked by x > 0 when x is a ReLU output."""
        net = self.net
        x, a1, a2, IH, IW, OH, OW, n = saved
        self.c3.wgrad(g, a2, n, OH, OW)
        dz2 = net.buf(tag + '.dz2', (n * OH * OW, self.planes))
        self.c3.dgrad(g, n, OH, OW, dz2, ref=a2)
        self.c2.wgrad(dz2, a1, n, OH, OW)
        dz1 = net.buf(tag + '.dz1', (n * OH * OW, self.planes))
        self.c2.dgrad(dz2, n, OH, OW, dz1, ref=a1)
        self.c1.wgrad(dz1, x, n, IH, IW)
        if self.down is not None:
            self.down.wgrad(g, x, n, IH, IW)
        if not self.need_dx:
            return None
        dx = net.buf(tag + '.dx', (n * IH * IW, self.inpl))
        ref = x if x_is_relu_out else None
        if self.down is not None:
            if self.stride != 1 and tag not in net._precleared:
                O.memset_zero(dx)                             # scatter writes only the strided positions (resnet_v1 clears the backbone's ahead of time)
            self.c1.dgrad(dz1, n, IH, IW, dx)
            self.down.dgrad(g, n, IH, IW, dx, add=dx, ref=ref)
        else:
            self.c1.dgrad(dz1, n, IH, IW, dx, add=g, ref=ref)
        return dx


class Network(object):
    def __init__(self, batch_size=1):
        self._feat_stride = [16, ]
        self._batch_size = batch_size
        self._predictions = {}
        self._losses = {}
        self._anchor_targets = {}
        self._proposal_targets = {}
        self._mode = 'TRAIN'
        self.training = True
        self.device = 'cuda'
        self._bufs = {}
        self._buf_users = {}        # buffer key -> keys of the launch tapes that captured its address
        self.convs = []
        self._step = 0
        self.parity = None          # dict of injected sampling keys / dropout masks (tests); None = production RNG
        self.dp = None              # data-parallel gradient reducer (lang2seg_amd/parallel.py)
        self._early_op = None       # optimiser taking early partial updates during backward (optim.SGD.partial)
        self.wgq = WgradQueue(self) # weight gradients of the current backward stage, launched together by flush_wgrads()
        self.cap_projected = True   # captioner recurrence in the projected-attention form (3 launches per token)
        self._cap_state = None      # exchange state of the resident recurrence launches (csrc/cap_recur.hip), allocated on first use
        self.fuse_roialign = bool(cfg.TRAIN.get('FUSE_ROIALIGN', False))   # RoIAlign + layer4[0].conv1 + layer4[0].downsample as one launch (bf16)
        self.knockout = frozenset() # experiment only: parts of the step to leave out ('wgrad', 'wgrad3', 'wgrad4', 'cap'); set by bench.py --knockout

    # ------------------------------------------------------------------ construction
    def create_architecture(self, num_classes, tag=None, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2)):
        assert tag is not None
        self._tag = tag
        self._num_classes = num_classes
        self._anchor_scales, self._anchor_ratios = tuple(anchor_scales), tuple(anchor_ratios)
        self._num_anchors = len(anchor_scales) * len(anchor_ratios)
        self.dt = BF16 if cfg.COMPUTE_DTYPE == 'bf16' else F32
        self._init_modules()

    def buf(self, name, shape, dtype=None, zero=False):
        """persistent activation plan: one device buffer per (site, shape), allocated on first use."""
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        rec = getattr(self, '_rec_key', None)
        if rec is not None:
            self._buf_users.setdefault(key, set()).add(rec)     # the tape being recorded holds this buffer's address
        if t is None:
            td = O.TORCH_DT[self.dt] if dtype is None else dtype
            t = torch.zeros(tuple(shape), dtype=td, device=self.device)
            self._bufs[key] = t
            if zero and rec is not None:
                O.memset_zero(t)                                 # the clear has to be ON the tape even when the buffer is new
        elif zero:
            O.memset_zero(t)
        return t

    # ------------------------------------------------------------------ HIP streams
    use_streams = True

    def streams(self):
        if not hasattr(self, '_streams'):
            mk = lambda n: torch.cuda.Stream()          # (no priorities: any priority stream halves throughput on this stack)
            self._streams = dict(lang=mk('lang'), cap=mk('cap'), wg=mk('wg'), wg2=mk('wg2'), tr=mk('tr'))
            self._wg_flip = 0
        return self._streams

    def sfork(self, from_stream, to_stream):
        """device-side ordering edge between two streams (recorded on the launch tape when one is being recorded)."""
        O.stream_fork(from_stream, to_stream)

    def fork_wgrad(self, alt=False, fixed=None):
        """context: run the enclosed launches on a weight-gradient stream, ordered after everything already enqueued on
        the current stream (event fork); joined by join_wgrad() before the optimiser / gradient all-reduce.
        fixed: 'wg' / 'wg2' instead of alternating (the grouped weight-gradient launches all go to 'wg', in order)."""
        import contextlib
        if not self.use_streams:
            return contextlib.nullcontext()
        S = self.streams()
        if fixed is not None:
            name = fixed
        else:
            self._wg_flip ^= 1                               # two weight-gradient streams, alternated (small independent launches)
            name = 'wg2' if self._wg_flip else 'wg'
        self.sfork(torch.cuda.current_stream(), S[name])
        return torch.cuda.stream(S[name])

    WGRAD_WS_BYTES = 128 << 20       # (the stream-K launch of the large 3x3 problems needs 2 slabs of 192 KiB per workgroup: 101 MB)

    def wgrad_ws(self, alt=False):
        """split-K slabs of the grouped weight-gradient launches (one buffer: they all run on the 'wg' stream, in order)"""
        if alt:                                              # (A/B: a second buffer for a launch that runs on another stream)
            if getattr(self, '_wg_ws2', None) is None:
                self._wg_ws2 = torch.empty(self.WGRAD_WS_BYTES // 4, dtype=torch.float32, device=self.device)
            return self._wg_ws2
        ws = getattr(self, '_wg_ws', None)
        if ws is None:
            ws = self._wg_ws = torch.empty(self.WGRAD_WS_BYTES // 4, dtype=torch.float32, device=self.device)
        return ws

    def flush_wgrads(self, tag):
        """launch the weight gradients queued since the last flush as grouped launches (one per tile variant) on the weight-gradient
        stream, after everything enqueued so far on the current stream.  Called at the end of every backward stage; join_wgrad()
        flushes whatever is left, so no gradient can be missed."""
        return self.wgq.flush(tag)

    def join_wgrad(self):
        self.flush_wgrads('final')
        if self.use_streams:
            if getattr(self, 'stamp_buf', None) is not None:
                for k in ('wg', 'wg2'):
                    with torch.cuda.stream(self.streams()[k]):
                        self._mark('%s stream done' % k)
            self.sfork(self.streams()['wg'], torch.cuda.current_stream())
            self.sfork(self.streams()['wg2'], torch.cuda.current_stream())

    # ------------------------------------------------------------------ launch-tape replay of the whole step
    def tape_step(self, dev, train_op):
        """forward + backward + optimiser replayed from a recorded launch tape (csrc/tape.hip): the ~800 launches and the
        stream forks/joins of the step are issued by one C call instead of ~10 ms of Python; the branches still run on
        their own HIP streams.  One tape per (image size, token counts, lr, grad scale); inputs go through static buffers;
        randomness comes from the device-side step counter."""
        key = (tuple(dev['data'].shape), dev['T'], dev['S'], float(train_op.lr), float(train_op.grad_scale), self.training)
        if not hasattr(self, '_tapes'):
            import collections
            self._tapes = collections.OrderedDict()                  # least recently used first
            self._tape_hist = collections.deque(maxlen=int(getattr(self, 'tape_window', 64)))   # hit / miss of the last steps
        ent = self._tapes.get(key)
        # Real data feeds dozens of image sizes x 10-20 token counts: when most steps meet a new key, recording (two device syncs, an
        # eagerly issued step, an activation plan pinned per key) costs more than replaying saves.  Below a 50 % hit rate over the last
        # 64 steps a miss is simply run eagerly and not recorded (the keys already on tape keep replaying); recording resumes when the
        # stream of shapes settles.
        self._tape_hist.append(ent is not None)
        if ent is None and len(self._tape_hist) == self._tape_hist.maxlen and sum(self._tape_hist) < self._tape_hist.maxlen // 2 \
                and not getattr(self, 'tape_always', False):
            return self._eager_step(dev, train_op, key)
        main = torch.cuda.current_stream()
        S = self.streams()
        slist = [main, S['lang'], S['cap'], S['wg'], S['wg2'], S['tr']]
        dp = self.dp
        if ent is None:
            # a new (image size, token counts): the step is executed once, eagerly, and recorded while it runs
            while len(self._tapes) >= int(getattr(self, 'max_tapes', 64)) or \
                    (self._tapes and self.plan_bytes() > int(getattr(self, 'max_plan_bytes', 96 << 30))):
                self._evict_tape()                                   # by count and by the bytes of the activation plans the tapes pin
            st = {k: dev[k].clone() for k in ('data', 'gt_boxes', 'gt_masks', 'labels', 'cap_in', 'cap_tgt', 'cap_mask')}
            d = dict(dev); d.update(st)
            torch.cuda.synchronize()
            h = O.tape_begin(slist)
            self._tape_stages = []
            self._rec_key = key
            try:
                loss = self.forward_backward(d)                       # dp_ready() cuts the tape at every bucket hand-off
                if dp is not None:
                    O.tape_mark(); self._tape_stages.append('finish')
                    O.tape_pause(True)
                    try:
                        dp.finish()
                    finally:
                        O.tape_pause(False)
                train_op.step()
                self._mark('optimiser')
            finally:
                O.tape_end(h)
                self._rec_key = None
            stages, self._tape_stages = self._tape_stages, None
            self._tapes[key] = (h, st, loss, stages)
            return loss
        self._tapes.move_to_end(key)
        h, st, loss, stages = ent
        for k, v in st.items():
            if v.data_ptr() != dev[k].data_ptr():
                v.copy_(dev[k], non_blocking=True)
        if not stages:
            O.tape_run(h, slist)
            return loss
        # data parallel: replay segment by segment; between segments the finished gradient bucket goes to RCCL on the
        # reducer's stream (torch.distributed cannot be recorded), overlapping with the rest of the backward pass
        from ..parallel import replay_segments
        replay_segments(stages, lambda i: O.tape_run_segment(h, slist, i), dp)
        return loss

    def _eager_step(self, dev, train_op, key):
        """a step issued eagerly on a tape miss (low hit rate).  Its activation buffers are registered under a pseudo key per shape so
        that they can be dropped again: the shapes used most recently keep their plans (`max_eager_plans`), older ones are freed, and
        the byte cap of the tape plans applies to them too - otherwise one activation plan per new image size would accumulate for ever."""
        import collections
        if not hasattr(self, '_eager_keys'):
            self._eager_keys = collections.OrderedDict()
        ek = ('eager',) + key
        self._eager_keys[ek] = True
        self._eager_keys.move_to_end(ek)
        self._rec_key = ek
        try:
            loss = self.forward_backward(dev)
            if self.dp is not None:
                self.dp.finish()
            train_op.step()
        finally:
            self._rec_key = None
        cap = int(getattr(self, 'max_plan_bytes', 96 << 30))
        while len(self._eager_keys) > int(getattr(self, 'max_eager_plans', 8)) or (len(self._eager_keys) > 1 and self.plan_bytes() > cap):
            old, _ = self._eager_keys.popitem(last=False)
            self._drop_plan(old)
        return loss

    def _drop_plan(self, key):
        """free every activation buffer that only the plan `key` (a tape or an eager shape) was holding.  Steps are issued asynchronously
        and the side streams (weight gradients, optimiser) are not joined at step end, so the device is drained first: the caching
        allocator orders reuse on the allocating stream only."""
        torch.cuda.synchronize()
        for bk in [bk for bk, users in self._buf_users.items() if key in users]:
            users = self._buf_users[bk]
            users.discard(key)
            if not users:
                del self._buf_users[bk]
                self._bufs.pop(bk, None)
        for ck in [ck for ck in self.wgq.tables if ck[-1] == key]:   # problem tables of the grouped weight-gradient launches of that plan
            del self.wgq.tables[ck]

    def plan_bytes(self):
        """bytes of the persistent activation buffers (one plan per image size / token count that is on a tape)"""
        return sum(t.numel() * t.element_size() for t in self._bufs.values())

    def _evict_tape(self):
        """drop the least recently used tape and every activation buffer only it was holding (real data: one activation plan per
        image size would otherwise accumulate)"""
        key, (h, st, loss, stages) = self._tapes.popitem(last=False)
        O.tape_destroy(h)                                            # (synchronises: the tape's last replay may still be running on the side streams)
        self._drop_plan(key)

    def dp_ready(self, stage):
        """a gradient bucket is final: hand it to the data-parallel reducer (and cut the launch tape there while recording)."""
        # no bucket may be handed over (or updated early) with weight gradients of its stage still queued: the grouped launch would add
        # this rank's local dW on top of the all-reduced sum afterwards (a backbone variant that forgets its own flush_wgrads, vgg16)
        self.flush_wgrads('dp:' + stage)
        if self.dp is None:
            self._early_op.partial(stage)                        # single process: the optimiser updates the finished prefix early
            return
        if hasattr(self.dp, 'skips') and self.dp.skips(stage):   # this hand-off is not taken: the gradients ride with the next bucket, the tape is not cut
            return
        rec = getattr(self, '_tape_stages', None) is not None
        if rec:
            O.tape_mark(); self._tape_stages.append(stage)
            O.tape_pause(True)                                   # the reducer's own launches (bf16 pack / unpack) are issued by the host at
        try:                                                     # every replay, between two segments: they must not be on the tape as well
            self.dp.ready(stage)
        finally:
            if rec:
                O.tape_pause(False)

    def _mark(self, name):
        """optional phase marker (tools/phase_times.py): a timing event on the current stream."""
        pe = getattr(self, 'phase_events', None)
        if pe is not None:
            e = torch.cuda.Event(enable_timing=True); e.record(); pe.append((name, e))
        sb = getattr(self, 'stamp_buf', None)                    # tools/step_timeline.py: device clock stamps, recorded on the tape like any launch
        if sb is not None and len(self.stamp_names) < sb.numel():
            O.stamp(sb, len(self.stamp_names)); self.stamp_names.append(name)

    def seed_counter(self):
        c = getattr(self, '_seed_counter', None)
        if c is None:
            c = self._seed_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        return c

    # ------------------------------------------------------------------ hipGraph replay of the whole step
    def graph_step(self, dev, train_op):
        """forward + backward + optimiser as ONE captured graph per (image size, token counts, lr): ~1100 kernel launches
        collapse into one hipGraphLaunch.  Inputs are copied into static buffers; randomness comes from the device-side
        step counter, so every replay draws new sampling keys / dropout masks."""
        key = (tuple(dev['data'].shape), dev['T'], dev['S'], float(train_op.lr), float(train_op.grad_scale))
        ent = self._graphs.get(key) if hasattr(self, '_graphs') else None
        if not hasattr(self, '_graphs'):
            self._graphs = {}
        if ent is None:
            st = {k: dev[k].clone() for k in ('data', 'gt_boxes', 'gt_masks', 'labels', 'cap_in', 'cap_tgt', 'cap_mask')}
            d = dict(dev); d.update(st)
            # warm-up on the static buffers (allocates the activation plan), then capture
            loss = self.forward_backward(d); train_op.step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss = self.forward_backward(d); train_op.step()
            ent = self._graphs[key] = (g, st, loss)
        g, st, loss = ent
        for k, v in st.items():
            if v.data_ptr() != dev[k].data_ptr():
                v.copy_(dev[k])
        g.replay()
        return loss

    SPLITK_WS_FLOATS = 4 * 1024 * 1024

    def splitk_ws(self, need):
        """fp32 workspace for the split-K slabs of a small convolution (csrc/conv_igemm.hip: splitk_factor decides whether and how far
        to split; the slabs are added in slab order by a second launch).  One per stream: the caption branch and the main path run
        split launches at the same time."""
        if not getattr(self, 'conv_split_k', False) or need > self.SPLITK_WS_FLOATS // 2 or self.device == 'cpu':
            return None                                  # (no launch of the step asks for a split: no workspace is allocated; tools set conv_split_k)
        pool = self.__dict__.setdefault('_skws', {})
        key = torch.cuda.current_stream().cuda_stream
        ws = pool.get(key)
        if ws is None:
            ws = pool[key] = torch.empty(self.SPLITK_WS_FLOATS, dtype=torch.float32, device=self.device)
        return ws

    def _init_modules(self):
        raise NotImplementedError

    # ------------------------------------------------------------------ module-like API
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def cuda(self):
        return self

    def state_dict(self):
        self.join_update()                          # (an update may still be running on the weight-gradient stream)
        dp = getattr(self, 'dp', None)
        if dp is not None and getattr(dp, 'master_stale', False):
            # sharded data-parallel update with the shadow on the wire: the fp32 masters of the other ranks' slices are behind until they are
            # gathered - a collective, so a state_dict() taken on one rank only must be preceded by dp.gather_master() on EVERY rank
            # (model/train_val.py does that before a snapshot)
            raise RuntimeError('state_dict() of a sharded data-parallel run: call net.dp.gather_master() on every rank first')
        return self.P.state_dict()

    def load_state_dict(self, sd, strict=False):
        self.join_update()
        self.P.load_state_dict(sd, strict)
        self.refresh_weights(full=True)

    def named_parameters(self):
        """(name, fp32 tensor in the REFERENCE layout) for every trainable tensor (TV:194-220 iterates these)."""
        from .params import from_internal
        self.join_update()
        dp = getattr(self, 'dp', None)
        if dp is not None and getattr(dp, 'master_stale', False):      # as state_dict(): the masters of other ranks' shadow-gathered slices are behind
            raise RuntimeError('named_parameters() of a sharded data-parallel run: call net.dp.gather_master() on every rank first')
        for k in self.P.trainable:
            yield k, from_internal(k, self.P.view(k), self.P.shapes[k])

    def refresh_weights(self, full=False):
        """after every optimiser step: rebuild the data-gradient (transposed, BN-folded) weight copies and the fp32
        transposes of the skinny language-side matrices in ONE launch; `full` also re-casts the frozen layers."""
        if full:
            for c in self.convs:
                if not c.trainable:
                    O.weight_cast(c.w_master, c.scale, c.wf, c.Np, c.k * c.k, c.Cin)
        if getattr(self, '_tr_table', None) is None:
            from .._lib import TransposeDesc
            items = [c for c in self.convs if c.wb is not None] + list(getattr(self, 'extra_transposes', []))
            arr = (TransposeDesc * len(items))()
            for i, c in enumerate(items):
                arr[i].src, arr[i].scale, arr[i].dst = c.w_master.data_ptr(), (c.scale.data_ptr() if c.scale is not None else None), c.wb.data_ptr()
                arr[i].Cout, arr[i].taps, arr[i].Cin, arr[i].force_f32 = c.Np, getattr(c, 't_taps', c.k * c.k), getattr(c, 't_cin', c.Cin), int(getattr(c, 'force_f32', 0))
                if self.dt == BF16 and isinstance(c, ConvOp) and c.trainable and not arr[i].force_f32:
                    # the bf16 shadow the update kernel keeps current holds the same values (bf16(scale * w)): half the bytes to read
                    arr[i].src, arr[i].scale, arr[i].force_f32 = c.wf.data_ptr(), None, 2
            self._tr_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            self._tr_n = len(items)
            self._tr_tiles = sum(((c.Cin + 63) // 64) * ((c.Np + 63) // 64) * c.k * c.k for c in items)
        if self.use_streams and not full:
            # the transposed copies are first read by the NEXT step's mask head / backward: rebuild them on a side stream that
            # overlaps with the next step's backbone forward (joined by join_transposes())
            S = self.streams()
            self.sfork(torch.cuda.current_stream(), S['tr'])
            with torch.cuda.stream(S['tr']):
                O.weight_transpose_batched(self._tr_table, self._tr_n, self._tr_tiles, self.dt)
            self._tr_pending = True
        else:
            O.weight_transpose_batched(self._tr_table, self._tr_n, self._tr_tiles, self.dt)

    update_on_wg = False
    defer_heads = False      # optim.SGD.defer: the heads stage's weight gradients + their part of the update run behind the rest of the update
    conv_algo = None             # A/B: l2s_conv_desc.algo for the wide 1x1 launches issued while it is set (roi_pdma: the RoI head's N >= 1024 GEMMs on the persistent tile)
    roi_pdma = False
    layer1_fused = True          # bf16: the frozen layer1 as 4 launches (conv1 of block 0 + one fused launch per bottleneck, csrc/bottleneck_fused.hip)
    cap_persistent = True        # the captioner recurrence as one resident launch per direction where the shapes allow (rnn_size = att_hid_size = 512, <= 224 locations)
    _precleared = frozenset()    # tags of the bottleneck blocks whose scatter target (dx of a stride-2 block) this pass has cleared already, off the main queue
    prio_floor = 0               # wave priority the convolution launches get at least (raised around a latency-bound chain: cap_map_prio)
    cap_map_prio = 0             # priority of layer4's data-gradient launches on the map (caption stream), beside the RoI head's backward
    rpn_wgrad_early = False      # A/B: with rpn_bwd_early, launch the RPN's weight gradients right away instead of with the heads stage
    rpn_bwd_early = True         # RPN losses + the RPN's own backward on the language stream beside the proposal chain (False: on the main stream behind the RoI head's backward)
    join_before_layer1 = False   # A/B: the frozen layer1 also waits for the previous step's update (it then runs alone instead of beside it)
    stem_mfma = True             # bf16 mode: stem + pooling as one launch on the matrix cores (stem_mfma.hip); False: the f32 stem + pooling launches
    wgrad_overwrite = True       # grouped weight gradients write (instead of add to) a tensor's gradient at its first problem of the step
    update_clears_grad = False   # optim.SGD(keep_grad=False): the update zeroes the gradients it consumes; forward_backward does not clear
    SLOT_UPDATE_REST = 0     # event slot (csrc/tape.hip): the first part of the update (everything outside ParamStore.defer_range) is done
    SLOT_UPDATE_L2 = 1       # ... the update of the LAST segments of the flat buffer (layer2) is done: all the next step's layer2 has to wait for
    SLOT_WGRADS = 2          # ... every launch of the two weight-gradient streams (they read the previous pass's activation buffers) is done
    update_split = False     # set by optim.SGD (side-stream update, one process): layer2 joins the slot, layer3 the whole weight-gradient stream
    layer2_side = False      # set with update_split: layer2's weight gradients and their update run on 'wg2', beside the backlog of 'wg'; then
                             # SLOT_UPDATE_L2 = [layer2's weight gradients + update] and SLOT_WGRADS = [every other weight gradient of the step]

    def join_update(self, full=True, layer2_only=False):
        """the current stream waits for the optimiser update of the previous step when that ran on the weight-gradient stream
        (optim.SGD.side): called before the first launch that reads a trainable weight or writes a gradient.  Unconditional once the
        mode is on (a step recorded on a launch tape must contain the edge).
        full=False (inside the step, with deferred heads): wait only for the first part of the update - the encoder, the dynamic FCs,
        the captioner's own matrices and the backbone - which the update marks under SLOT_UPDATE_REST; the deferred weight gradients
        of the heads stage and their part of the update keep running on the weight-gradient stream until join_deferred()."""
        if self.use_streams and self.update_on_wg:
            if layer2_only and self.update_split and not self.defer_heads and not getattr(self, '_pass_without_step', False):
                # Round 5: the tail of a step is [last weight gradients -> update of layer2 -> (join of the early partial updates) -> transposes];
                # the next step's layer2 needs the second item only.  The transposes (data-gradient copies: read in backward) and the partial
                # update of layer3 (its weights are first read ~0.4 ms later) are joined before layer3 (resnet_v1._backbone_fwd).
                # (The slots are recorded by optim.SGD.step: a backward pass that was NOT followed by a step - tests, gradient checks - left weight
                # gradients on the side streams that no slot covers; the next pass then takes the whole-stream join below.)
                O.event_wait(self.SLOT_UPDATE_L2, torch.cuda.current_stream())
                if not self.layer2_side:                         # (else: join_backlog(), before the launch that overwrites layer2's output)
                    O.event_wait(self.SLOT_WGRADS, torch.cuda.current_stream())
            elif self.defer_heads and not full:
                O.event_wait(self.SLOT_UPDATE_REST, torch.cuda.current_stream())
            else:
                self.sfork(self.streams()['wg'], torch.cuda.current_stream())
                if self.layer2_side:
                    self.sfork(self.streams()['wg2'], torch.cuda.current_stream())

    def join_backlog(self):
        """layer2_side: the current stream waits for the previous step's weight gradients on 'wg' (everything but layer2's).  The first of them
        to lose a race would be layer3[0]'s, which read layer2's output: called before the launch that writes it."""
        if self.use_streams and self.update_on_wg and self.layer2_side and self.update_split and not self.defer_heads \
                and not getattr(self, '_pass_without_step', False):
            O.event_wait(self.SLOT_WGRADS, torch.cuda.current_stream())

    def join_deferred(self):
        """the current stream waits for the deferred tail of the previous step (heads-stage weight gradients, their update): called
        before the first launch that reads a weight of ParamStore.defer_range, overwrites an activation those weight gradients read, or
        writes a gradient of that range"""
        if self.use_streams and self.update_on_wg and self.defer_heads:
            self.sfork(self.streams()['wg'], torch.cuda.current_stream())

    def join_transposes(self):
        # unconditional: whether a refresh is pending is host state, and a step recorded on a launch tape must contain the edge
        if self.use_streams:
            self.sfork(self.streams()['tr'], torch.cuda.current_stream())
            self._tr_pending = False

    # ------------------------------------------------------------------ blobs
    def upload_blob(self, blobs, idx):
        """H2D of one (image, expression) pair (NET:628-648, 702-703).  Cached per blobs object so a blob that is
        already resident is not re-sent for every sentence of the same image."""
        cache = blobs.setdefault('_device', {}) if isinstance(blobs, dict) else {}
        dev = self.device
        if 'data' not in cache:
            cache['data'] = torch.from_numpy(np.ascontiguousarray(blobs['data'], dtype=np.float32)).to(dev)
        key = ('sent', idx)
        if key not in cache:
            labels = blobs['labels']
            labels = labels.cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
            lab = labels[idx:idx + 1]
            max_len = int((lab != 0).sum(1).max())                          # NET:629-630
            lab = np.ascontiguousarray(lab[:, :max_len]).astype(np.int64)
            if blobs.get('cap_labels') is not None:
                cap = np.ascontiguousarray(blobs['cap_labels'][idx:idx + 1]).astype(np.int64)
                # AttModel.py:75-93: steps i = 0.. until seq[:, i] == 0 for i >= 1
                S = 1
                while S < cap.shape[1] - 1 and cap[0, S] != 0:
                    S += 1
                cm = np.ascontiguousarray(blobs['cap_masks'][idx:idx + 1]).astype(np.float32)
            else:                                               # TEST mode / variants without a captioner
                cap = np.zeros((1, 2), np.int64); cm = np.zeros((1, 2), np.float32); S = 1
            if '_gt_masks_ref' in cache:                        # loaders/cycle_loader.py: one device mask per referred object
                r = cache['_sent_ref_host'][idx]
                gm = cache['_gt_masks_ref'][r:r + 1]
            else:
                gm = torch.from_numpy(np.ascontiguousarray(blobs['gt_masks'][idx:idx + 1], dtype=np.uint8)).to(dev)
            cache[key] = dict(
                gt_boxes=torch.from_numpy(np.ascontiguousarray(blobs['gt_boxes'][idx:idx + 1], dtype=np.float32)).to(dev),
                gt_masks=gm,
                labels=torch.from_numpy(lab[0]).to(dev), T=max_len, S=S,
                cap_in=torch.from_numpy(cap[0, :S].copy()).to(dev), cap_tgt=torch.from_numpy(cap[0, 1:S + 1].copy()).to(dev),
                cap_mask=torch.from_numpy(cm[0, 1:S + 1].copy()).to(dev))
        d = dict(cache[key])
        d['data'] = cache['data']
        d['im_info'] = np.asarray(blobs['im_info'], dtype=np.float32).reshape(-1)[:3]
        return d

    # ------------------------------------------------------------------ train step (NET:702-719)
    def train_step(self, blobs, idx, train_op):
        loss = self.train_step_async(blobs, idx, train_op)
        vals = loss.cpu().numpy()          # the single host sync of the step (the reference does seven, NET:704-710)
        if not np.isfinite(vals[np.asarray(self._loss_slots())]).all():
            self.check_cap_recur()
        return tuple(float(vals[i]) for i in self._loss_slots())

    def check_cap_recur(self):
        """The resident captioner recurrence (csrc/cap_recur.hip) poisons its outputs with NaN when a workgroup gives up its bounded spin
        (its peers were not co-resident) and leaves an error word in its exchange state.  Called when a loss comes back non-finite: the word is
        read, cleared together with the exchange state, and the cause is raised instead of a bare NaN (the caller may set
        `cap_persistent = False` to continue on the per-token launches)."""
        from .._lib import L2SError
        st = getattr(self, '_cap_state', None)
        if st is None:
            return
        torch.cuda.synchronize()
        err = [int(s[1].item()) for s in st]              # state[1]: the give-up flag (int32 words: launch count, flag, ...)
        if any(err):
            for s in st:
                s.zero_()
            raise L2SError('captioner recurrence: a workgroup of the resident %s launch gave up waiting for its peers (not co-resident); the '
                           'exchange state was reset - set Network.cap_persistent = False to run the per-token launches' %
                           ' / '.join(n for n, e in zip(('forward', 'backward'), err) if e))

    def _loss_slots(self):
        from .variants import loss_names, SLOT
        return [SLOT[k] for k in loss_names(getattr(self, 'variant', 'cycle'))]

    def _loss_names(self):
        from .variants import loss_names
        return loss_names(getattr(self, 'variant', 'cycle'))

    def train_step_async(self, blobs, idx, train_op):
        """same as train_step without the loss read-back; returns the device loss[8] buffer."""
        dev = self.upload_blob(blobs, idx)
        self._early_op = train_op if (self.dp is None and self.use_streams and self.parity is None
                                      and getattr(train_op, 'early', False) and str(self.device).startswith('cuda')) else None
        if getattr(self, 'use_tape', False) and self.use_streams and self.parity is None:
            loss = self.tape_step(dev, train_op)
        elif getattr(self, 'use_graph', False) and self.dp is None and self.parity is None:
            loss = self.graph_step(dev, train_op)
        else:
            loss = self.forward_backward(dev)
            if self.dp is not None:
                self.dp.finish()
            train_op.step()
        self._early_op = None
        self._step += 1
        return loss

    def train_step_with_summary(self, blobs, idx, train_op):
        r = self.train_step(blobs, idx, train_op)
        return r + ([(n, v) for n, v in zip(self._loss_names(), r)],)

    def get_summary(self, blobs, idx):
        was = self.training
        self.eval()
        dev = self.upload_blob(blobs, idx)
        loss = self.forward_backward(dev, backward=False).cpu().numpy()
        self.train(was)
        return [(n, float(loss[i])) for n, i in zip(self._loss_names(), self._loss_slots())]
