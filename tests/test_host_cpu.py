"""CPU-only tests of the host logic: config, anchors, loader contract, parameter layout, C-ABI
surface (the library loads and exports every symbol the header declares — no compute calls),
and the data-parallel gradient reducer over gloo with world_size 2."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_defaults_and_overrides(tmp_path):
    from lang2seg_amd.model.config import cfg, cfg_from_list, cfg_from_file
    assert cfg.TRAIN.LEARNING_RATE == 1e-4 and cfg.TRAIN.BATCH_SIZE == 256 and cfg.TRAIN.RPN_PRE_NMS_TOP_N == 12000
    assert cfg.TEST.RPN_POST_NMS_TOP_N == 300 and cfg.ANCHOR_SCALES == [4, 8, 16, 32] and cfg.POOLING_MODE == 'crop'
    cfg_from_list(['TRAIN.STEPSIZE', '[360000]', 'ANCHOR_RATIOS', '[0.5,1,2]'])
    assert cfg.TRAIN.STEPSIZE == [360000]
    with pytest.raises(AssertionError):
        cfg_from_list(['TRAIN.NOPE', '1'])
    with pytest.raises(AssertionError):
        cfg_from_list(['TRAIN.BATCH_SIZE', '1.5'])       # type-checked like config.py:383-386
    cfg_from_file(os.path.join(ROOT, 'experiments/cfgs/res101.yml'))
    assert cfg.TRAIN.SNAPSHOT_PREFIX == 'res101_mask_rcnn' and cfg['TRAIN']['DISPLAY'] == 20
    bad = tmp_path / 'bad.yml'; bad.write_text('NOT_A_KEY: 1\n')
    with pytest.raises(KeyError):
        cfg_from_file(str(bad))


def test_anchors_vs_reference_fixture():
    from lang2seg_amd.nets import anchors as A
    g = np.load(os.path.join(ROOT, 'tests/golden/ref_leaf.npz'))
    assert np.array_equal(A.base_anchors((8, 16, 32), (0.5, 1, 2)), g['anchors.base'].astype(np.float32))
    assert np.array_equal(A.all_anchors(5, 7, (4, 8, 16, 32), (0.5, 1, 2)), g['anchors.pre_5x7'])


def test_synthetic_loader_blobs_contract():
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    L = SyntheticLoader(num_images=3, sents_per_image=2, H=64, W=96, T=5, vocab_size=50)
    b = L.getBatch('train', 1)
    assert b['data'].shape == (1, 64, 96, 3) and b['data'].dtype == np.float32
    assert b['im_info'].shape == (1, 3) and list(b['im_info'][0][:2]) == [64, 96]
    assert b['gt_boxes'].shape == (2, 5) and b['gt_masks'].shape == (2, 64, 96) and b['gt_masks'].dtype == np.uint8
    assert b['labels'].shape == (2, 5) and b['cap_labels'].shape == (2, 7) and b['cap_masks'].shape == (2, 7)
    assert (b['cap_labels'][:, 0] == 0).all() and (b['cap_labels'][:, -1] == 0).all() and (b['cap_labels'][:, 1:6] == b['labels']).all()
    assert set(b['bounds'].keys()) == {'it_pos_now', 'it_max', 'wrapped'}
    for _ in range(2):
        b = L.getBatch('train', 1)
    assert b['bounds']['wrapped'] and L.iterators['train'] == 0 and len(L.perm['train']) == 3


def test_c_abi_exports_every_declared_symbol():
    from lang2seg_amd import _lib
    hdr = open(os.path.join(ROOT, 'include/lang2seg_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(l2s_\w+)\s*\(', hdr))
    assert len(declared) > 50
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), 'symbol %s declared in include/lang2seg_hip.h is not exported' % name
    assert set(_lib.SIGS.keys()) == declared, (set(_lib.SIGS.keys()) ^ declared)
    assert lib.l2s_version() >= 100


def test_product_library_has_no_tunables():
    """verdict r4 item 8: the C ABI is re-entrant per stream - no process-global A/B switch.  The product library exports no setter (every
    tunable of csrc/knobs.h is a compile-time constant there; the tools build with l2s_tools_set is a different file that only tools/ loads),
    the header has no 'tools:' entry, bench.py has at most 15 flags and none of them patches the package."""
    import subprocess
    from lang2seg_amd import _lib
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if l.strip()]
    assert 'l2s_tools_set' not in syms
    for bad in ('l2s_conv_dma256', 'l2s_conv_pdma_wgs', 'l2s_wgrad_row3_dma', 'l2s_wgrad_grid_cap', 'l2s_sgd_blocks'):
        assert bad not in syms, bad
    assert not [s_ for s_ in syms if 'knobs' in s_ or s_.startswith('g_')], [s_ for s_ in syms if 'knobs' in s_ or s_.startswith('g_')]
    hdr = open(os.path.join(ROOT, 'include/lang2seg_hip.h')).read()
    assert 'tools:' not in hdr and 'A/B' not in hdr
    with pytest.raises(_lib.L2SError):
        _lib.tools_set('sgd_blocks', 128)
    bench_src = open(os.path.join(ROOT, 'bench.py')).read()
    assert bench_src.count('ap.add_argument(') <= 15
    assert 'tools_set' not in bench_src


def test_conv_plan_table():
    """l2s_conv_plan_name (the host-side kernel choice of l2s_conv_igemm; no launch, so it runs without a GPU): the plan of every convolution
    shape of the BASELINE step, and of the other image sizes a training run meets - the rules were tuned on the 38x63 map, and a rule that
    quietly caught a neighbouring shape (an automatic split-K on 38x50 maps) once cost 7 % on 600x800 images."""
    import ctypes as C
    from lang2seg_amd import _lib
    L = _lib.load()

    def plan(n, H, W, Cin, Cout, k=1, form='fwd', ws=False, stride=1):
        d = _lib.ConvDesc()
        d.x = d.w = d.y = 1 << 20
        OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        d.n_img, d.IH, d.IW, d.Cin, d.OH, d.OW, d.Cout = n, H, W, Cin, OH, OW, Cout
        d.KH = d.KW = k; d.stride = stride; d.pad = k // 2; d.ldx = Cin; d.ldy = d.ldadd = d.ldref = Cout; d.xcd_mode = -1
        if form == 'dgrad':
            d.ref = 1 << 20
        if ws:
            d.ws = 1 << 20; d.ws_floats = 4 << 20
        return L.l2s_conv_plan_name(C.byref(d), _lib.BF16).decode()

    # the 600x1000 step: backbone chain on the 38x63 map, layer2 on 75x125, layer4 on the RoIs and on the map, RPN
    assert plan(1, 38, 63, 1024, 256) == 'igemm_ws64_kernel'                       # layer3 conv1: 152 tiles x 16 slices
    assert plan(1, 38, 63, 256, 256, 3) == 'igemm_p3_kernel<32,256>'               # layer3 conv2 and its data gradient
    assert plan(1, 38, 63, 256, 256, 3, 'dgrad') == 'igemm_p3_kernel<32,256>'
    assert plan(1, 38, 63, 256, 1024) == 'igemm_ring_kernel<64,64>'                # layer3 conv3: 608 tiles x 4 slices -> three per CU
    assert plan(1, 75, 125, 128, 128, 3) == 'igemm_p3_kernel<64,384>'              # layer2 conv2 (W + 1 > 64: the 384-row patch)
    assert plan(1, 75, 125, 128, 512) == 'igemm_ring_kernel<128,64>'               # layer2 conv3: a 128x128 grid below one round
    assert plan(256, 7, 7, 512, 512, 3) == 'igemm_dma_kernel<256,128>'             # layer4 @ RoIs conv2
    assert plan(256, 7, 7, 2048, 512) == 'igemm_dma_kernel<256,128>'               # layer4 @ RoIs conv1
    assert plan(256, 7, 7, 512, 2048) == 'igemm_dma256_kernel'                     # layer4 @ RoIs conv3: a wide plain GEMM -> the 256x256 LDS-DMA tile
    assert plan(256, 7, 7, 1024, 2048) == 'igemm_dma256_kernel'                    # layer4[0].downsample
    assert plan(256, 7, 7, 2048, 1024, form='dgrad') == 'igemm_dma256_kernel'      # ... and its data gradient (N = 1024)
    assert plan(64, 7, 7, 512, 2048) == 'igemm_ring_kernel<128,64>'                # fewer than 4096 pixels: not worth 256-row tiles
    assert plan(1, 38, 63, 512, 512, 3) == 'igemm_p3_kernel<64,256>'               # layer4 on the map conv2
    assert plan(1, 38, 63, 512, 2048) == 'igemm_ring_kernel<128,64>'               # layer4 on the map conv3
    assert plan(1, 38, 63, 1024, 512, 3) == 'igemm_p3_kernel<64,256>'              # RPN 3x3
    # other image sizes (600x800, 800x600, 480x640) and a workspace on offer: never a split, the patch tile wherever a row fits it
    for (H, W) in ((38, 50), (50, 38), (30, 40), (19, 32)):
        for (Cin, Cout, k) in ((1024, 256, 1), (256, 1024, 1), (256, 256, 3), (2048, 512, 1), (512, 2048, 1), (512, 512, 3), (1024, 512, 3)):
            for form in ('fwd', 'dgrad'):
                p = plan(1, H, W, Cin, Cout, k, form, ws=True)
                assert 'splitk' not in p and p != 'invalid', (H, W, Cin, Cout, k, form, p)
                if k == 3:
                    assert p.startswith('igemm_p3_kernel'), (H, W, Cin, Cout, p)
    # what the patch tile does not take: several images, rows wider than its patch, strided taps
    assert not plan(2, 19, 23, 64, 64, 3).startswith('igemm_p3')
    assert not plan(1, 3, 200, 64, 64, 3).startswith('igemm_p3')
    assert not plan(1, 38, 63, 256, 256, 3, stride=2).startswith('igemm_p3')


def test_param_layout_and_state_dict_keys():
    from lang2seg_amd._lib import F32
    from lang2seg_amd.nets.params import ParamStore, to_internal, from_internal
    from oracle import weights as OW, net as ON
    opt = OW.default_opt(vocab_size=37, seq_length=6)
    P = ParamStore(opt, 101, 81, 12, 1, 'cpu', F32)
    ref = OW.param_shapes(opt)
    for k, shp in ref.items():
        assert tuple(P.shapes[k]) == tuple(shp), k
    onet = ON.OracleNet.__new__(ON.OracleNet); onet.cfg = ON.DEFAULT_CFG
    for k in ref:
        assert P.is_trainable(k) == onet._is_trainable(k), k
    offs = sorted((P.offsets[k], int(np.prod(P.shapes[k]))) for k in P.trainable)
    for (o1, c1), (o2, _) in zip(offs, offs[1:]):
        assert o1 + c1 <= o2                           # no overlap
    assert all(P.offsets[k] % 4 == 0 for k in P.trainable if not k.endswith('bias') or k.startswith('caption'))
    # grouped heads are contiguous
    g = P.groups['rcnn_w']; assert P.offsets[g[1]] == P.offsets[g[0]] + 81 * 2048
    g = P.groups['dyn_w']; assert all(P.offsets[g[i + 1]] == P.offsets[g[i]] + 1024 * 1024 for i in range(6))
    t = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).view(2, 3, 4, 5)
    assert torch.equal(from_internal('x.weight', to_internal('x.weight', t).reshape(-1), (2, 3, 4, 5)), t)
    u = torch.arange(6 * 4 * 2 * 2, dtype=torch.float32).view(6, 4, 2, 2)
    assert torch.equal(from_internal('mask_up_sampling.weight', to_internal('mask_up_sampling.weight', u).reshape(-1), (6, 4, 2, 2)), u)


def _dp_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lang2seg_amd._lib import F32
    from lang2seg_amd.nets.params import ParamStore
    from lang2seg_amd.parallel import GradReducer
    from oracle import weights as OW
    opt = OW.default_opt(vocab_size=37, seq_length=6)

    class Net(object):
        pass
    net = Net(); net.P = ParamStore(opt, 50, 81, 12, 1, 'cpu', F32)
    P = net.P
    g = torch.Generator().manual_seed(100 + rank)
    P.grad.copy_(torch.randn(P.total, generator=g))
    mine = P.grad.clone()
    others = [torch.randn(P.total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    ok = True
    # every wire format x collective form: fp32 sums exact to rounding; bf16 buckets within the format's 2^-8 per addition
    for wire, algo, tol in (('fp32', 'allreduce', 1e-6), ('fp32', 'rs_ag', 1e-6), ('bf16', 'allreduce', 3e-2), ('bf16', 'rs_ag', 3e-2)):
        P.grad.copy_(mine)
        red = GradReducer(net, world, wire=wire, algo=algo)
        b = [red.bounds[s] for s in GradReducer.STAGES]
        ok = ok and all(x <= y for x, y in zip(b, b[1:])) and b[-1] == P.total
        for s in ['caption', 'heads', 'language', 'layer3:16', 'layer3:8', 'layer3', 'layer2']:
            red.ready(s)
        red.finish()
        want = sum(others)
        ok = ok and torch.allclose(P.grad, want, atol=tol * float(want.abs().max())) and red.done == 0
        if wire == 'bf16':
            ok = ok and not torch.equal(P.grad, want)          # the buckets really went through bf16
        # every rank holds the same bits afterwards (the ranks must not drift apart)
        chk = P.grad.double().sum().reshape(1).clone()
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        ok = ok and all(torch.equal(lst[0], x) for x in lst)
    # sharded update (round 4): reduce-scatter -> every rank runs the optimiser on ITS slice of the bucket -> all-gather of the WEIGHTS.
    # The optimiser here is SGD with momentum in torch (the HIP kernel's arithmetic, element by element); afterwards every rank must hold the
    # same weights, equal to the unsharded update of the summed gradients (fp32 wire: to fp32 rounding of the sum; bf16 wire: 2^-8 of it),
    # and its own slices' momentum must be the full update's.
    class TorchSGD(object):
        def __init__(self, P, lr, mu, scale):
            self.P, self.lr, self.mu, self.scale, self.ranges = P, lr, mu, scale, []
        def update_range(self, lo, hi, shadow=False):
            P = self.P
            g = P.grad[lo:hi] * self.scale
            P.mom[lo:hi] = self.mu * P.mom[lo:hi] + g
            P.param[lo:hi] -= self.lr * P.mom[lo:hi]
            if shadow:                                                   # (the HIP kernel writes dtype(rowscale * w); no BN fold in this stand-in)
                P.shadow[lo:hi] = P.param[lo:hi].to(P.shadow.dtype)
            self.ranges.append((lo, hi))
        def refresh_shadow_range(self, lo, hi):
            self.P.shadow[lo:hi] = self.P.param[lo:hi].to(self.P.shadow.dtype)
    from lang2seg_amd._lib import BF16
    from lang2seg_amd.parallel import shard_plan
    for dt in (F32, BF16):
        # F32 compute mode: every tensor is read as fp32 master somewhere -> every all-gather carries masters.  BF16: the convolution weights are
        # read through the shadow only -> their sub-buckets gather the shadow, everything else (encoder, captioner, biases, dynamic-filter FCs,
        # mask head) gathers masters (in round 5 those stayed at their initial values on the ranks that do not own them)
        net = Net(); net.P = ParamStore(opt, 50, 81, 12, 1, 'cpu', dt)
        P = net.P
        w0 = torch.randn(P.total, generator=torch.Generator().manual_seed(7))
        m0 = torch.randn(P.total, generator=torch.Generator().manual_seed(8))
        want_g = sum(others) * (1.0 / world)
        want_m = 0.9 * m0 + want_g
        want_w = w0 - 0.1 * want_m
        so_mask = torch.zeros(P.total, dtype=torch.bool)
        for lo_, hi_, so in P.shadow_only_runs():
            so_mask[lo_:hi_] = bool(so)
        ok = ok and (bool(so_mask.any()) == (dt == BF16))
        for k in P.trainable:                                                # nothing but convolution weights may ever be shadow-only
            if P.shadow_only(k):
                ok = ok and k.endswith('.weight') and not k.startswith(('rnn_encoder.', 'dynamic_fc', 'response_fc', 'mask_', 'caption_model.core',
                                                                           'caption_model.embed', 'caption_model.logit', 'caption_model.ctx2att'))
        for wire, tol in (('fp32', 1e-6), ('bf16', 3e-2)):
            P.grad.copy_(mine); P.param.copy_(w0); P.mom.copy_(m0); P.shadow.zero_()
            upd = TorchSGD(P, 0.1, 0.9, 1.0 / world)
            red = GradReducer(net, world, wire=wire, algo='rs_ag', shard_update=upd, rank=rank)
            plans = []
            lo_ = 0
            for st in ['caption', 'heads', 'language', 'layer3:16', 'layer3:8', 'layer3', 'layer2']:
                plans += shard_plan(P, lo_, red.bounds[st]); lo_ = red.bounds[st]
                red.ready(st)
            red.finish()
            scale = float(want_g.abs().max())
            stale = torch.zeros(P.total, dtype=torch.bool)
            for l, h in red.stale_master_ranges():
                stale[l:h] = True
            ok = ok and red.gather_shadow and (red.master_stale == (dt == BF16)) and (bool(stale.any()) == (dt == BF16))
            ok = ok and not bool((stale & ~so_mask).any())                 # a master may only be behind where NO kernel reads masters
            if dt == BF16:
                ok = ok and {c for _, _, c in plans} == {'shadow', 'master'} and len(plans) <= 12
            stol = 0.1 * tol * scale + 1e-6 + (2.0 ** -8 * float(want_w.abs().max()) if dt == BF16 else 0.0)
            ok = ok and torch.allclose(P.shadow.float(), want_w, atol=stol)   # the shadow is current everywhere, on every rank
            ok = ok and torch.allclose(P.param[~stale], want_w[~stale], atol=0.1 * tol * scale + 1e-6) and torch.equal(P.param[stale], w0[stale])
            lst = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(lst, P.shadow.double().sum().reshape(1).clone())
            ok = ok and all(torch.equal(lst[0], x) for x in lst)              # every rank holds the same shadow, bit for bit
            lst = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(lst, P.param[~so_mask].double().sum().reshape(1).clone())
            ok = ok and all(torch.equal(lst[0], x) for x in lst)              # ... and the same masters wherever masters are read
            red.gather_master()
            ok = ok and not red.master_stale and not red.stale_master_ranges()
            ok = ok and torch.allclose(P.param, want_w, atol=0.1 * tol * scale + 1e-6)
            own = torch.zeros(P.total, dtype=torch.bool)
            for l, h in upd.ranges:
                own[l:h] = True
            ok = ok and int(own.sum()) < P.total * 0.51 + 64 * world * len(plans)   # this rank updated about 1 / world of the elements (+ the sub-bucket tails)
            for l, h in upd.ranges:
                ok = ok and torch.allclose(P.mom[l:h], want_m[l:h], atol=tol * scale + 1e-6)
            lst = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(lst, P.param.double().sum().reshape(1).clone())
            ok = ok and all(torch.equal(lst[0], x) for x in lst)              # every rank holds the same weights, bit for bit
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs)
    assert ret.get(0) and ret.get(1)


@pytest.mark.parametrize('variant', ['baseline', 'spatial', 'response', 'cycle', 'cycle_response', 'vgg'])
def test_shard_plan_covers_every_bucket_and_never_shadows_a_master_read_tensor(variant):
    """parallel.shard_plan over the default buckets of every variant (bf16 layout): the sub-buckets tile [0, total) without gap or overlap, a
    'shadow' sub-bucket holds nothing but tensors no kernel reads as fp32 masters (ParamStore.shadow_only), and the plan keeps most of the
    convolution weights on the bf16 wire (the point of it); f32 layout: everything 'master'."""
    from lang2seg_amd._lib import F32, BF16
    from lang2seg_amd.nets.params import ParamStore
    from lang2seg_amd.parallel import GradReducer, bucket_bounds, shard_plan
    from oracle import weights as OW
    opt = OW.default_opt(vocab_size=3349, seq_length=20)
    if variant == 'vgg':
        opt['C4_feat_dim'] = 512
    for dt in (BF16, F32):
        P = ParamStore(opt, 101, 81, 12, 0 if variant == 'vgg' else 1, 'cpu', dt, variant)
        b = bucket_bounds(P)
        skip = set(GradReducer.SKIP_STAGES)
        ends = sorted({b[s_] for s_ in b if s_ not in skip} | {P.total})
        plan, lo = [], 0
        for hi in ends:
            if hi > lo:
                plan += shard_plan(P, lo, hi); lo = hi
        assert plan[0][0] == 0 and plan[-1][1] == P.total and all(x[1] == y[0] for x, y in zip(plan, plan[1:])) and all(x[1] > x[0] for x in plan)
        so = np.zeros(P.total, bool)
        for a_, b_, f in P.shadow_only_runs():
            so[a_:b_] = bool(f)
        n_shadow = 0
        for a_, b_, cls in plan:
            if cls == 'shadow':
                assert so[a_:b_].all(), (variant, a_, b_)
                n_shadow += b_ - a_
        if dt == F32:
            assert n_shadow == 0
        else:
            conv = sum(int(np.prod(P.shapes[k])) for k in P.trainable if P.shadow_only(k))
            assert n_shadow >= 0.95 * conv and len(plan) <= 12, (variant, n_shadow, conv, len(plan))


def test_dp_ready_flushes_queued_weight_gradients_first():
    """Network.dp_ready: the queued weight gradients of the stage are launched BEFORE the bucket goes to the reducer (or to the early
    optimiser update), whatever the backbone variant did or forgot"""
    from lang2seg_amd.nets.network import Network
    calls = []

    class Red(object):
        def ready(self, stage):
            calls.append(('ready', stage))

    class Early(object):
        def partial(self, stage):
            calls.append(('partial', stage))

    class Stub(object):
        _tape_stages = None
        def flush_wgrads(self, tag):
            calls.append(('flush', tag))
    st = Stub(); st.dp = Red(); st._early_op = None
    Network.dp_ready(st, 'layer3')
    st2 = Stub(); st2.dp = None; st2._early_op = Early()
    Network.dp_ready(st2, 'heads')
    assert calls == [('flush', 'dp:layer3'), ('ready', 'layer3'), ('flush', 'dp:heads'), ('partial', 'heads')]


def _dp_replay_worker(rank, world, port, ret):
    """two ranks replay a (mock) launch tape cut at the bucket hand-offs: segment i produces the gradients of bucket i, everything not
    yet produced is NaN - a bucket handed to the reducer too early, or a hand-off that is skipped, poisons or loses the sum"""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lang2seg_amd._lib import F32
    from lang2seg_amd.nets.params import ParamStore
    from lang2seg_amd.parallel import GradReducer, replay_segments
    from oracle import weights as OW
    opt = OW.default_opt(vocab_size=37, seq_length=6)

    class Net(object):
        pass
    net = Net(); net.P = ParamStore(opt, 50, 81, 12, 1, 'cpu', F32)
    P = net.P
    ok = True
    for wire, algo in (('fp32', 'allreduce'), ('bf16', 'rs_ag')):
        red = GradReducer(net, world, wire=wire, algo=algo)
        stages = ['caption', 'heads', 'language', 'layer3:16', 'layer3:8', 'layer3', 'finish']      # what Network.tape_step records
        ends = [red.bounds[s] for s in stages[:-1]] + [P.total]
        src = [torch.randn(P.total, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
        P.grad.fill_(float('nan'))
        log = []

        def run_segment(i):
            lo = 0 if i == 0 else ends[i - 1]
            if i < len(ends):
                P.grad[lo:ends[i]] = src[rank][lo:ends[i]]          # the backward stage of this segment wrote its bucket
            log.append(i)
        for _ in range(2):                                           # two replays of the same tape
            P.grad.fill_(float('nan')); del log[:]
            replay_segments(stages, run_segment, red)
            want = sum(src)
            tol = 1e-6 if wire == 'fp32' else 3e-2
            ok = ok and log == list(range(len(stages) + 1)) and red.done == 0
            ok = ok and bool(torch.isfinite(P.grad).all()) and torch.allclose(P.grad, want, atol=tol * float(want.abs().max()))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_dp_tape_segment_replay_gloo_world2():
    """Network.tape_step's data-parallel replay loop (parallel.replay_segments) with two gloo ranks: every bucket reaches the reducer
    after the segment that produces it and before the update segment, in both wire formats / collective forms."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_dp_replay_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs)
    assert ret.get(0) and ret.get(1)


def test_eval_postprocessing_vs_reference():
    """evaluation-path leaves (model/test.py, utils/mask_utils.py of the reference, run by tests/golden/make_golden.py
    leaf_eval with a scipy<=1.2 `imresize` restatement on PIL): recover_masks, class-wise box decoding, nearest GT resize."""
    from golden_util import load
    from lang2seg_amd.utils.mask_utils import recover_masks, imresize
    from lang2seg_amd.model import test as T
    g = load('leaf_eval')
    rec = recover_masks(g['rm.masks'].copy(), g['rm.rois'].copy(), 147, 220)
    assert np.array_equal(rec, g['rm.out'])
    assert np.array_equal((rec > 122.).astype(np.uint8), g['rm.bin'])
    assert np.allclose(T.bbox_transform_inv_np(g['bt.boxes'], g['bt.deltas']), g['bt.pred'], rtol=1e-6, atol=1e-4)
    assert np.array_equal(imresize(g['nn.mask'], size=(94, 147), interp='nearest'), g['nn.out'])
    assert abs(T.computeIoU_box([0, 0, 9, 9], [5, 5, 14, 14]) - 25.0 / 175.0) < 1e-12
    sc = np.zeros((4, 5), np.float32); sc[2, 3] = 0.9; sc[1, 0] = 0.99     # background column is ignored
    bx = np.arange(4 * 20, dtype=np.float32).reshape(4, 20)
    r, c, b = T.best_detection(sc, bx)
    assert (r, c) == (2, 3) and np.array_equal(b, bx[2, 12:16])


# ---------------------------------------------------------------------------------------------------------------------
# caption warm start (caption_models/__init__.py:45-51, train_cycle_2.py:69-76)
def _write_infos(path, proto=0, **over):
    import argparse, pickle
    o = dict(caption_model='att2in2', rnn_type='lstm', rnn_size=512, num_layers=1, vocab_size=37, seq_length=6)
    o.update(over)
    with open(path, 'wb') as f:
        pickle.dump({'opt': argparse.Namespace(**o), 'iter': 7, 'best_val_score': -2.5, 'perm': np.arange(5)}, f, protocol=proto)


def test_caption_infos_reader(tmp_path):
    """the no-import reader of infos-best.pkl: python-2 style protocol-0 pickles and newer ones, no foreign code executed,
    and (where the reference checkout is present) the reference's own four files against tests/golden/ref_caption_infos.json"""
    import json, pickle
    from lang2seg_amd.utils import caption_ckpt as CK
    for proto in (0, 2, 4):
        f = str(tmp_path / ('infos%d.pkl' % proto))
        _write_infos(f, proto)
        r = CK.read_infos(f)
        assert r['iter'] == 7 and r['best_val_score'] == -2.5 and 'perm' not in r
        assert r['opt'] == dict(caption_model='att2in2', rnn_type='lstm', rnn_size=512, num_layers=1, vocab_size=37, seq_length=6)
    marker = tmp_path / 'pwned'

    class Evil(object):
        def __reduce__(self):
            return (os.system, ('touch %s' % marker,))
    import argparse
    f = str(tmp_path / 'evil.pkl')
    with open(f, 'wb') as fid:
        pickle.dump({'opt': argparse.Namespace(rnn_size=3), 'x': Evil()}, fid, protocol=0)
    assert CK.read_infos(f)['opt'] == {'rnn_size': 3} and not marker.exists()
    gold = json.load(open(os.path.join(ROOT, 'tests/golden/ref_caption_infos.json')))
    assert len(gold) == 4
    if os.path.isdir('/root/reference'):
        for key, g in gold.items():
            r = CK.read_infos(os.path.join('/root/reference', key, 'infos-best.pkl'))
            assert r['iter'] == g['iter'] and r['best_val_score'] == g['best_val_score']
            for k, v in g['opt'].items():
                assert r['opt'][k] == v, (key, k)


class _FakeCapNet(object):
    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.sd = {'caption_model.embed.0.weight': torch.randn(5, 4, generator=g), 'caption_model.core.i2h.bias': torch.randn(6, generator=g),
                   'resnet.conv1.weight': torch.randn(2, 3, generator=g)}

    def state_dict(self):
        return dict(self.sd)

    def load_state_dict(self, sd, strict=False):
        self.sd = dict(sd)


def test_caption_warm_start_rules(tmp_path):
    from lang2seg_amd.utils import caption_ckpt as CK
    opt = dict(dataset_splitBy='refcoco_unc', start_from='caption_log_res5_2', caption_model='att2in2', rnn_type='lstm', rnn_size=512, num_layers=1)
    root = str(tmp_path)
    with pytest.raises(FileNotFoundError):
        CK.check_infos(opt, root)                                   # directory missing
    d = tmp_path / 'refcoco_unc' / 'caption_log_res5_2'
    d.mkdir(parents=True)
    with pytest.raises(FileNotFoundError):
        CK.check_infos(opt, root)                                   # infos-best.pkl missing
    with pytest.raises(FileNotFoundError):
        CK.load_caption_weights(_FakeCapNet(), opt, root)
    _write_infos(str(d / 'infos-best.pkl'), 0, rnn_size=1024)
    with pytest.raises(ValueError, match='rnn_size'):
        CK.check_infos(opt, root)
    _write_infos(str(d / 'infos-best.pkl'), 0)
    assert CK.check_infos(opt, root)['iter'] == 7
    assert CK.check_infos(dict(opt, start_from=None), root) is None
    net = _FakeCapNet()
    good = {'embed.0.weight': torch.full((5, 4), 2.0), 'core.i2h.bias': torch.full((6,), 3.0)}
    torch.save(dict(good, extra=torch.zeros(1)), str(d / 'model-best.pth'))
    with pytest.raises(KeyError, match='unexpected'):
        CK.load_caption_weights(net, opt, root)                     # strict, like nn.Module.load_state_dict
    torch.save({'embed.0.weight': good['embed.0.weight']}, str(d / 'model-best.pth'))
    with pytest.raises(KeyError, match='missing'):
        CK.load_caption_weights(net, opt, root)
    torch.save(dict(good, **{'core.i2h.bias': torch.zeros(7)}), str(d / 'model-best.pth'))
    with pytest.raises(ValueError, match='shape'):
        CK.load_caption_weights(net, opt, root)
    torch.save(good, str(d / 'model-best.pth'))
    before = net.sd['resnet.conv1.weight'].clone()
    assert CK.load_caption_weights(net, opt, root)
    assert torch.equal(net.sd['caption_model.embed.0.weight'], good['embed.0.weight'])
    assert torch.equal(net.sd['caption_model.core.i2h.bias'], good['core.i2h.bias']) and torch.equal(net.sd['resnet.conv1.weight'], before)


# ---------------------------------------------------------------------------------------------------------------------
# solver: missing pretrained file, data-parallel resume with per-rank sidecars
class _FakeSolverNet(object):
    _batch_size = 1
    variant = 'cycle'
    knockout = frozenset()
    dp = None

    def __init__(self):
        self.w = {'a.weight': torch.zeros(3)}
        self.ctr = torch.zeros(1, dtype=torch.int64)

    def state_dict(self):
        return dict(self.w)

    def load_state_dict(self, sd, strict=False):
        self.w = dict(sd)

    def seed_counter(self):
        return self.ctr


@pytest.mark.parametrize('variant', ['baseline', 'spatial', 'response', 'cycle', 'cycle_response', 'vgg'])
@pytest.mark.parametrize('branch', ['default', 'from_frcn'])
def test_both_branches_of_construct_graph(variant, branch):
    """tests/golden/ref_solver_tables.json: the param groups of every variant's reference solver in both branches of construct_graph() -
    cfg.TRAIN.FROM_FRCN False (the lr x 10 rule where the solver has it) and True (train_val.py:175-185: lr x GAMMA for everything but the mask
    branch) - against ParamStore.param_group and the oracle's rule."""
    import json
    from lang2seg_amd._lib import F32
    from lang2seg_amd.nets.params import ParamStore
    from lang2seg_amd.nets.variants import solver_cfg
    from oracle import weights as OW, net as ON
    T = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_solver_tables.json')))['%s/%s' % (variant, branch)]
    sc = solver_cfg(variant)
    assert (T['LEARNING_RATE'], T['GAMMA'], T['WEIGHT_DECAY'], T['DOUBLE_BIAS'], T['BIAS_DECAY'], T['momentum']) == \
        (sc.TRAIN.LEARNING_RATE, sc.TRAIN.GAMMA, sc.TRAIN.WEIGHT_DECAY, bool(sc.TRAIN.DOUBLE_BIAS), bool(sc.TRAIN.BIAS_DECAY), sc.TRAIN.MOMENTUM)
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    if variant == 'vgg':
        opt['C4_feat_dim'] = 512
    P = ParamStore(opt, 101, 81, 12, 0 if variant == 'vgg' else 1, 'cpu', F32, variant)
    ref = {k: (lr, wd) for k, lr, wd in zip(T['keys'], T['lr'], T['wd'])}
    assert set(P.trainable) == set(ref) - {'resnet.fc.weight', 'resnet.fc.bias'}
    frcn = branch == 'from_frcn'
    ocfg = dict(ON.DEFAULT_CFG['TRAIN'], FROM_FRCN=frcn)
    for k in P.trainable:
        f, wd_on = P.param_group(k, sc.TRAIN.DOUBLE_BIAS, sc.TRAIN.BIAS_DECAY, None, frcn, sc.TRAIN.GAMMA)
        assert abs(sc.TRAIN.LEARNING_RATE * f - ref[k][0]) <= 1e-12 and abs(wd_on * sc.TRAIN.WEIGHT_DECAY - ref[k][1]) <= 1e-15, (k, f, wd_on, ref[k])
        m, w = OW.param_group(variant, k, ocfg)
        assert abs(T['LEARNING_RATE'] * m - ref[k][0]) <= 1e-12 and abs(w - ref[k][1]) <= 1e-15, (k, m, w, ref[k])
    if frcn:
        assert any('mask' in k and abs(ref[k][0] - T['LEARNING_RATE'] * (2 if ('bias' in k and T['DOUBLE_BIAS']) else 1)) < 1e-15 for k in ref) or variant == 'vgg'


def test_from_snapshot_reads_the_reference_written_pair(tmp_path):
    """f3 against files the REFERENCE wrote: SolverWrapper.snapshot() of train_val_cycle.py:57-104, driven by tests/golden/make_golden.py on the
    tiny cycle network (tests/golden/ref_snapshot/: its zip structure + sidecar as written, payloads regenerated and CRC-checked record by record,
    golden_util.materialize_ref_snapshot).  The build's from_snapshot must restore from it what the reference's own from_snapshot (:106-165)
    restores - recorded in the manifest: every key by name + shape, the `[:, :-1]` partial rule (:121-124), numpy / python RNG streams, loader
    cursors and permutations, the iteration."""
    import random
    from golden_util import materialize_ref_snapshot
    from lang2seg_amd.model.train_val import SolverWrapper
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    sfile, nfile, man = materialize_ref_snapshot(str(tmp_path / 'snap'), drop_last_cin_of='rpn_net.weight')
    saved = torch.load(sfile, map_location='cpu')
    assert [e['key'] for e in man['keys']] == list(saved.keys())             # the reference's key order, names, shapes, dtypes
    for e in man['keys']:
        assert list(saved[e['key']].shape) == e['shape'] and str(saved[e['key']].dtype) == e['dtype'], e['key']

    class Net(_FakeSolverNet):
        def __init__(self):
            _FakeSolverNet.__init__(self)
            self.w = {k: torch.full(tuple(v.shape), 0.25, dtype=v.dtype) if v.dtype.is_floating_point else torch.zeros_like(v) for k, v in saved.items()}

    for spath, want in ((sfile, man['restore_full']), (man['_partial'], man['restore_partial'])):
        ld = SyntheticLoader(num_images=2, H=32, W=32, T=3, vocab_size=10)
        net = Net()
        sw = SolverWrapper(net, ld, str(tmp_path / 'o'), str(tmp_path / 't'))
        np.random.seed(1); random.seed(1)
        # (the reference's loader of this fixture has 11 train / 5 val images: from_snapshot checks the permutation against the shard it feeds)
        ld.split_ix = {'train': list(range(11)), 'val': list(range(5))}
        last = sw.from_snapshot(spath, nfile)
        assert last == want['last_snapshot_iter'] == man['iter']
        assert ld.iterators['train'] == want['iter_train'] and ld.iterators['val'] == want['iter_val']
        assert [int(x) for x in ld.perm['train']] == want['perm_train'] and [int(x) for x in ld.perm['val']] == want['perm_val']
        assert [float(x) for x in np.random.rand(3)] == want['next_np_rand'] and random.random() == want['next_py_random']
        got = net.state_dict()
        for k, v in saved.items():
            if spath != sfile and k == 'rpn_net.weight':
                continue
            assert torch.equal(got[k], v), k
        if spath != sfile:
            w = got['rpn_net.weight']
            assert torch.equal(w[:, :-1], saved['rpn_net.weight'][:, :-1]) and bool((w[:, -1] == want['rpn_net.weight.last_channel']).all())
            assert abs(float(w.double().sum()) - want['rpn_net.weight.sum']) < 1e-6 * abs(want['rpn_net.weight.sum']) + 1e-9
    assert net.ctr.item() == 0                                               # a reference-written sidecar ends at the iteration: no device RNG counter


def test_reference_reads_the_build_written_pair():
    """f3, the reverse direction: the pair the build's SolverWrapper.snapshot() wrote on the MI355X (structure + sidecar committed under
    tests/golden/build_snapshot/, written by tests/test_train_step_gpu.py::test_resume_from_reference_written_snapshot) was loaded by the
    REFERENCE's from_snapshot in the build container (tests/golden/make_golden.py read_build_snapshot; transcript
    tests/golden/build_snapshot_readback.json): every tensor of the synthetic weight set arrives, the cursors and the iteration are the
    build's, and the only keys the reference misses are BatchNorm's num_batches_tracked (a torch >= 0.4 buffer its own files never had)."""
    import json, pickle
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    r = json.load(open(os.path.join(here, 'build_snapshot_readback.json')))
    man = json.load(open(os.path.join(here, 'build_snapshot', 'manifest.json')))
    ref_man = json.load(open(os.path.join(here, 'ref_snapshot', 'manifest.json')))
    assert r['ok'] and not r['mismatching'] and r['last_snapshot_iter'] == man['iter'] == 9
    assert r['printed'][0] == 'size partially match: 0'
    lacking = int(r['printed'][1].split(':')[1])
    assert lacking == sum(1 for e in ref_man['keys'] if e['key'].endswith('num_batches_tracked')) == r['keys_of_reference_net'] - r['keys_in_file']
    assert (r['iter_train'], r['iter_val'], r['perm_train'], r['perm_val']) == (man['expect']['iter_train'], man['expect']['iter_val'],
                                                                                  man['expect']['perm_train'], man['expect']['perm_val'])
    # the build's file keeps the reference's key order / shapes / dtypes (minus those BatchNorm counters)
    assert [(e['key'], e['shape'], e['dtype']) for e in man['keys']] == [(e['key'], e['shape'], e['dtype']) for e in ref_man['keys']
                                                                          if not e['key'].endswith('num_batches_tracked')]
    with open(os.path.join(here, 'build_snapshot', man['pkl']), 'rb') as f:      # the sidecar: the reference's seven fields in its order, then one more
        st0, st1, it_tr, perm_tr, it_val, perm_val, it = [pickle.load(f) for _ in range(7)]
        extra = pickle.load(f)
    assert st0[0] == 'MT19937' and isinstance(st1, tuple) and it == 9 and isinstance(extra, int)


@pytest.mark.parametrize('tag', ['tiny', 'tiny_baseline', 'tiny_spatial', 'tiny_response', 'tiny_cycle_response', 'tiny_vgg'])
def test_segment_table_is_the_reference_solvers_param_groups(tag):
    """ParamStore.param_group / nets.variants.SOLVERS / model.config_vgg against the param groups the reference's own
    SolverWrapper.construct_graph() of each variant built (fixture keys solver.*): lr x 10 on rnn_encoder / dynamic_fc / response keys for
    baseline, spatial, response and vgg (train_val.py:193-198, train_val_response.py, train_val_vgg.py), not for the two cycle solvers;
    WEIGHT_DECAY 5e-4 and DOUBLE_BIAS for VGG (config_vgg.py:28,40)."""
    from golden_util import load, variant_of
    from lang2seg_amd._lib import F32
    from lang2seg_amd.nets.params import ParamStore
    from lang2seg_amd.nets.variants import solver_cfg, SOLVERS
    from oracle import weights as OW
    g = load(tag)
    v = variant_of(g)
    sc = solver_cfg(v)
    assert SOLVERS[v]['module'] == str(g['solver.module'])
    for k in ('LEARNING_RATE', 'MOMENTUM', 'WEIGHT_DECAY', 'GAMMA'):
        assert float(sc.TRAIN[k]) == float(g['solver.' + k]), k
    assert bool(sc.TRAIN.DOUBLE_BIAS) == bool(g['solver.DOUBLE_BIAS']) and bool(sc.TRAIN.BIAS_DECAY) == bool(g['solver.BIAS_DECAY'])
    opt = OW.default_opt(vocab_size=int(g['meta_V']), seq_length=int(g['meta_T']))
    if v == 'vgg':
        opt['C4_feat_dim'] = 512
    P = ParamStore(opt, 101, 81, 12, 0 if v == 'vgg' else 1, 'cpu', F32, v)
    ref = {str(k): (float(lr), float(wd)) for k, lr, wd in zip(g['solver.keys'], g['solver.lr'], g['solver.wd'])}
    # resnet.fc is a parameter of the reference's module that never receives a gradient (torch.optim.SGD skips it): not in the flat buffer
    assert set(P.trainable) == set(ref) - {'resnet.fc.weight', 'resnet.fc.bias'}
    for k in P.trainable:
        f, wd_on = P.param_group(k, sc.TRAIN.DOUBLE_BIAS, sc.TRAIN.BIAS_DECAY)
        assert abs(sc.TRAIN.LEARNING_RATE * f - ref[k][0]) <= 1e-12 and abs(wd_on * sc.TRAIN.WEIGHT_DECAY - ref[k][1]) <= 1e-15, (k, f, wd_on, ref[k])
    if v in ('baseline', 'spatial', 'response', 'vgg'):
        k = 'dynamic_fc.weight' if v == 'baseline' else 'dynamic_fc_0.weight'
        assert P.param_group(k, sc.TRAIN.DOUBLE_BIAS, sc.TRAIN.BIAS_DECAY)[0] == 10.0
        assert P.param_group('rnn_encoder.mlp.0.bias', sc.TRAIN.DOUBLE_BIAS, sc.TRAIN.BIAS_DECAY)[0] == (20.0 if v == 'vgg' else 10.0)
    else:
        assert P.param_group('dynamic_fc_0.weight', sc.TRAIN.DOUBLE_BIAS, sc.TRAIN.BIAS_DECAY)[0] == 1.0


def test_initialize_raises_on_missing_pretrained(tmp_path):
    from lang2seg_amd.model.train_val import SolverWrapper
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    ld = SyntheticLoader(num_images=2, H=32, W=32, T=3, vocab_size=10)
    sw = SolverWrapper(_FakeSolverNet(), ld, str(tmp_path / 'o'), str(tmp_path / 't'), pretrained_model=str(tmp_path / 'nope.pth'))
    with pytest.raises(FileNotFoundError):
        sw.initialize()
    SolverWrapper(_FakeSolverNet(), ld, str(tmp_path / 'o'), str(tmp_path / 't'), pretrained_model=None).initialize()
    net = _FakeSolverNet(); net.knockout = frozenset(['wgrad'])
    with pytest.raises(RuntimeError, match='knockout'):
        SolverWrapper(net, ld, str(tmp_path / 'o'), str(tmp_path / 't'))


def _resume_worker(rank, world, port, outdir, ret):
    import random
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lang2seg_amd.model.train_val import SolverWrapper
    from lang2seg_amd.model.config import cfg
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader

    def shard():
        ld = SyntheticLoader(num_images=5, H=32, W=32, T=3, vocab_size=10)     # odd image count: shards of 3 and 2 images
        for k in ld.split_ix:
            ld.split_ix[k] = ld.split_ix[k][rank::world]
            ld.perm[k] = np.arange(len(ld.split_ix[k]))
        return ld
    np.random.seed(cfg.RNG_SEED + rank); random.seed(100 + rank)
    ld = shard()
    for _ in range(4 + rank):                               # the ranks' cursors and permutations drift apart
        ld.getBatch('train', 1)
    net = _FakeSolverNet(); net.ctr.fill_(11 + rank)
    sw = SolverWrapper(net, ld, outdir, outdir, rank=rank, world=world)
    sw.snapshot(8)
    expect_next = np.random.rand()                          # the draw that follows the snapshot
    state = (ld.iterators['train'], ld.perm['train'].copy())
    dist.barrier()
    n, nfiles, sfiles = sw.find_previous()
    ok = n == 1 and all('.rank' not in f for f in nfiles)
    # resume in a fresh loader / solver
    np.random.seed(999); random.seed(999)
    ld2 = shard()
    net2 = _FakeSolverNet()
    sw2 = SolverWrapper(net2, ld2, outdir, outdir, rank=rank, world=world)
    last = sw2.from_snapshot(sfiles[-1], nfiles[-1])
    ok = ok and last == 8 and ld2.iterators['train'] == state[0] and np.array_equal(ld2.perm['train'], state[1])
    ok = ok and len(ld2.perm['train']) == len(ld2.split_ix['train']) and int(net2.ctr.item()) == 11 + rank
    ok = ok and np.random.rand() == expect_next
    for _ in range(7):                                      # keeps walking its own shard without running past its end
        ld2.getBatch('train', 1)
    from lang2seg_amd.model.config import cfg_from_list
    pre = os.path.join(outdir, cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_8')
    # (a) rank 1's sidecar is missing (a snapshot written by a one-rank run): the refusal is decided TOGETHER - every rank raises, nobody is
    # left waiting in a collective - unless TRAIN.ALLOW_RESHARD_RESUME (a declared key: settable from a config file / the command line)
    dist.barrier()
    if rank == 1:
        os.replace(pre + '.rank1.pkl', pre + '.rank1.hidden')
    dist.barrier()
    try:
        sw2.from_snapshot(sfiles[-1], nfiles[-1]); ok = False
    except ValueError as e:
        ok = ok and 'ALLOW_RESHARD_RESUME' in str(e)
    cfg_from_list(['TRAIN.ALLOW_RESHARD_RESUME', 'True'])
    try:
        ok = ok and sw2.from_snapshot(sfiles[-1], nfiles[-1]) == 8
    finally:
        cfg_from_list(['TRAIN.ALLOW_RESHARD_RESUME', 'False'])
    dist.barrier()
    if rank == 1:
        os.replace(pre + '.rank1.hidden', pre + '.rank1.pkl')
    # (b) a sidecar whose permutation does not fit this rank's shard (written with another world size) must be refused on that rank,
    # not silently mis-indexed (rank 0's own file is fine: it resumes)
    if rank == 1:
        import shutil
        shutil.copy(pre + '.rank1.pkl', pre + '.rank1.keep')
        shutil.copy(pre + '.pkl', pre + '.rank1.pkl')
    dist.barrier()
    try:
        sw2.from_snapshot(sfiles[-1], nfiles[-1])
        ok = ok and rank == 0
    except ValueError:
        ok = ok and rank == 1
    dist.barrier()
    if rank == 1:
        os.replace(pre + '.rank1.keep', pre + '.rank1.pkl')
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_resume_per_rank_sidecars(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_resume_worker, args=(r, 2, port, str(tmp_path), ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs)
    assert ret.get(0) and ret.get(1)
    names = sorted(os.listdir(str(tmp_path)))
    assert any(n.endswith('_iter_8.rank1.pkl') for n in names) and any(n.endswith('_iter_8.pkl') for n in names) and any(n.endswith('_iter_8.pth') for n in names)


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` without a launcher starts the two ranks itself (a child `torch.distributed.run`, before any GPU
    call), the ranks find each other, and rank 0's single JSON line reports n_gpus = 2 with an all-reduced rank count of 2.
    (--launcher-check stops after the rendezvous; gloo stands in for RCCL on this GPU-less box.)"""
    import json, subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launcher-check'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d == {'launcher_check': True, 'n_gpus': 2, 'ranks_seen': 2, 'backend': 'gloo'}
    # a launcher that started a different number of ranks than --gpus says is an error, not a silently mislabelled line
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launcher-check'], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_no_permute_result_consumed_behind_a_younger_lds_write():
    """ISA check (tools/scan_lgkm_order.py): no kernel consumes a ds_bpermute result while a younger ds_write of the same wave may still be
    outstanding -- the instruction pattern behind round 2's sporadically wrong attention-backward sums -- and no
    kernel contains a packed fp32 VALU op (round 4: sporadically wrong low halves of v_pk_fma_f32 in the LSTM step beside other kernels,
    DESIGN.md section 4.6; the library is built with the target feature off)."""
    import shutil
    if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        pytest.skip('no hipcc')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import scan_lgkm_order
    assert scan_lgkm_order.main() == 0
