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
    red = GradReducer(net, world)
    b = [red.bounds[s] for s in GradReducer.STAGES]
    ok = all(x <= y for x, y in zip(b, b[1:])) and b[-1] == P.total
    for s in ['caption', 'heads', 'language', 'layer3', 'layer2']:
        red.ready(s)
    red.finish()
    others = [torch.randn(P.total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    ok = ok and torch.allclose(P.grad, sum(others), atol=1e-6) and red.done == 0
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
