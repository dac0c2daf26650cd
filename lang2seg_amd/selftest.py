"""smoke(): one small train step on cuda:0 through the HIP path, checked against the CPU oracle
(the oracle is only the checker here; see oracle/__init__.py)."""
import copy
import numpy as np
import torch


def build_net(opt, cfg_train, dtype='f32', sd=None, num_layers=101, variant='cycle'):
    from .model.config import cfg
    from .nets.resnet_v1 import resnetv1
    for k, v in cfg_train.items():
        cfg.TRAIN[k] = v
    cfg.COMPUTE_DTYPE = dtype
    if variant == 'vgg':
        from .nets.vgg16 import vgg16
        net = vgg16(opt, batch_size=1)
    else:
        net = resnetv1(opt, batch_size=1, num_layers=num_layers, variant=variant)
    net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    if sd is not None:
        net.load_state_dict(sd)
    net.train()
    return net


def parity_from_samp(samp, device='cuda'):
    u = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(device)
    p = {k: u(samp[k]) for k in ['rpn_fg_keys', 'rpn_bg_keys', 'roi_fg_keys', 'roi_bg_keys'] if samp.get(k) is not None}
    for k in ('roi_fg_keys', 'roi_bg_keys'):          # room for the appended gt rows
        if k in p:
            fill = 0 if k == 'roi_fg_keys' else -1
            p[k] = torch.cat((p[k], torch.full((8,), fill, dtype=torch.int32, device=device)))
    p['roi_bg_rand'] = torch.zeros(4096, dtype=torch.int32, device=device)
    if samp.get('forced_proposals') is not None:
        r, s = samp['forced_proposals']
        p['forced_proposals'] = (torch.from_numpy(np.ascontiguousarray(r)).to(device), torch.from_numpy(np.ascontiguousarray(s)).to(device))
    return p


def smoke():
    assert torch.cuda.is_available(), 'smoke() needs a GPU'
    from oracle import weights as OW, synth as OS, net as ON
    from .optim import SGD
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0)
    blob = OS.make_blob(160, 224, 6, 60, seed=5)
    ocfg = copy.deepcopy(ON.DEFAULT_CFG)
    over = dict(BATCH_SIZE=16, RPN_PRE_NMS_TOP_N=600, RPN_POST_NMS_TOP_N=100, RPN_BATCHSIZE=64)
    ocfg['TRAIN'].update(over)
    rs = np.random.RandomState(0)
    nA = 10 * 14 * 12
    samp = dict(rpn_fg_keys=rs.permutation(nA).astype(np.uint32), rpn_bg_keys=rs.permutation(nA).astype(np.uint32),
                roi_fg_keys=rs.permutation(100).astype(np.uint32), roi_bg_keys=rs.permutation(100).astype(np.uint32))
    net = build_net(opt, over, 'f32', sd)
    net.parity = parity_from_samp(samp)
    dev = net.upload_blob(blob, 0)
    loss = net.forward_backward(dev)
    torch.cuda.synchronize()
    lv = loss.cpu().numpy()
    # oracle on the device's own proposal list (sort/NMS order is discontinuous in the scores)
    n = int(net.t['proposal_n'].item())
    samp['forced_proposals'] = (net.t['proposal_rois'].cpu().numpy()[:n], net.t['proposal_scores'].cpu().numpy()[:n])
    onet = ON.OracleNet(sd, opt, ocfg)
    _, L = onet.forward_train(blob, samp)
    names = ['rpn_cross_entropy', 'rpn_loss_box', 'cross_entropy', 'loss_box', 'loss_mask', 'loss_caption', 'total_loss']
    for i, k in enumerate(names):
        ref = float(L[k])
        assert abs(lv[i] - ref) < 1e-3 * max(1.0, abs(ref)), (k, lv[i], ref)
    SGD(net, 1e-4).step()
    torch.cuda.synchronize()
    print('smoke ok: losses', [round(float(v), 5) for v in lv[:7]])
