"""Helpers shared by the parity tests and `__graft_entry__.smoke()`: build a network from a state dict and translate recorded
sampling draws into the device-side parity inputs.  (Nothing in this package imports `oracle/`; the checker lives in tests/ and
in __graft_entry__.py.)"""
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
