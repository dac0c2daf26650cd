#!/usr/bin/env python
"""VGG: which buffers differ between an eagerly issued and a tape-replayed second step (first replay)?  Lists every persistent buffer of the
activation plan whose contents differ after step 1, plus the parameters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lang2seg_amd import selftest
from lang2seg_amd.optim import SGD
from oracle import weights as OW, synth as OS

variant = sys.argv[1] if len(sys.argv) > 1 else 'vgg'
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
opt = OW.default_opt(vocab_size=60, seq_length=6)
if variant == 'vgg':
    opt['C4_feat_dim'] = 512
sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant) if variant == 'vgg' else OW.make_state_dict(opt, seed=3, head_gain=4.0)
over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
blobs = [OS.make_blob(320, 416, 6, 60, seed=5), OS.make_blob(320, 416, 6, 60, seed=6)]
print('token counts', [(b['labels'] != 0).sum(1).max() for b in blobs])
snaps = {}
for name, tape in (('eager', False), ('tape', True), ('tape2', True)):
    net = selftest.build_net(opt, over, 'bf16', sd, variant=variant)
    net.use_tape = tape
    try:
        sgd = SGD(net, lr, momentum=0.9, weight_decay=1e-4, keep_grad=True)
    except TypeError:
        sgd = SGD(net, lr, momentum=0.9, weight_decay=1e-4)
    for i in range(nsteps):
        net.train_step_async(dict(blobs[i % 2]), 0, sgd)
        torch.cuda.synchronize(); net.join_update(); torch.cuda.synchronize()
    snaps[name] = ({k: v.clone() for k, v in net._bufs.items()}, net.P.param.clone(), net.P.grad.clone(), net.seed_counter().clone())
for other in ('tape', 'tape2'):
    a, b = snaps['eager'], snaps[other]
    print('==== eager vs', other, ' counters', int(a[3]), int(b[3]), ' params equal', torch.equal(a[1], b[1]), ' grads equal', torch.equal(a[2], b[2]))
    keys = [k for k in a[0] if k in b[0]]
    diff = [(k[0], tuple(k[1]), int((a[0][k] != b[0][k]).sum())) for k in keys if not torch.equal(a[0][k], b[0][k])]
    same = [k[0] for k in keys if torch.equal(a[0][k], b[0][k])]
    print('only in eager:', [k[0] for k in a[0] if k not in b[0]][:20])
    print('only in tape :', [k[0] for k in b[0] if k not in a[0]][:20])
    print('%d buffers differ, %d equal' % (len(diff), len(same)))
    for d in diff:
        print('   differs', d)
    for k in keys:
        if k[0] in ('enc.gates', 'enc.hfull', 'enc.act', 'enc.hidden', 'dyn.filt', 'enc.cfull') and not torch.equal(a[0][k], b[0][k]):
            x, y = a[0][k].float().flatten(), b[0][k].float().flatten()
            idx = (x != y).nonzero().flatten()
            print('   %s: max |diff| %.3e (max |value| %.3e); first differing flat indices %s; eager %s tape %s' % (
                k[0], float((x - y).abs().max()), float(x.abs().max()), idx[:6].tolist(), x[idx[:4]].tolist(), y[idx[:4]].tolist()))
    print('   equal:', same)
