#!/usr/bin/env python
"""Is the VGG step deterministic?  K steps with a real learning rate, eager and from the tape, twice each; prints which runs agree bit for bit
and, for the first pair that differs, the first step at which the gradient buffers differ and which tensors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lang2seg_amd import selftest
from lang2seg_amd.optim import SGD
from oracle import weights as OW, synth as OS

variant = sys.argv[1] if len(sys.argv) > 1 else 'vgg'
opt = OW.default_opt(vocab_size=60, seq_length=6)
if variant == 'vgg':
    opt['C4_feat_dim'] = 512
sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant) if variant == 'vgg' else OW.make_state_dict(opt, seed=3, head_gain=4.0)
over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
blobs = [OS.make_blob(320, 416, 6, 60, seed=5), OS.make_blob(320, 416, 6, 60, seed=6)]
SGD.defer = False
runs = {}
for name, tape, sync in (('eager-a', False, False), ('eager-b', False, False), ('eager-sync', False, True), ('tape-a', True, False), ('tape-b', True, False), ('tape-sync', True, True)):
    net = selftest.build_net(opt, over, 'bf16', sd, variant=variant)
    net.use_tape = tape
    sgd = SGD(net, 1e-3, momentum=0.9, weight_decay=1e-4, keep_grad=True)
    hist = []
    for i in range(6):
        net.train_step_async(dict(blobs[i % 2]), 0, sgd)
        if sync:
            torch.cuda.synchronize()
        net.join_update()
        torch.cuda.synchronize()
        hist.append((net.P.grad.clone(), net.P.param.clone(), {k: v.clone() for k, v in net.t.items() if isinstance(v, torch.Tensor)}))
    runs[name] = (net, hist)
names = list(runs)
ref = names[0]
for n in names[1:]:
    net, h = runs[n]
    h0 = runs[ref][1]
    first = None
    for i, ((g, p, t), (g0, p0, t0)) in enumerate(zip(h, h0)):
        if not torch.equal(g, g0) or not torch.equal(p, p0):
            first = i
            break
    print('%-10s vs %-8s: %s' % (n, ref, 'identical' if first is None else 'first difference at step %d' % first))
    if first is not None:
        g, p, t = h[first]; g0, p0, t0 = h0[first]
        P = net.P
        bad = [(k, int((P.view(k, g) != P.view(k, g0)).sum())) for k in P.trainable if not torch.equal(P.view(k, g), P.view(k, g0))]
        print('   gradient tensors that differ (%d of %d):' % (len(bad), len(P.trainable)), bad[:12])
        for k in sorted(t):
            if k in t0 and t[k].shape == t0[k].shape and not torch.equal(t[k], t0[k]):
                print('   activation differs:', k, int((t[k] != t0[k]).sum()), 'of', t[k].numel())
