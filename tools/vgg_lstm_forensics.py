#!/usr/bin/env python
"""Forensics for the first replayed VGG step: recompute the forward LSTM direction on the host from the buffers of the run (x, the weights
as they stood at the end of step 0) and see which h(t-1) reproduces the device's pre-activation gates of row 1: the h(0) of THIS step, or the
h(0) left in the buffer by the PREVIOUS step (a stale read)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lang2seg_amd import selftest, ops as O
from lang2seg_amd.optim import SGD
from oracle import weights as OW, synth as OS

opt = OW.default_opt(vocab_size=60, seq_length=6); opt['C4_feat_dim'] = 512
sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant='vgg')
over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
blobs = [OS.make_blob(320, 416, 6, 60, seed=5), OS.make_blob(320, 416, 6, 60, seed=6)]
def B(net, name):
    return [v for k, v in net._bufs.items() if k[0] == name][0]
for name, tape in (('eager', False), ('tape', True), ('tape2', True), ('tape3', True)):
    net = selftest.build_net(opt, over, 'bf16', sd, variant='vgg')
    net.use_tape = tape
    sgd = SGD(net, 1e-3, momentum=0.9, weight_decay=1e-4, keep_grad=True)
    net.train_step_async(dict(blobs[0]), 0, sgd)
    torch.cuda.synchronize(); net.join_update(); torch.cuda.synchronize()
    P = net.P
    W = {k: P.view('rnn_encoder.rnn.' + k).clone().double().cpu() for k in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')}
    h_old = B(net, 'enc.hfull').clone().double().cpu(); c_old = B(net, 'enc.cfull').clone().double().cpu()
    net.train_step_async(dict(blobs[1]), 0, sgd)
    torch.cuda.synchronize(); net.join_update(); torch.cuda.synchronize()
    x = B(net, 'enc.x').double().cpu(); g = B(net, 'enc.gates').double().cpu(); h = B(net, 'enc.hfull').double().cpu(); c = B(net, 'enc.cfull').double().cpu()
    Hh = 512
    G = x @ W['weight_ih_l0'].view(4 * Hh, Hh).t() + W['bias_ih_l0']
    Whh = W['weight_hh_l0'].view(4 * Hh, Hh)
    out = []
    for t in range(6):
        res = {}
        for tag, hp in (('h(t-1) of this step', h[t]), ('h(t-1) left by the previous step', h_old[t])):
            pre = G[t] + Whh @ hp + W['bias_hh_l0']
            res[tag] = float((pre - g[t]).abs().max())
        out.append(res)
    print(name)
    # pattern of the wrong pre-activations at t = 1: which (gate q, unit j), and is the error explained by 128-byte lines of h(0) that were
    # read STALE (the previous step's values) by that unit's wave?
    t = 1
    pre = G[t] + Whh @ h[t] + W['bias_hh_l0']
    err = (g[t] - pre)
    wrong = (err.abs() > 1e-5).nonzero().flatten().tolist()
    print('   t=1: %d of %d pre-activations wrong; (q, j) = %s' % (len(wrong), 4 * Hh, [(i // Hh, i % Hh) for i in wrong[:24]]))
    dh = (h_old[t] - h[t])
    expl = 0
    for i in wrong[:200]:
        contrib = (Whh[i] * dh).view(16, 32).sum(1)            # per 128-byte line of h(0)
        e = float(err[i])
        best = None
        for a_ in range(16):
            if abs(float(contrib[a_]) - e) < 1e-5 * max(1, abs(e)) + 2e-6: best = (a_,)
        if best is None:
            import itertools
            for r in (2, 3):
                for comb in itertools.combinations(range(16), r):
                    if abs(float(contrib[list(comb)].sum()) - e) < 2e-6: best = comb; break
                if best: break
        if best is not None:
            expl += 1
            if expl <= 6: print('      (q,j)=(%d,%d): error %.3e = stale lines %s of h(0)' % (i // Hh, i % Hh, e, best))
    print('   explained by <= 3 stale 128-byte lines: %d of %d' % (expl, min(len(wrong), 200)))
    for t, r in enumerate(out):
        print('   t=%d  max |host gates - device gates|: %s' % (t, '   '.join('%s: %.2e' % kv for kv in r.items())))
