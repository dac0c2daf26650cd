#!/usr/bin/env python
"""What W_hh does the encoder's LSTM see in the first replayed VGG step?  A device copy of the tensor is taken ON THE LANGUAGE STREAM right
before the encoder forward (recorded on the tape like any launch) and compared with the tensor as it stood at the end of step 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lang2seg_amd import selftest, ops as O
from lang2seg_amd.optim import SGD
from lang2seg_amd.nets import resnet_v1 as RV
from oracle import weights as OW, synth as OS

opt = OW.default_opt(vocab_size=60, seq_length=6); opt['C4_feat_dim'] = 512
sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant='vgg')
over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
blobs = [OS.make_blob(320, 416, 6, 60, seed=5), OS.make_blob(320, 416, 6, 60, seed=6)]
orig = RV.resnetv1._encoder_fwd
def probed(self, d):
    for nm in ('param', 'mom', 'grad'):
        src = getattr(self.P, nm)
        snap = self.__dict__.setdefault('_snap_' + nm, torch.zeros_like(src))
        O.memcpy(snap, src)
    return orig(self, d)
RV.resnetv1._encoder_fwd = probed
for name, tape in (('eager', False), ('tape', True), ('tape2', True)):
    net = selftest.build_net(opt, over, 'bf16', sd, variant='vgg')
    net.use_tape = tape
    sgd = SGD(net, 1e-3, momentum=0.9, weight_decay=1e-4, keep_grad=True)
    net.train_step_async(dict(blobs[0]), 0, sgd)
    torch.cuda.synchronize(); net.join_update(); torch.cuda.synchronize()
    end0 = {nm: getattr(net.P, nm).clone() for nm in ('param', 'mom', 'grad')}
    net.train_step_async(dict(blobs[1]), 0, sgd)
    torch.cuda.synchronize(); net.join_update(); torch.cuda.synchronize()
    P = net.P
    for nm in ('param', 'mom', 'grad'):
        snap = getattr(net, '_snap_' + nm)
        bad = [(k, int((P.view(k, snap) != P.view(k, end0[nm])).sum()), int(np.prod(P.shapes[k]))) for k in P.trainable if not torch.equal(P.view(k, snap), P.view(k, end0[nm]))]
        print('%-6s %-5s seen by the encoder of step 1 vs end of step 0: %d tensors differ %s' % (name, nm, len(bad), bad[:10]))
        if nm == 'param' and bad:
            k = bad[0][0]
            a, b = P.view(k, snap), P.view(k, end0[nm])
            idx = (a != b).nonzero().flatten()
            print('        %s: flat indices %s ... %s; seen %s end0 %s' % (k, idx[:8].tolist(), idx[-4:].tolist(), a[idx[:4]].tolist(), b[idx[:4]].tolist()))
            end1 = P.view(k)
            print('        of the differing entries, equal to the END-OF-STEP-1 value: %d of %d' % (int((a[idx] == end1[idx]).sum()), len(idx)))
