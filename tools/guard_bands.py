#!/usr/bin/env python
"""Out-of-bounds writes of the step's kernels: every buffer of the activation plan gets a guard band of canary bytes on both sides; after
some steps (eager, multi-stream) every band must still hold the canary.  Usage: guard_bands.py [vgg|cycle|...] [bf16|f32] [H W]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lang2seg_amd import selftest, ops as O
from lang2seg_amd.nets import network as NW
from lang2seg_amd.optim import SGD
from oracle import weights as OW, synth as OS

variant = sys.argv[1] if len(sys.argv) > 1 else 'vgg'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
H = int(sys.argv[3]) if len(sys.argv) > 3 else 320
W = int(sys.argv[4]) if len(sys.argv) > 4 else 416
GB = 16384                       # guard bytes on each side
CANARY = 0xA5
bands = {}

def guarded_buf(self, name, shape, dtype=None, zero=False):
    key = (name, tuple(shape), dtype)
    t = self._bufs.get(key)
    if t is None:
        td = O.TORCH_DT[self.dt] if dtype is None else dtype
        es = torch.empty((), dtype=td).element_size()
        n = int(np.prod(shape)) if len(shape) else 1
        nb = (n * es + 255) // 256 * 256
        raw = torch.full((GB + nb + GB,), CANARY, dtype=torch.uint8, device=self.device)
        raw[GB:GB + n * es].zero_()
        t = raw[GB:GB + n * es].view(td).view(tuple(shape))
        self._bufs[key] = t
        bands[key] = (raw, n * es, nb)
    if zero:
        O.memset_zero(t)
    return t

NW.Network.buf = guarded_buf
opt = OW.default_opt(vocab_size=60, seq_length=6)
if variant == 'vgg':
    opt['C4_feat_dim'] = 512
sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant) if variant == 'vgg' else OW.make_state_dict(opt, seed=3, head_gain=4.0)
over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
blobs = [OS.make_blob(H, W, 6, 60, seed=5), OS.make_blob(H, W, 6, 60, seed=6)]
net = selftest.build_net(opt, over, dtype, sd, variant=(variant if variant != 'cycle' else None))
sgd = SGD(net, 1e-3, momentum=0.9, weight_decay=1e-4)
for i in range(3):
    net.train_step_async(dict(blobs[i % 2]), 0, sgd)
torch.cuda.synchronize(); net.join_update(); torch.cuda.synchronize()
bad = 0
for key, (raw, nbytes, nb) in bands.items():
    lo = raw[:GB]; hi = raw[GB + nbytes:]
    blo = (lo != CANARY).nonzero().flatten(); bhi = (hi != CANARY).nonzero().flatten()
    if len(blo) or len(bhi):
        bad += 1
        print('OOB WRITE around %-28s shape %-18s: %d bytes below (nearest %d B before the start), %d bytes above (first at +%d B past the end, last +%d)' % (
            key[0], key[1], len(blo), (GB - int(blo.max())) if len(blo) else 0, len(bhi), int(bhi.min()) if len(bhi) else 0, int(bhi.max()) if len(bhi) else 0))
print('%s %s %dx%d: %d buffers, %d with damaged guard bands' % (variant, dtype, H, W, len(bands), bad))
