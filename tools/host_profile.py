#!/usr/bin/env python
"""Host-side cost of one train step: cProfile over a few eager steps + pure enqueue time."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lang2seg_amd.model.config import cfg
from lang2seg_amd.nets.resnet_v1 import resnetv1
from lang2seg_amd.optim import SGD
from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
from lang2seg_amd import _lib
T, V = 20, 3349
opt = dict(vocab_size=V, word_embedding_size=512, word_vec_size=512, rnn_hidden_size=512, bidirectional=1, word_drop_out=0.5,
           rnn_drop_out=0.2, rnn_num_layers=1, rnn_type='lstm', variable_lengths=1, C4_feat_dim=1024, cap_loss_weight=1.0,
           caption_model='att2in2', input_encoding_size=512, rnn_size=512, num_layers=1, drop_prob_lm=0.5, seq_length=T,
           fc_feat_size=4096, att_feat_size=4096, att_hid_size=512)
net = resnetv1(opt, 1, 101); net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
net.train(); optim = SGD(net, 1e-4)
loader = SyntheticLoader(num_images=2, T=T, vocab_size=V)
blobs = [loader.getBatch('train') for _ in range(2)]
for i in range(3):
    net.train_step_async(blobs[i % 2], 0, optim)
torch.cuda.synchronize()
# count launches
cnt = [0]
orig = _lib.call
def counting(name, *a):
    cnt[0] += 1
    return orig(name, *a)
import lang2seg_amd.ops as O
O.call = counting
net.train_step_async(blobs[0], 0, optim); torch.cuda.synchronize()
print('C-ABI calls per step:', cnt[0])
O.call = orig
t0 = time.time()
for i in range(10):
    net.train_step_async(blobs[i % 2], 0, optim)
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print('host enqueue ms/step %.2f   total ms/step %.2f' % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(5):
    net.train_step_async(blobs[i % 2], 0, optim)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
