#!/usr/bin/env python
"""Per-phase GPU time of the train step in single-stream eager mode (no overlap): where the main path goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd.model.config import cfg
from lang2seg_amd.nets.resnet_v1 import resnetv1
from lang2seg_amd.optim import SGD
from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
T, V = 20, 3349
opt = dict(vocab_size=V, word_embedding_size=512, word_vec_size=512, rnn_hidden_size=512, bidirectional=1, word_drop_out=0.5,
           rnn_drop_out=0.2, rnn_num_layers=1, rnn_type='lstm', variable_lengths=1, C4_feat_dim=1024, cap_loss_weight=1.0,
           caption_model='att2in2', input_encoding_size=512, rnn_size=512, num_layers=1, drop_prob_lm=0.5, seq_length=T,
           fc_feat_size=4096, att_feat_size=4096, att_hid_size=512)
net = resnetv1(opt, 1, 101); net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
net.train(); net.use_streams = False
optim = SGD(net, 1e-4)
blob = SyntheticLoader(num_images=1, T=T, vocab_size=V).getBatch('train')
for _ in range(3):
    net.train_step_async(blob, 0, optim)
torch.cuda.synchronize()
acc = {}
N = 5
for _ in range(N):
    net.phase_events = []
    net._mark('start')
    dev = net.upload_blob(blob, 0)
    net.forward_backward(dev)
    net._mark('(end fwd/bwd)')
    optim.step()
    net._mark('sgd + transposes')
    torch.cuda.synchronize()
    ev = net.phase_events
    for (n0, e0), (n1, e1) in zip(ev, ev[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
tot = 0
for k, v in acc.items():
    print('%-34s %7.3f ms' % (k, v / N)); tot += v / N
print('%-34s %7.3f ms' % ('TOTAL (single stream)', tot))
