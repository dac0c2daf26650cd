#!/usr/bin/env python
"""The forward pass of the backbone (stem, layer1 .. layer3) of the bench step ALONE on the device, replayed from its own launch tape:
the same launches on the same buffers as in the step, nothing beside them.  In the step the segment takes ~975 us (tools/step_timeline.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lang2seg_amd import ops as O
from lang2seg_amd.model.config import cfg
from lang2seg_amd.nets.resnet_v1 import resnetv1
from lang2seg_amd.optim import SGD
from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
T, V = 20, 3349
cfg.COMPUTE_DTYPE = 'bf16'
opt = dict(vocab_size=V, word_embedding_size=512, word_vec_size=512, rnn_hidden_size=512, bidirectional=1, word_drop_out=0.5,
           rnn_drop_out=0.2, rnn_num_layers=1, rnn_type='lstm', variable_lengths=1, C4_feat_dim=1024, cap_loss_weight=1.0,
           caption_model='att2in2', input_encoding_size=512, rnn_size=512, num_layers=1, drop_prob_lm=0.5, seq_length=T,
           fc_feat_size=4096, att_feat_size=4096, att_hid_size=512)
np.random.seed(cfg.RNG_SEED)
net = resnetv1(opt, batch_size=1, num_layers=101)
net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
net.train()
optim = SGD(net, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.MOMENTUM, cfg.TRAIN.WEIGHT_DECAY)
blob = SyntheticLoader(num_images=1, sents_per_image=1, H=600, W=1000, T=T, vocab_size=V).getBatch('train')
net.upload_blob(blob, 0)
cap = {}
orig = net._backbone_fwd


def grab(d, saved):
    cap['d'] = d
    return orig(d, saved)


net._backbone_fwd = grab
net.use_tape = False
for i in range(3):
    net.train_step_async(blob, 0, optim)
torch.cuda.synchronize()
net._backbone_fwd = orig
net._pass_without_step = True            # (no slot waits: whole-stream joins, and the streams are idle)
st = torch.cuda.current_stream()
S = net.streams()
streams = [st] + [S[k] for k in ('lang', 'cap', 'wg', 'wg2', 'tr')]


def fwd():
    net._backbone_fwd(cap['d'], {})


fwd(); torch.cuda.synchronize()
h = O.tape_begin(streams)
for _ in range(4):
    fwd()
O.tape_end(h)
torch.cuda.synchronize()
O.tape_run(h, streams); torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    O.tape_run(h, streams)
b.record(); torch.cuda.synchronize()
print('backbone forward alone (stem + layer1 + layer2 + layer3, %dx%d): %.1f us per pass' % (600, 1000, a.elapsed_time(b) / 20 * 1e3))
