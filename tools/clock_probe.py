#!/usr/bin/env python
"""The core clock at the phase marks of the replayed train step (tools build of the library: tools/build_tools_lib.py): a one-wave launch
with a fixed chain of dependent integer operations at every Network._mark site, timed on the constant 100 MHz device clock.  The same
probe alone on an idle device gives the reference.  Why: the small convolutions of the backbone take 11 us per launch in the step and
8 us in tools/conv_bench.py / tools/cold_weights_bench.py."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lang2seg_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'liblang2seg_hip_tools.so')
from lang2seg_amd import ops as O
from lang2seg_amd.model.config import cfg
from lang2seg_amd.nets.resnet_v1 import resnetv1
from lang2seg_amd.nets.network import Network
from lang2seg_amd.optim import SGD
from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader

lib = _lib.load()
lib.l2s_tools_clock_probe.argtypes = [C.c_void_p, C.c_void_p]
buf = torch.zeros(128, 3, dtype=torch.int64, device='cuda')
names = []


def probe_alone():
    for _ in range(20):
        lib.l2s_tools_clock_probe(buf[127].data_ptr(), O.stream())
    torch.cuda.synchronize()
    return buf[127].cpu().numpy().copy()


time_alone = probe_alone()
print('alone, idle device: %d ticks of 10 ns, %d s_memtime counts for the chain' % (time_alone[0], time_alone[1]))


def mark(self, name):
    if len(names) < 120:
        lib.l2s_tools_clock_probe(buf[len(names)].data_ptr(), O.stream()); names.append(name)


Network._mark = mark
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
net.use_tape = True
acc = []
for rep in range(8):
    for i in range(8):
        net.train_step_async(blob, 0, optim)          # (the marks are recorded on the launch tape with the first step: names fills once)
    torch.cuda.synchronize()
    if rep:
        acc.append(buf[:len(names)].cpu().numpy().copy())
a = np.median(np.stack(acc), 0)
print('%-52s %8s %10s %14s' % ('mark', 'ticks', 's_memtime', 'clock / idle'))
for j, nm in enumerate(names):
    print('%-52s %8d %10d %14.2f' % (nm, a[j, 0], a[j, 1], time_alone[0] / max(a[j, 0], 1)))
