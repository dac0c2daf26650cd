import sys, os, time
sys.path.insert(0, '/root/repo')
os.chdir(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.getcwd())
import numpy as np, torch
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
net.use_tape = True
for i in range(10):
    net.train_step_async(blob, 0, optim)
torch.cuda.synchronize()
hs = []
for r in range(10):
    t0 = time.perf_counter()
    net.train_step_async(blob, 0, optim)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print('one step into an empty queue: host call %.2f ms (min %.2f), until the device is done %.2f ms' % (np.median([h[0] for h in hs]), min(h[0] for h in hs), np.median([h[1] for h in hs])))
t0 = time.perf_counter()
for i in range(40):
    net.train_step_async(blob, 0, optim)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('40 steps back to back: host returned after %.2f ms per step, device done after %.2f ms per step' % ((t1 - t0) / 40 * 1e3, (t2 - t0) / 40 * 1e3))
