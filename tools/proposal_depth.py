#!/usr/bin/env python
"""How deep the greedy NMS scan of the BENCH step goes: boxes kept among the 12000 sorted proposals and the row of the last one, over the
first steps of training from the bench's initial weights (synthetic image, lr 0.001).  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for step in range(40):
    net.train_step_async(blob, 0, optim)
    torch.cuda.synchronize()
    if step < 5 or step % 10 == 9:
        t = net.t
        nk = int(t['proposal_n'].item())
        keep = net.buf('prop.keep', (2000,), torch.int32)
        sc = net.buf('prop.ss', (12000,), torch.float32)
        print('step %2d: kept %d; last kept row %d; sorted scores %.6f .. %.6f (distinct %d)' % (
            step, nk, int(keep[nk - 1].item()), float(sc[0]), float(sc[-1]), len(torch.unique(sc))), flush=True)
# the same problem alone (nothing else on the device): the scan's own time in this regime
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from conv_bench import timeit
from lang2seg_amd import ops as O
sb = net.buf('prop.sb', (12000, 4), torch.float32).clone()
ws = torch.empty(O.nms_workspace_bytes(12000) // 8 + 8, dtype=torch.int64, device='cuda')
keep = torch.full((2000,), -1, dtype=torch.int32, device='cuda'); num = torch.zeros(1, dtype=torch.int32, device='cuda')
tn = timeit(lambda: O.nms(sb, 12000, 0.7, 0, 2000, ws, keep, num))
print('l2s_nms alone on the last step\'s sorted boxes: %.1f us (kept %d)' % (tn * 1e6, int(num.item())))
if len(sys.argv) > 1:
    np.save(sys.argv[1], sb.cpu().numpy())
for nn, mk, what in ((12000, 1, 'stage-0 mask + one block + four stages that return at once'), (4096, 2000, 'one stage: mask + 64 blocks'),
                     (64, 2000, 'one block: two launches'), (8192, 2000, 'two stages')):
    t = timeit(lambda: O.nms(sb, nn, 0.7, 0, mk, ws, keep, num))
    print('  n %5d max_keep %4d: %.1f us (kept %d)  - %s' % (nn, mk, t * 1e6, int(num.item()), what))
