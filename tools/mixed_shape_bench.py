#!/usr/bin/env python
"""Train-step throughput on a stream of DIFFERENT image sizes / token counts (what real data looks like), eager issue against the
per-shape launch tapes (Network.tape_step: a new shape is recorded while it first executes, then replayed).  BASELINE config."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def main():
    from lang2seg_amd.model.config import cfg
    from lang2seg_amd.nets.resnet_v1 import resnetv1
    from lang2seg_amd.optim import SGD
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    cfg.COMPUTE_DTYPE = 'bf16'
    V = 3349
    opt = dict(vocab_size=V, word_embedding_size=512, word_vec_size=512, rnn_hidden_size=512, bidirectional=1, word_drop_out=0.5,
               rnn_drop_out=0.2, rnn_num_layers=1, rnn_type='lstm', variable_lengths=1, C4_feat_dim=1024, cap_loss_weight=1.0,
               caption_model='att2in2', input_encoding_size=512, rnn_size=512, num_layers=1, drop_prob_lm=0.5, seq_length=20,
               fc_feat_size=4096, att_feat_size=4096, att_hid_size=512)
    shapes = [(600, 800, 8), (600, 900, 12), (600, 1000, 20), (800, 600, 5), (600, 904, 9), (600, 800, 14)]     # (H, W, tokens)
    loaders = [SyntheticLoader(num_images=2, sents_per_image=1, H=h, W=w, T=t, vocab_size=V, seed=100 + i) for i, (h, w, t) in enumerate(shapes)]
    blobs = [ld.getBatch('train') for ld in loaders for _ in range(2)]
    rs = np.random.RandomState(0)
    order = rs.randint(0, len(blobs), 120)
    for tape in (0, 1):
        np.random.seed(cfg.RNG_SEED)
        net = resnetv1(opt, batch_size=1, num_layers=101)
        net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
        net.train()
        net.use_tape = bool(tape)
        optim = SGD(net, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.MOMENTUM, cfg.TRAIN.WEIGHT_DECAY)
        for b in blobs:
            net.upload_blob(b, 0)
        for i in list(range(len(blobs))) + list(order[:24]):          # every shape seen (and recorded) before the timed region
            net.train_step_async(blobs[i], 0, optim)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in order[24:]:
            net.train_step_async(blobs[i], 0, optim)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print('%s: %.1f img/s (%.2f ms/step) over %d steps of %d distinct (size, tokens) shapes' % (
            'tape ' if tape else 'eager', (len(order) - 24) / dt, dt / (len(order) - 24) * 1e3, len(order) - 24, len(shapes)))
        del net
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
