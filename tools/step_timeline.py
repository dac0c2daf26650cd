#!/usr/bin/env python
"""When each stream of the REPLAYED train step reaches its phase marks, on the device's own clock and without a profiler.

Every Network._mark() site becomes a one-thread launch that stores the 100 MHz device clock (l2s_stamp) on whatever stream is
current there; the stamps are recorded on the launch tape like any other launch, so the numbers are those of the normal replayed
step (+ ~30 one-thread launches).  Shows which stream the main queue waits for at the caption join and how long the
weight-gradient streams run past the end of the backward pass.

    python tools/step_timeline.py [--steps 40] [--knockout wgrad,cap]
"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=12, help='measured repetitions (eight pipelined steps each)')
    ap.add_argument('--knockout', default='')
    ap.add_argument('--lib', default='', help='another build of the C-ABI library (tools/ab_build.sh / tools/ab_variant2.sh)')
    ap.add_argument('--sgd-early', type=int, default=-1)
    ap.add_argument('--defer', type=int, default=-1, help='1 / 0 = optim.SGD.defer on / off')
    ap.add_argument('--wgrad-cap', type=int, default=0)
    ap.add_argument('--stem-mfma', type=int, default=-1)
    ap.add_argument('--sgd-blocks', type=int, default=0, help='persistent workgroups of the update kernel')
    args = ap.parse_args()
    if args.lib:
        from lang2seg_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
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
    if args.stem_mfma >= 0:
        net.stem_mfma = bool(args.stem_mfma)
    net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    net.train()
    if args.wgrad_cap > 0:
        from lang2seg_amd import _lib as _L2
        _L2.tools_set('wgrad_grid_cap', args.wgrad_cap)          # needs --lib <tools build> (tools/build_tools_lib.py)
    if args.sgd_blocks > 0:
        from lang2seg_amd import _lib as _L3
        _L3.tools_set('sgd_blocks', args.sgd_blocks)
    if args.defer >= 0:
        SGD.defer = bool(args.defer)
    optim = SGD(net, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.MOMENTUM, cfg.TRAIN.WEIGHT_DECAY)
    if args.sgd_early >= 0:
        optim.early = bool(args.sgd_early)
    blob = SyntheticLoader(num_images=1, sents_per_image=1, H=600, W=1000, T=T, vocab_size=V).getBatch('train')
    net.upload_blob(blob, 0)
    net.knockout = frozenset(x for x in args.knockout.split(',') if x)
    net.use_tape = True
    net.stamp_buf = torch.zeros(96, dtype=torch.int64, device='cuda')
    net.stamp_names = []
    acc = []
    for rep in range(args.steps + 1):
        # the stamps of the LAST of eight pipelined steps: the host has run ahead of the device by then, as in a training loop
        # (a step issued into an empty queue is paced by the host's ~3 us per launch for its first ~2 ms)
        for i in range(8):
            net.train_step_async(blob, 0, optim)
        torch.cuda.synchronize()
        if rep >= 1:
            acc.append(net.stamp_buf[:len(net.stamp_names)].cpu().numpy().astype(np.int64))
    a = np.stack(acc)                                    # [steps, marks] ticks of 10 ns
    names = net.stamp_names
    i0 = names.index('step start')
    rel = (a - a[:, i0:i0 + 1]) * 0.01                   # us since the step's first launch
    med = np.median(rel, 0)
    order = np.argsort(med)
    print('%-52s %10s %10s' % ('mark (stream order within the step)', 'median us', 'p90 us'))
    for j in order:
        print('%-52s %10.1f %10.1f' % (names[j], med[j], np.percentile(rel[:, j], 90)))


if __name__ == '__main__':
    main()
