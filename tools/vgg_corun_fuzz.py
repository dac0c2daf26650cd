#!/usr/bin/env python
"""Which main-stream launch disturbs the encoder's LSTM?  A tape of [encoder forward on the language stream || the first N layers of the VGG
backbone on the main stream] is replayed many times; the encoder's output must be the same every time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
if os.environ.get('L2S_FUZZ_LIB'):
    from lang2seg_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ['L2S_FUZZ_LIB'])
from lang2seg_amd import selftest, ops as O
from lang2seg_amd.optim import SGD
from oracle import weights as OW, synth as OS

opt = OW.default_opt(vocab_size=60, seq_length=6); opt['C4_feat_dim'] = 512
sd = OW.make_state_dict(opt, seed=3, head_gain=4.0, variant='vgg')
over = dict(BATCH_SIZE=32, RPN_PRE_NMS_TOP_N=1500, RPN_POST_NMS_TOP_N=300, RPN_BATCHSIZE=64)
blob = OS.make_blob(320, 416, 6, 60, seed=5)
net = selftest.build_net(opt, over, 'bf16', sd, variant='vgg')
dev = net.upload_blob(dict(blob), 0)
main = torch.cuda.current_stream()
S = net.streams()
slist = [main, S['lang'], S['cap'], S['wg'], S['wg2'], S['tr']]
full_plan = list(net.vgg_plan)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for nlayers in ([int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 2, 3, 5, 8, len(full_plan)]):
    net.vgg_plan = full_plan[:nlayers]
    net.t = {}
    torch.cuda.synchronize()
    h = O.tape_begin(slist)
    net._rec_key = ('fuzz', nlayers)
    try:
        net.sfork(main, S['lang'])
        with torch.cuda.stream(S['lang']):
            hidden = net._encoder_fwd(dev)
        if nlayers:
            net._backbone_fwd(dev, {})
        net.sfork(S['lang'], main)
    finally:
        O.tape_end(h); net._rec_key = None
    torch.cuda.synchronize()
    ref = hidden.clone()
    names = ['enc.emb', 'enc.x', 'enc.gates', 'enc.act', 'enc.hfull', 'enc.cfull', 'enc.gates_reverse', 'enc.hfull_reverse']
    refs = {n: [v for k, v in net._bufs.items() if k[0] == n][0].clone() for n in names}
    bad = 0; worst = 0.0; firsts = {}
    for r in range(reps):
        O.tape_run(h, slist)
        torch.cuda.synchronize()
        if not torch.equal(hidden, ref):
            bad += 1; worst = max(worst, float((hidden - ref).abs().max()))
            if bad <= 3:
                # hypotheses for the wrong pre-activation gates of the forward direction: with G = the linear's output (before the
                # recurrence added W_hh h + b_hh in place) and ref = the right value: dev - ref == ref - G (added twice), dev == G (not added)
                P = net.P
                Wih = P.view('rnn_encoder.rnn.weight_ih_l0').double().view(2048, 512); bih = P.view('rnn_encoder.rnn.bias_ih_l0').double()
                xx = [v for k, v in net._bufs.items() if k[0] == 'enc.x'][0].double()
                G = xx @ Wih.t() + bih
                cur = [v for k, v in net._bufs.items() if k[0] == 'enc.gates'][0].double(); rf = refs['enc.gates'].double()
                idx = (cur != rf).nonzero()
                for (tt, cc) in idx[:8].tolist():
                    dev_, ref_, g_ = float(cur[tt, cc]), float(rf[tt, cc]), float(G[tt, cc])
                    print('      t=%d q=%d j=%3d: dev %.6f ref %.6f linear %.6f | dev-ref %.3e  ref-linear %.3e | other rows of ref at this column: %s' % (
                        tt, cc // 512, cc % 512, dev_, ref_, g_, dev_ - ref_, ref_ - g_, [round(float(rf[r, cc]), 6) for r in range(6)]))
            for n in names:
                cur = [v for k, v in net._bufs.items() if k[0] == n][0]
                if not torch.equal(cur, refs[n]):
                    idx = (cur.flatten() != refs[n].flatten()).nonzero().flatten()
                    firsts.setdefault(n, []).append((int(idx[0]), len(idx)))
    if firsts:
        print('   buffers that differed (first flat index, count) per bad replay:', {k: v[:6] for k, v in firsts.items()})
    print('first %2d layers of the VGG backbone beside the encoder: %d of %d replays gave a different hidden state (max |diff| %.2e)' % (nlayers, bad, reps, worst))
    O.tape_destroy(h)
