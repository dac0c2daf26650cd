#!/usr/bin/env python
"""How much of a tensor's f32 gradient error is the TENSOR's (summation-order noise any f32 implementation has) and how much the kernels'?
The CPU oracle (test infrastructure) run twice on one input - once on one thread, once on eight (oneDNN / MKL then sum their partial
results in another order) - relative L2 difference of every gradient, worst first.  Used for DESIGN.md section 2's note on
resnet.layer2.0.conv1.weight, the tensor with the largest f32 deviation in every variant.  CPU only, ~1 min.
    python tools/oracle_order_noise.py [variant] [relative image perturbation, e.g. 1e-7]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def grads(threads, variant, perturb=0.0):
    from oracle import weights as OW, synth as OS, net as ON
    torch.set_num_threads(threads)
    opt = OW.default_opt(vocab_size=60, seq_length=6)
    sd = OW.make_state_dict(opt, seed=3, head_gain=4.0) if variant == 'cycle' else OW.make_state_dict(opt, seed=3, head_gain=4.0, variant=variant)
    blob = OS.make_blob(160, 224, 6, 60, seed=5)
    if perturb:
        # rounding-sized input noise: what another (equally valid) f32 summation order upstream amounts to
        blob = dict(blob); blob['data'] = (blob['data'] * (1.0 + perturb * np.random.RandomState(1).randn(*blob['data'].shape))).astype(np.float32)
    ocfg = copy.deepcopy(ON.DEFAULT_CFG)
    ocfg['TRAIN'].update(dict(BATCH_SIZE=16, RPN_PRE_NMS_TOP_N=600, RPN_POST_NMS_TOP_N=100, RPN_BATCHSIZE=64))
    rs = np.random.RandomState(0)
    nA = 10 * 14 * 12
    samp = dict(rpn_fg_keys=rs.permutation(nA).astype(np.uint32), rpn_bg_keys=rs.permutation(nA).astype(np.uint32),
                roi_fg_keys=rs.permutation(100).astype(np.uint32), roi_bg_keys=rs.permutation(100).astype(np.uint32))
    onet = ON.OracleNet(sd, opt, ocfg, variant=variant)
    onet.forward_train(blob, samp)
    return {k: g.detach().double().clone() for k, g in onet.backward().items()}


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else 'cycle'
    perturb = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    a, b = grads(1, variant), grads(8 if not perturb else 1, variant, perturb)
    rows = []
    for k in a:
        d = float((a[k] - b[k]).norm() / (a[k].norm() + 1e-300))
        cancel = float(a[k].abs().sum() and a[k].norm() / a[k].abs().sum())
        rows.append((d, k, float(a[k].norm())))
    rows.sort(reverse=True)
    print('variant %s: relative L2 difference of the oracle\'s OWN f32 gradients, %s (worst 12 of %d tensors)' % (variant, '1 thread against 8' if not perturb else 'image perturbed by %g relative' % perturb, len(rows)))
    for d, k, n in rows[:12]:
        print('  %-52s %.2e   (|g| %.3e)' % (k, d, n))
    import statistics
    print('  median over all tensors %.2e' % statistics.median(r[0] for r in rows))


if __name__ == '__main__':
    main()
