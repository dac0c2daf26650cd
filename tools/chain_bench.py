#!/usr/bin/env python
"""The layer3 bottleneck chain as the step runs it: DEPENDENT launches (conv1 1x1 -> conv2 3x3 -> conv3 1x1 + residual, 23 blocks; and the
mirrored data-gradient chain), replayed from a launch tape on one stream.  Unlike tools/conv_bench.py (the same launch repeated, so
consecutive launches overlap head and tail) every launch here waits for its predecessor, which is what layer1-3 forward / layer3-2 backward
look like inside the step.  GPU only.
    python tools/chain_bench.py [--lib build/ab_<rev>/liblang2seg_hip.so] [--blocks 23] [--hw 38x63] [--planes 256]"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default='')
    ap.add_argument('--blocks', type=int, default=23)
    ap.add_argument('--hw', default='38x63')
    ap.add_argument('--planes', type=int, default=256)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--algo', default='', help='conv1,conv2,conv3 algo fields (l2s_conv_desc.algo), same for the mirrored data-gradient launch')
    args = ap.parse_args()
    if args.lib:
        from lang2seg_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from lang2seg_amd import ops as O
    H, W = [int(v) for v in args.hw.split('x')]
    A1, A2, A3 = [int(v) for v in args.algo.split(',')] if args.algo else (0, 0, 0)
    P, C, NB, M = args.planes, 4 * args.planes, args.blocks, H * W
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    w1 = [(rn(P, C) / C ** 0.5).bfloat16() for _ in range(NB)]
    w2 = [(rn(P, 9 * P) / (9 * P) ** 0.5).bfloat16() for _ in range(NB)]
    w3 = [(rn(C, P) / P ** 0.5 * 0.3).bfloat16() for _ in range(NB)]
    b1 = [rn(P) * 0.1 for _ in range(NB)]; b2 = [rn(P) * 0.1 for _ in range(NB)]; b3 = [rn(C) * 0.1 for _ in range(NB)]
    x = [torch.relu(rn(M, C)).bfloat16()] + [torch.empty(M, C, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    a1 = [torch.empty(M, P, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    a2 = [torch.empty(M, P, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    plans = {}

    def fwd():
        for b in range(NB):
            O.conv_igemm(x[b], w1[b], a1[b], 1, H, W, C, H, W, P, bias=b1[b], relu=True, algo=A1); plans['conv1 fwd'] = O.LAST_PLAN
            O.conv_igemm(a1[b], w2[b], a2[b], 1, H, W, P, H, W, P, 3, 3, 1, 1, bias=b2[b], relu=True, algo=A2); plans['conv2 fwd'] = O.LAST_PLAN
            O.conv_igemm(a2[b], w3[b], x[b + 1], 1, H, W, P, H, W, C, bias=b3[b], add=x[b], relu=True, algo=A3); plans['conv3 fwd'] = O.LAST_PLAN

    # data-gradient weights [Cin][taps][Cout] (values are irrelevant for timing; same shapes as the step's transposed copies)
    w3t = [(rn(P, C) / C ** 0.5).bfloat16() for _ in range(NB)]
    w2t = [(rn(P, 9 * P) / (9 * P) ** 0.5).bfloat16() for _ in range(NB)]
    w1t = [(rn(C, P) / P ** 0.5 * 0.3).bfloat16() for _ in range(NB)]
    gr = [torch.empty(M, C, device=dev, dtype=torch.bfloat16) for _ in range(NB)] + [rn(M, C).bfloat16()]
    dz2 = [torch.empty(M, P, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    dz1 = [torch.empty(M, P, device=dev, dtype=torch.bfloat16) for _ in range(NB)]

    def bwd():
        for b in reversed(range(NB)):
            O.conv_igemm(gr[b + 1], w3t[b], dz2[b], 1, H, W, C, H, W, P, ref=a2[b], algo=A1); plans['conv3 dgrad'] = O.LAST_PLAN
            O.conv_igemm(dz2[b], w2t[b], dz1[b], 1, H, W, P, H, W, P, 3, 3, 1, 1, ref=a1[b], algo=A2); plans['conv2 dgrad'] = O.LAST_PLAN
            O.conv_igemm(dz1[b], w1t[b], gr[b], 1, H, W, P, H, W, C, add=gr[b + 1], ref=x[b], algo=A3); plans['conv1 dgrad'] = O.LAST_PLAN

    def timeit(fn):
        st = torch.cuda.current_stream()
        fn(); torch.cuda.synchronize()
        h = O.tape_begin([st]); fn(); O.tape_end(h)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            O.tape_run(h, [st]); torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); O.tape_run(h, [st]); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        return sorted(ts)[len(ts) // 2]

    tf, tb = timeit(fwd), timeit(bwd)
    flop = 2.0 * M * (P * C * 2 + 9 * P * P) * NB
    print('%s  %dx%d planes %d, %d blocks' % (args.lib or 'in-tree', H, W, P, NB))
    print('  forward chain   %7.1f us = %5.2f us per block, %5.1f TFLOP/s   %s' % (tf, tf / NB, flop / tf / 1e6, {k: v for k, v in plans.items() if 'fwd' in k}))
    print('  data-grad chain %7.1f us = %5.2f us per block, %5.1f TFLOP/s   %s' % (tb, tb / NB, flop / tb / 1e6, {k: v for k, v in plans.items() if 'dgrad' in k}))


if __name__ == '__main__':
    main()
