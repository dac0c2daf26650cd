#!/usr/bin/env python
"""A/B of the convolution kernel families (l2s_conv_desc.algo) on the large-M shapes of the step, interleaved rounds in one process.
GPU only.  usage: dma_bench.py [rounds] [--shapes large|small] [--algos 0,1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd import ops as O
from conv_bench import timeit

SHAPES = [
    # name, n_img, H, W, Cin, Cout, k, pad
    ('l4r 3x3', 256, 7, 7, 512, 512, 3, 1),
    ('l4r 1x1 out', 256, 7, 7, 512, 2048, 1, 0),
    ('l4r 1x1 in', 256, 7, 7, 2048, 512, 1, 0),
    ('l4r 1x1 in0', 256, 7, 7, 1024, 512, 1, 0),
    ('l4r down', 256, 7, 7, 1024, 2048, 1, 0),
    ('l4r dgrad in0', 256, 7, 7, 512, 1024, 1, 0),
    ('l4r dgrad dn', 256, 7, 7, 2048, 1024, 1, 0),
    ('rpn 3x3', 1, 38, 63, 1024, 512, 3, 1),
    ('l4m 3x3', 1, 38, 63, 512, 512, 3, 1),
    ('l4m 1x1 out', 1, 38, 63, 512, 2048, 1, 0),
]


SMALL = [
    ('l3 3x3', 1, 38, 63, 256, 256, 3, 1),
    ('l3 1x1 in', 1, 38, 63, 1024, 256, 1, 0),
    ('l3 1x1 out', 1, 38, 63, 256, 1024, 1, 0),
    ('l2 3x3', 1, 75, 125, 128, 128, 3, 1),
    ('l2 1x1 in', 1, 75, 125, 512, 128, 1, 0),
    ('l2 1x1 out', 1, 75, 125, 128, 512, 1, 0),
    ('rpn 3x3', 1, 38, 63, 1024, 512, 3, 1),
    ('l4m 3x3', 1, 38, 63, 512, 512, 3, 1),
    ('l4m 1x1 in', 1, 38, 63, 2048, 512, 1, 0),
    ('l4m 1x1 out', 1, 38, 63, 512, 2048, 1, 0),
]


def main():
    global SHAPES
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('rounds', nargs='?', type=int, default=3)
    ap.add_argument('--shapes', default='large', choices=['large', 'small'])
    ap.add_argument('--algos', default='1,2', help='l2s_conv_desc.algo values to compare (0 = the plan\'s own choice)')
    args = ap.parse_args()
    if args.shapes == 'small':
        SHAPES = SMALL
    rounds = args.rounds
    algos = [int(a) for a in args.algos.split(',')]
    print('%-14s %s   (us per launch: fwd-form with bias+residual+ReLU / dgrad-form with ReLU mask; median of %d rounds)' % ('shape', ' '.join('algo%d' % a for a in algos), rounds))
    for name, n, H, W, Cin, Cout, k, p in SHAPES:
        M = n * H * W
        x = torch.randn(M, Cin, device='cuda').bfloat16()
        w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.05).bfloat16()
        y = torch.empty(M, Cout, device='cuda', dtype=torch.bfloat16)
        r = torch.randn(M, Cout, device='cuda').bfloat16()
        bias = torch.randn(Cout, device='cuda')
        flop = 2.0 * M * Cout * k * k * Cin
        res = {a: ([], []) for a in algos}
        for _ in range(rounds):
            for a in algos:
                res[a][0].append(timeit(lambda: O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, p, bias=bias, add=r, relu=True, algo=a)))
                res[a][1].append(timeit(lambda: O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, p, ref=r, algo=a)))
        med = lambda v: sorted(v)[len(v) // 2]
        print('%-14s %s' % (name, '   '.join('%6.1f/%6.1f us %5.0f/%5.0f TF' % (med(res[a][0]) * 1e6, med(res[a][1]) * 1e6, flop / med(res[a][0]) / 1e12, flop / med(res[a][1]) / 1e12) for a in algos)), flush=True)


if __name__ == '__main__':
    main()
