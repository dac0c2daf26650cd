#!/usr/bin/env python
"""Sweep tile / split-K of the weight-gradient kernel on the step's shapes (us per launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd import ops as O
from tools.conv_bench import timeit, SHAPES
print('%-14s %-5s' % ('shape', 'tile') + ''.join('%8s' % ('s=%d' % s) for s in (0, 1, 2, 4, 8, 16, 32)))
for name, n, H, W, Cin, Cout, k, s, p in SHAPES:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    M = n * OH * OW
    x = torch.randn(n * H * W, Cin, device='cuda').bfloat16()
    dy = torch.randn(M, Cout, device='cuda').bfloat16()
    dw = torch.zeros(Cout, k * k * Cin, device='cuda')
    ws = torch.empty(64 << 18, device='cuda')                   # 64 MiB of split-K slabs (the launcher splits less when they do not fit)
    for tile in (64, 128):
        row = []
        for sk in (0, 1, 2, 4, 8, 16, 32):
            t = timeit(lambda: O.conv_wgrad(dy, x, dw, n, H, W, Cin, OH, OW, Cout, k, k, s, p, split_k=sk, tile=tile, ws=ws))
            row.append(t * 1e6)
        print('%-14s %-5d' % (name, tile) + ''.join('%8.1f' % v for v in row))
