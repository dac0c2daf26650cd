#!/usr/bin/env python
"""Epilogue cost of the igemm kernel: same GEMM with and without the residual add / ReLU-mask operand."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd import ops as O
from tools.conv_bench import timeit
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for name, n, H, W, Cin, Cout, k in [('l4r 1x1 out', 256, 7, 7, 512, 2048, 1), ('l4r down', 256, 7, 7, 1024, 2048, 1), ('l4r 3x3', 256, 7, 7, 512, 512, 3), ('l3 1x1 out', 1, 38, 63, 256, 1024, 1), ('l3 1x1 in', 1, 38, 63, 1024, 256, 1), ('l3 3x3', 1, 38, 63, 256, 256, 3), ('l3 tiny K', 1, 38, 63, 64, 256, 1)]:
    p = k // 2
    M = n * H * W
    x = torch.randn(M, Cin, device='cuda').bfloat16()
    w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.05).bfloat16()
    y = torch.empty(M, Cout, device='cuda', dtype=torch.bfloat16)
    r = torch.randn(M, Cout, device='cuda').bfloat16()
    bias = torch.randn(Cout, device='cuda')
    flop = 2.0 * M * Cout * k * k * Cin
    res = []
    for kw in (dict(bias=bias, add=r, relu=True), dict(bias=bias, relu=True), dict(), dict(ref=r)):
        t = timeit(lambda: O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, p, tile=tile, **kw))
        res.append(t * 1e6)
    print('%-12s full %6.1f  no-add %6.1f  plain %6.1f  ref-mask %6.1f us   (plain = %.0f TF)' % (name, *res, flop / res[2] / 1e6))
