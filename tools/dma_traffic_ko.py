#!/usr/bin/env python
"""What would fewer operand bytes buy the dominant launch?  The tools build of igemm_dma_kernel<256,128> can request its A operand for tap 0 only
(l2s_conv_desc.prio bit 8: the other eight taps' requests carry the out-of-range offset, so zeros land in LDS and nothing crosses L2) - the
L2 -> LDS traffic of a patch tile that stages the input once for the nine taps, with every other instruction of the loop unchanged (results are
garbage).  layer4@RoIs 3x3 forward and data-gradient form, interleaved rounds.  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from build_tools_lib import build
from lang2seg_amd import _lib
_lib.LIB_PATH = build()
from lang2seg_amd import ops as O
from conv_bench import timeit
n, H, W, C = 256, 7, 7, 512
M = n * H * W
x = torch.randn(M, C, device='cuda').bfloat16(); w = (torch.randn(C, 9 * C, device='cuda') * 0.05).bfloat16()
y = torch.empty(M, C, device='cuda', dtype=torch.bfloat16); bias = torch.randn(C, device='cuda'); r = torch.randn(M, C, device='cuda').bfloat16()
flop = 2.0 * M * C * 9 * C
for rnd in range(3):
    for ko in (0, 1):
        tf = timeit(lambda: O.conv_igemm(x, w, y, n, H, W, C, H, W, C, 3, 3, 1, 1, bias=bias, add=r, relu=True, algo=2, prio=ko << 8))
        td = timeit(lambda: O.conv_igemm(x, w, y, n, H, W, C, H, W, C, 3, 3, 1, 1, ref=r, algo=2, prio=ko << 8))
        print('round %d  A requests %-22s fwd %.1f us (%.0f TFLOP/s)  dgrad form %.1f us' % (rnd, 'for tap 0 only (1/9)' if ko else 'for every tap', tf * 1e6, flop / tf / 1e12, td * 1e6), flush=True)
