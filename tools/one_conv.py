#!/usr/bin/env python
"""Run one conv shape repeatedly (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd import ops as O
n, H, W, Cin, Cout, k, s, p = [int(x) for x in sys.argv[1:9]]
mode = sys.argv[9] if len(sys.argv) > 9 else 'fwd'
algo = int(sys.argv[10]) if len(sys.argv) > 10 else 0
OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
M = n * OH * OW
x = torch.randn(n * H * W, Cin, device='cuda').bfloat16()
w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.05).bfloat16()
y = torch.empty(M, Cout, device='cuda', dtype=torch.bfloat16)
dy = torch.randn(M, Cout, device='cuda').bfloat16()
dw = torch.zeros(Cout, k * k * Cin, device='cuda')
bias = torch.randn(Cout, device='cuda')
for _ in range(10):
    if mode == 'fwd':
        O.conv_igemm(x, w, y, n, H, W, Cin, OH, OW, Cout, k, k, s, p, bias=bias, relu=True, algo=algo)
    elif mode == 'dgrad':            # the data-gradient form: ReLU-mask operand, no bias
        O.conv_igemm(x, w, y, n, H, W, Cin, OH, OW, Cout, k, k, s, p, ref=dy, algo=algo)
    else:
        O.conv_wgrad(dy, x, dw, n, H, W, Cin, OH, OW, Cout, k, k, s, p)
torch.cuda.synchronize()
