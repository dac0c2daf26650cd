#!/usr/bin/env python
"""In-kernel clock stamps of the LDS-DMA convolution tile (algo 4 = the instrumented build): where a slot of the K loop spends its cycles.
GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd import ops as O

n, H, W, Cin, Cout, k, p = 256, 7, 7, 512, 512, 3, 1
if len(sys.argv) > 1:
    n, H, W, Cin, Cout, k, p = [int(v) for v in sys.argv[1:8]]
M = n * H * W
x = torch.randn(M, Cin, device='cuda').bfloat16()
w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.05).bfloat16()
y = torch.empty(M, Cout, device='cuda', dtype=torch.bfloat16)
ws = torch.zeros(1 << 16, device='cuda')
for _ in range(200):                      # warm clocks
    O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, p, algo=2)
O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, p, algo=4, ws=ws)
torch.cuda.synchronize()
st = ws.view(torch.int64)[:2 * 24 * 8].cpu().view(2, 24, 8)
names = ['top', 'reads issued', 'dma issued', 'lgkm0', 'barrier->M', 'mfma issued', 'vm wait', 'barrier->L']
for g in range(2):
    print('group %d: cycles since previous stamp (columns: %s); last column = whole iteration' % (g, ', '.join(names[1:])))
    for t in range(2, 20):
        d = [int(st[g, t, i] - st[g, t, i - 1]) for i in range(1, 8)]
        tot = int(st[g, t + 1, 0] - st[g, t, 0])
        print('  t=%2d ' % t + ' '.join('%6d' % v for v in d) + '  | %6d' % tot)
print('group 1 top minus group 0 top (cycles): ' + ' '.join('%d' % int(st[1, t, 0] - st[0, t, 0]) for t in range(2, 12)))
