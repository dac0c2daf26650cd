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
# two seconds of back-to-back launches on random data first (MI355X_MICROARCH.md, DVFS item 6), replayed from a launch tape
stq = torch.cuda.current_stream()
O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, p, algo=2); torch.cuda.synchronize()
h = O.tape_begin([stq])
for _ in range(500):
    O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, p, algo=2)
O.tape_end(h)
import time
t0 = time.time()
while time.time() - t0 < 2.0:
    O.tape_run(h, [stq]); torch.cuda.synchronize()
# (1) the loop as the product runs it: the stamped build with its in-loop stamps switched off (prio bit 9), two clocks around the loop
O.tape_run(h, [stq])
O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, p, algo=4, ws=ws, prio=0x200)
torch.cuda.synchronize()
lp = ws.view(torch.int64)[2 * 24 * 8 + 48:2 * 24 * 8 + 52].cpu().tolist()
for g in range(2):
    print('group %d, no stamps inside the loop: the whole K loop took %d core cycles in %d ticks of the 100 MHz clock (%.1f us): %.2f GHz'
          % (g, lp[2 * g], lp[2 * g + 1], lp[2 * g + 1] * 0.01, lp[2 * g] / max(lp[2 * g + 1], 1) * 0.1))
# (2) with the seven stamps per slice
ws.zero_()
O.tape_run(h, [stq])
O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, p, algo=4, ws=ws)
torch.cuda.synchronize()
st = ws.view(torch.int64)[:2 * 24 * 8].cpu().view(2, 24, 8)
rt = ws.view(torch.int64)[2 * 24 * 8:2 * 24 * 8 + 48].cpu().view(2, 24)
for g in range(2):
    dc = int(st[g, 22, 0] - st[g, 2, 0]); dr = int(rt[g, 22] - rt[g, 2])
    print('group %d: slices 2 .. 22 took %d s_memtime counts in %d ticks of the 100 MHz clock: the loop runs at %.2f GHz' % (g, dc, dr, dc / max(dr, 1) * 0.1))
names = ['top', 'reads issued', 'dma issued', 'lgkm0', 'barrier->M', 'mfma issued', 'vm wait', 'barrier->L']
for g in range(2):
    print('group %d: cycles since previous stamp (columns: %s); last column = whole iteration' % (g, ', '.join(names[1:])))
    for t in range(2, 20):
        d = [int(st[g, t, i] - st[g, t, i - 1]) for i in range(1, 8)]
        tot = int(st[g, t + 1, 0] - st[g, t, 0])
        print('  t=%2d ' % t + ' '.join('%6d' % v for v in d) + '  | %6d' % tot)
print('group 1 top minus group 0 top (cycles): ' + ' '.join('%d' % int(st[1, t, 0] - st[0, t, 0]) for t in range(2, 12)))
