#!/usr/bin/env python
"""In-kernel clock stamps of the LDS-DMA filter-row weight-gradient tile (csrc/conv_wgrad_dma.hip, instrumented build): where the two slots
of a slice spend their cycles.  Waves 0 (group 0) and 4 (group 1) of workgroup 0, slices 2..20 of its first pass.  GPU only."""
import sys, os, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd import _lib as L_
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_tools_lib import build
L_.LIB_PATH = build()          # the stamped / knock-out builds of the kernel exist only in the tools build of the library
from lang2seg_amd import ops as O
from lang2seg_amd.nets.network import WgradQueue
from lang2seg_amd._lib import BF16


class Net(object):
    dt = BF16; device = 'cuda'; _rec_key = None
    def fork_wgrad(self, alt=False, fixed=None): return contextlib.nullcontext()
    def wgrad_ws(self):
        if not hasattr(self, '_ws'): self._ws = torch.zeros(40 << 20, dtype=torch.float32, device='cuda')
        return self._ws


net = Net()
bf = lambda *s: (torch.randn(*s, device='cuda') * 0.1).bfloat16()
data = [(bf(n * H * W, 512), bf(n * H * W, 512), n, H, W) for (n, H, W) in ((256, 7, 7), (1, 38, 63))]
dws = [torch.zeros(512, 9 * 512, device='cuda') for _ in range(3)]


def run():
    q = WgradQueue(net)
    for dw in dws:
        for g, x, n, H, W in data:
            q.add(dw, g, x, n, H, W, 512, H, W, 512, 3, 1, 1)
    q.flush('stamps')


lib = L_.load()
for _ in range(30):
    run()
L_.tools_set('row3_form', int(sys.argv[1]) if len(sys.argv) > 1 else 8)
run()
torch.cuda.synchronize()
G = 256
st = net.wgrad_ws()[G * 3 * 3 * 128 * 128:].view(torch.int64)[:2 * 32 * 8].cpu().view(2, 32, 8)
names = ['top', 'offsets + reads + permutes issued', 'requests issued', 'lgkm0', 'vm wait + barrier -> M', 'mfma issued', 'vm wait (g0)', 'barrier -> L']
for g in range(2):
    print('group %d: cycles since the previous stamp (%s); last column = whole slice' % (g, ', '.join(names[1:])))
    for t in range(6, 12):
        d = [int(st[g, t, i] - st[g, t, i - 1]) for i in range(1, 8)]
        print('  t=%2d ' % t + ' '.join('%6d' % v for v in d) + '  | %6d' % int(st[g, t + 1, 0] - st[g, t, 0]))
print('group 1 top minus group 0 top: ' + ' '.join('%d' % int(st[1, t, 0] - st[0, t, 0]) for t in range(2, 12)))
