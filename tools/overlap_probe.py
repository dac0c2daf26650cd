#!/usr/bin/env python
"""Do two streams' kernels really overlap?  Same launches on one stream vs split over two streams (replayed from a tape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd import ops as O

def mk(n, H, W, Cin, Cout, k):
    M = n * H * W
    x = torch.randn(M, Cin, device='cuda').bfloat16(); w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.05).bfloat16()
    y = torch.empty(M, Cout, device='cuda', dtype=torch.bfloat16)
    return lambda: O.conv_igemm(x, w, y, n, H, W, Cin, H, W, Cout, k, k, 1, k // 2, relu=True)

def mkw(n, H, W, Cin, Cout, k):
    M = n * H * W
    x = torch.randn(M, Cin, device='cuda').bfloat16(); dy = torch.randn(M, Cout, device='cuda').bfloat16()
    dw = torch.zeros(Cout, k * k * Cin, device='cuda')
    return lambda: O.conv_wgrad(dy, x, dw, n, H, W, Cin, H, W, Cout, k, k, 1, k // 2)

def run(fa, fb, iters=20):
    s0 = torch.cuda.current_stream(); s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
    res = []
    for mode in ('serial', 'two'):
        fa(); fb(); torch.cuda.synchronize()
        h = O.tape_begin([s0, s1, s2])
        if mode == 'serial':
            for _ in range(iters):
                fa(); fb()
        else:
            O.stream_fork(s0, s1); O.stream_fork(s0, s2)
            with torch.cuda.stream(s1):
                for _ in range(iters): fa()
            with torch.cuda.stream(s2):
                for _ in range(iters): fb()
            O.stream_fork(s1, s0); O.stream_fork(s2, s0)
        O.tape_end(h)
        torch.cuda.synchronize()
        O.tape_run(h, [s0, s1, s2]); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); O.tape_run(h, [s0, s1, s2]); b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / iters * 1e3)
    return res

cases = {
    'l3 3x3 fwd || l3 3x3 fwd': (mk(1, 38, 63, 256, 256, 3), mk(1, 38, 63, 256, 256, 3)),
    'l3 3x3 fwd || l3 3x3 wgrad': (mk(1, 38, 63, 256, 256, 3), mkw(1, 38, 63, 256, 256, 3)),
    'l4r 3x3 fwd || l4r 3x3 wgrad': (mk(256, 7, 7, 512, 512, 3), mkw(256, 7, 7, 512, 512, 3)),
    'l4r 3x3 fwd || l3 3x3 fwd': (mk(256, 7, 7, 512, 512, 3), mk(1, 38, 63, 256, 256, 3)),
    'l4r 1x1 fwd || l4r 1x1 wgrad': (mk(256, 7, 7, 512, 2048, 1), mkw(256, 7, 7, 512, 2048, 1)),
}
for name, (fa, fb) in cases.items():
    s, t = run(fa, fb)
    print('%-34s serial %7.1f us/pair   two streams %7.1f us/pair' % (name, s, t))
