#!/usr/bin/env python
"""Per-workgroup timeline of the 64x64 wave-specialised convolution tile (l2s_conv_desc.algo = 8, the instrumented build) on the layer3
shapes, two dependent launches back to back replayed from a tape: when the workgroups of the second launch start relative to the end of
the first (the launch boundary), how long the first K slice takes to land, the K loop, the epilogue.  100 MHz clock: 10 ns steps.  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lang2seg_amd import ops as O

H, W = 38, 63
ALGO = int(sys.argv[1]) if len(sys.argv) > 1 else 8          # 8: the default build with stamps; 9: the three-workgroups-per-CU build with stamps
M = H * W
CASES = [('conv1 fwd   K=1024 N=256  (bias, ReLU)', 1024, 256, 1, 'fwd'),
         ('conv3 fwd   K=256  N=1024 (bias, residual, ReLU)', 256, 1024, 1, 'res'),
         ('conv2 dgrad K=2304 N=256  (ReLU mask)', 256, 256, 3, 'mask'),
         ('conv1 dgrad K=256  N=1024 (add, ReLU mask)', 256, 1024, 1, 'addmask')]
for name, Cin, Cout, k, form in CASES:
    x = torch.randn(M, Cin, device='cuda').bfloat16()
    w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.05).bfloat16()
    y = [torch.empty(M, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(2)]
    r = torch.randn(M, Cout, device='cuda').bfloat16()
    bias = torch.randn(Cout, device='cuda')
    kw = dict(fwd=dict(bias=bias, relu=True), res=dict(bias=bias, add=r, relu=True), mask=dict(ref=r), addmask=dict(add=r, ref=r))[form]
    G = ((M + 63) // 64) * (Cout // 64)
    ws = [torch.zeros(8 * G + 64, device='cuda') for _ in range(2)]
    st = torch.cuda.current_stream()

    def pair():
        for i in range(2):
            O.conv_igemm(x, w, y[i], 1, H, W, Cin, H, W, Cout, k, k, 1, k // 2, algo=ALGO, ws=ws[i], **kw)
    for _ in range(50):
        pair()
    torch.cuda.synchronize()
    h = O.tape_begin([st]); pair(); pair(); O.tape_end(h)
    res = []
    for _ in range(7):
        O.tape_run(h, [st]); torch.cuda.synchronize()
        a = ws[0].view(torch.int64)[:4 * G].cpu().numpy().reshape(G, 4).astype(np.int64)
        b = ws[1].view(torch.int64)[:4 * G].cpu().numpy().reshape(G, 4).astype(np.int64)
        res.append((a, b))
    a, b = res[-1]
    us = lambda v: v * 0.01
    endA = a[:, 3].max()
    ent = us(b[:, 0] - endA)
    print('%s: %d workgroups' % (name, G))
    print('  boundary: first workgroup enters %.2f us after the last store of the previous launch drained; entries spread over %.2f us (median %.2f)'
          % (ent.min(), ent.max() - ent.min(), np.median(ent) - ent.min()))
    for lab, v in (('entry -> first slice landed', b[:, 1] - b[:, 0]), ('K loop', b[:, 2] - b[:, 1]), ('epilogue (stores drained)', b[:, 3] - b[:, 2]),
                   ('workgroup life', b[:, 3] - b[:, 0])):
        v = us(v)
        print('  %-28s median %5.2f us   p10 %5.2f   p90 %5.2f   max %5.2f' % (lab, np.median(v), np.percentile(v, 10), np.percentile(v, 90), v.max()))
    print('  launch span (first entry -> last drain) %.2f us; previous end -> this end %.2f us' % (us(b[:, 3].max() - b[:, 0].min()), us(b[:, 3].max() - endA)))
