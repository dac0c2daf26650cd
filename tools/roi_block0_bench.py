#!/usr/bin/env python
"""l2s_roialign_block0_fwd (RoIAlign + layer4[0].conv1 + layer4[0].downsample in one launch) against the three launches it replaces,
at the train-step size (256 RoIs on a 38x63x1024 map).  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lang2seg_amd import ops as O
from conv_bench import timeit

H, W, C, R, N1, N2, P = 38, 63, 1024, 256, 512, 2048, 7
g = torch.Generator().manual_seed(21)
fd = torch.randn(H * W, C, generator=g).cuda().bfloat16()
rs = np.random.RandomState(3)
x1 = rs.uniform(0, 800, R); y1 = rs.uniform(0, 450, R)
rois = torch.from_numpy(np.stack([np.zeros(R), x1, y1, x1 + rs.uniform(30, 300, R), y1 + rs.uniform(30, 200, R)], 1).astype(np.float32)).cuda()
w1 = (torch.randn(N1, C, generator=g) / 32).cuda().bfloat16(); w2 = (torch.randn(N2, C, generator=g) / 32).cuda().bfloat16()
b1 = torch.randn(N1, generator=g).cuda(); b2 = torch.randn(N2, generator=g).cuda()
pooled = torch.empty(R * 49, C, dtype=torch.bfloat16, device='cuda')
y1_ = torch.empty(R * 49, N1, dtype=torch.bfloat16, device='cuda'); y2_ = torch.empty(R * 49, N2, dtype=torch.bfloat16, device='cuda')
flop = 2.0 * R * 49 * (N1 + N2) * C
for dbg in (1, 2, 3, 4, 7):
    td = timeit(lambda: O.roialign_block0_fwd(fd, H, W, C, rois, R, P, 1.0 / 16.0, w1, b1, N1, w2, b2, N2, pooled, y1_, y2_, debug=dbg))
    print('knock-out %d (1 = no products, 2 = no weight loads, 4 = no stores): %.1f us' % (dbg, td * 1e6), flush=True)
for rnd in range(3):
    tf = timeit(lambda: O.roialign_block0_fwd(fd, H, W, C, rois, R, P, 1.0 / 16.0, w1, b1, N1, w2, b2, N2, pooled, y1_, y2_))
    ta = timeit(lambda: O.roialign_fwd(fd, H, W, C, rois, R, P, 1.0 / 16.0, pooled))
    tb = timeit(lambda: O.conv_igemm(pooled, w1, y1_, R, P, P, C, P, P, N1, bias=b1, relu=True))
    tc = timeit(lambda: O.conv_igemm(pooled, w2, y2_, R, P, P, C, P, P, N2, bias=b2))
    print('fused %.1f us (%.0f TFLOP/s) | crop %.1f + conv1 %.1f + downsample %.1f = %.1f us' % (tf * 1e6, flop / tf / 1e12, ta * 1e6, tb * 1e6, tc * 1e6, (ta + tb + tc) * 1e6), flush=True)
