"""Times l2s_roialign_bwd at the train-step size (256 RoIs on a 38x63x1024 map, 64 of them jittered copies of one box).
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lang2seg_amd import ops as O
H, W, C, R = 38, 63, 1024, 256
rs = np.random.RandomState(0)
rois = np.zeros((R, 5), np.float32)
rois[:, 1] = rs.uniform(0, 800, R); rois[:, 2] = rs.uniform(0, 450, R)
rois[:, 3] = np.minimum(rois[:, 1] + rs.uniform(30, 500, R), 999); rois[:, 4] = np.minimum(rois[:, 2] + rs.uniform(30, 400, R), 599)
gt = np.array([300, 150, 620, 480], np.float32)
rois[:64, 1:] = gt + rs.uniform(-30, 30, (64, 4))
rd = torch.from_numpy(rois).cuda()
dout = torch.randn(R * 49, C, device='cuda').bfloat16()
dfeat = torch.zeros(H * W, C, device='cuda')
for _ in range(5):
    O.roialign_bwd(dout, H, W, C, rd, R, 7, 1.0 / 16.0, dfeat)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    O.roialign_bwd(dout, H, W, C, rd, R, 7, 1.0 / 16.0, dfeat)
b.record(); torch.cuda.synchronize()
print('roialign_bwd %s: %.1f us' % ('gather', a.elapsed_time(b) * 1000 / 50))
