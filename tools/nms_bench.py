#!/usr/bin/env python
"""The proposal chain at the BASELINE size (28 728 anchors -> stable top 12 000 -> NMS 0.7 -> 2000) in the three score regimes of
tests/test_kernels_gpu.py::test_sort_nms_full_size: time of l2s_sort_topk and of l2s_nms (bit mask + greedy scan), and the number of
boxes kept.  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from lang2seg_amd import ops as O
from conv_bench import timeit
from oracle import boxes as OB
import test_kernels_gpu as TK

H, W, A, k = 38, 63, 12, 12000
base = torch.from_numpy(OB.generate_anchors(ratios=(0.5, 1, 2), scales=(4, 8, 16, 32)).astype(np.float32)).cuda()
for dist, seed in (('fresh', 11), ('ties', 12), ('clustered', 13)):
    heads = TK._full_size_heads(dist, H, W, A, seed)
    n = H * W * A
    hd = torch.from_numpy(heads).cuda()
    prob = torch.empty(H * W, 2 * A, device='cuda'); boxes = torch.empty(n, 4, device='cuda'); scores = torch.empty(n, device='cuda')
    O.rpn_decode(hd, heads.shape[1], base, H, W, A, 16, 600.0, 1000.0, prob, boxes, scores)
    sb = torch.empty(k, 4, device='cuda'); ss = torch.empty(k, device='cuda'); si = torch.empty(k, dtype=torch.int32, device='cuda')
    sws = torch.empty(O.sort_ws_ints(n), dtype=torch.int32, device='cuda')
    ts = timeit(lambda: O.sort_topk(scores, boxes, n, k, sws, sb, ss, si))
    ws = torch.empty(O.nms_workspace_bytes(k) // 8 + 8, dtype=torch.int64, device='cuda')
    keep = torch.full((2000,), -1, dtype=torch.int32, device='cuda'); num = torch.zeros(1, dtype=torch.int32, device='cuda')
    out = []
    tn = timeit(lambda: O.nms(sb, k, 0.7, 0, 2000, ws, keep, num))
    torch.cuda.synchronize()
    out.append('nms (bit mask + greedy scan) %.1f us' % (tn * 1e6))
    # how far the scan has to go: boxes kept among all 12000
    keep_all = torch.full((k,), -1, dtype=torch.int32, device='cuda')
    O.nms(sb, k, 0.7, 0, k, ws, keep_all, num); torch.cuda.synchronize()
    kept = int(num.item()); last = int(keep_all[min(kept, 2000) - 1].item())
    print('%-10s sort %.1f us | %s | kept %d of 12000; the 2000th (or last) kept box is row %d' % (dist, ts * 1e6, ' | '.join(out), kept, last), flush=True)
