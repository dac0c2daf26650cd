#!/usr/bin/env python
"""Where the cycles of nms_reduce_kernel go (tools build of the library: tools/build_tools_lib.py): the chain wave's total and its wait for
the block slots; wave 1's wait for its loads and for keep words.  Input: a [n, 4] .npy of sorted boxes (tools/proposal_depth.py <file>)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lang2seg_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'liblang2seg_hip_tools.so')
from lang2seg_amd import ops as O
sb = torch.from_numpy(np.load(sys.argv[1])).cuda()
lib = C.CDLL(_lib.LIB_PATH)
for n in (64, 2048, 4096):
    ws = torch.empty(O.nms_workspace_bytes(n) // 8 + 8, dtype=torch.int64, device='cuda')
    keep = torch.full((2000,), -1, dtype=torch.int32, device='cuda'); num = torch.zeros(1, dtype=torch.int32, device='cuda')
    for _ in range(3):
        O.nms(sb, n, 0.7, 0, 2000, ws, keep, num)
    torch.cuda.synchronize()
    d = (C.c_longlong * 16)()
    lib.l2s_tools_nms_dbg(d)
    print('n %5d kept %4d: chain %d cycles over %d blocks (%.0f per block), of which waiting for slots %d; (ready seen -> keep word published: %d); wave 1: total %d, loads %d, keep words %d, last keep word -> ready %d'
          % (n, int(num.item()), d[0], d[2], d[0] / max(d[2], 1), d[1], d[7], d[5], d[3], d[4], d[6]))
