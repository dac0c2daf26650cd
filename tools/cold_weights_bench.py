#!/usr/bin/env python
"""Do the backbone's small convolutions pay for weights that are not in L2?  A chain of layer3 bottleneck convolutions (38x63 map) replayed
from a launch tape, (a) every launch with the SAME weight tensor (tools/conv_bench.py's regime: L2-warm) and (b) cycling through 48 weight
tensors per shape (56 - 100 MB: the step's regime, where a layer's weights were last touched a step ago), and (c) as (b) with every weight
tensor read once by a small launch on another stream two convolutions ahead (a prefetch into the L2s).  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd import ops as O

H, W, NB = 38, 63, 48
M = H * W
dev = 'cuda'
x1024 = torch.randn(M, 1024, device=dev).bfloat16(); a256 = torch.randn(M, 256, device=dev).bfloat16()
y256 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16); y256b = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
y1024 = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
b256 = torch.randn(256, device=dev); b1024 = torch.randn(1024, device=dev)
w1 = [(torch.randn(256, 1024, device=dev) * 0.05).bfloat16() for _ in range(NB)]
w2 = [(torch.randn(256, 9 * 256, device=dev) * 0.05).bfloat16() for _ in range(NB)]
w3 = [(torch.randn(1024, 256, device=dev) * 0.05).bfloat16() for _ in range(NB)]


def block(i):
    O.conv_igemm(x1024, w1[i], y256, 1, H, W, 1024, H, W, 256, 1, 1, 1, 0, bias=b256, relu=True)
    O.conv_igemm(y256, w2[i], y256b, 1, H, W, 256, H, W, 256, 3, 3, 1, 1, bias=b256, relu=True)
    O.conv_igemm(y256b, w3[i], y1024, 1, H, W, 256, H, W, 1024, 1, 1, 1, 0, bias=b1024, add=x1024, relu=True)


def run(idx, reps=5):
    st = torch.cuda.current_stream()
    for i in idx:
        block(i)
    torch.cuda.synchronize()
    h = O.tape_begin([st])
    for i in idx:
        block(i)
    O.tape_end(h)
    torch.cuda.synchronize()
    O.tape_run(h, [st]); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        O.tape_run(h, [st])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / len(idx) * 1e3


print('per block of three convolutions (1x1 1024->256, 3x3 256->256, 1x1 256->1024 + shortcut), 38x63 map, chain of %d blocks:' % NB)
print('  same weights every block (L2-warm)   %.1f us' % run([0] * NB))
print('  %d weight sets in turn (%.0f MB)       %.1f us' % (NB, NB * (256 * 1024 * 2 + 9 * 256 * 256) * 2 / 1e6, run(list(range(NB)))))
