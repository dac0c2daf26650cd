import sys, os
sys.path.insert(0, os.getcwd())
import torch
from lang2seg_amd import ops as O
H, W, NB = 38, 63, 24
M = H * W
dev = 'cuda'
x1024 = torch.randn(M, 1024, device=dev).bfloat16()
y256 = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(2)]
y256b = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(2)]
y1024 = [torch.empty(M, 1024, device=dev, dtype=torch.bfloat16) for _ in range(2)]
b256 = torch.randn(256, device=dev); b1024 = torch.randn(1024, device=dev)
w1 = (torch.randn(256, 1024, device=dev) * 0.05).bfloat16()
w2 = (torch.randn(256, 9 * 256, device=dev) * 0.05).bfloat16()
w3 = (torch.randn(1024, 256, device=dev) * 0.05).bfloat16()
st = torch.cuda.current_stream(); s2 = torch.cuda.Stream()
def convs(i):
    return [lambda: O.conv_igemm(x1024, w1, y256[i], 1, H, W, 1024, H, W, 256, 1, 1, 1, 0, bias=b256, relu=True),
            lambda: O.conv_igemm(y256[i], w2, y256b[i], 1, H, W, 256, H, W, 256, 3, 3, 1, 1, bias=b256, relu=True),
            lambda: O.conv_igemm(y256b[i], w3, y1024[i], 1, H, W, 256, H, W, 1024, 1, 1, 1, 0, bias=b1024, add=x1024, relu=True)]
def run(two):
    seq = []
    for b in range(NB):
        seq += convs(0)
    def issue():
        for k, f in enumerate(seq):
            if two and (k & 1):
                with torch.cuda.stream(s2):
                    f()
            else:
                f()
    issue(); torch.cuda.synchronize()
    h = O.tape_begin([st, s2]); issue(); O.tape_end(h); torch.cuda.synchronize()
    O.tape_run(h, [st, s2]); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        O.tape_run(h, [st, s2])
    s2.synchronize()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / len(seq) * 1e3
print('chain of %d launches on one stream: %.2f us per launch' % (NB * 3, run(False)))
print('the same launches alternating between two streams, no ordering between them (results garbage): %.2f us per launch' % run(True))
# what a launch boundary costs: a chain of launches that do nothing (one workgroup stores one clock value) on one stream
from lang2seg_amd import _lib
import ctypes as C
buf = torch.zeros(8, dtype=torch.int64, device=dev)
def tiny():
    O.stamp(buf, 0)
tiny(); torch.cuda.synchronize()
h = O.tape_begin([st])
for _ in range(72):
    tiny()
O.tape_end(h); torch.cuda.synchronize()
O.tape_run(h, [st]); torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    O.tape_run(h, [st])
b.record(); torch.cuda.synchronize()
print('chain of 72 one-thread launches on one stream: %.2f us per launch' % (a.elapsed_time(b) / 5 / 72 * 1e3))
