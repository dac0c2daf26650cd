import sys, os
sys.path.insert(0, os.getcwd())
import torch
from lang2seg_amd import ops as O
H, W, NB = 38, 63, 24
M = H * W
dev = 'cuda'
x1024 = torch.randn(M, 1024, device=dev).bfloat16(); y256b = torch.randn(M, 256, device=dev).bfloat16()
y1024 = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16); y256 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
b1024 = torch.randn(1024, device=dev); b256 = torch.randn(256, device=dev)
w3 = (torch.randn(1024, 256, device=dev) * 0.05).bfloat16()
w1 = (torch.randn(256, 1024, device=dev) * 0.05).bfloat16()
w2 = (torch.randn(256, 9 * 256, device=dev) * 0.05).bfloat16()
st = torch.cuda.current_stream()
def timeit(f, n=48):
    f(); torch.cuda.synchronize()
    h = O.tape_begin([st])
    for _ in range(n): f()
    O.tape_end(h); torch.cuda.synchronize()
    O.tape_run(h, [st]); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): O.tape_run(h, [st])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / n * 1e3
for name, kw in (('auto', {}), ('tile 64', dict(tile=64)), ('tile 128', dict(tile=128)), ('algo staged', dict(algo=1)), ('algo ksplit', dict(algo=5)), ('algo dma', dict(algo=2))):
    try:
        t3 = timeit(lambda: O.conv_igemm(y256b, w3, y1024, 1, H, W, 256, H, W, 1024, 1, 1, 1, 0, bias=b1024, add=x1024, relu=True, **kw))
        t1 = timeit(lambda: O.conv_igemm(x1024, w1, y256, 1, H, W, 1024, H, W, 256, 1, 1, 1, 0, bias=b256, relu=True, **kw))
        t2 = timeit(lambda: O.conv_igemm(y256, w2, y256b, 1, H, W, 256, H, W, 256, 3, 3, 1, 1, bias=b256, relu=True, **kw))
        print('%-12s conv1 (1024->256) %.2f us   conv2 (3x3 256) %.2f us   conv3 (256->1024 + shortcut) %.2f us' % (name, t1, t2, t3))
    except Exception as e:
        print(name, 'failed', str(e)[:80])
