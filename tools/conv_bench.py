#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM kernels on the shapes of SURVEY.md Appendix A
(forward, data-gradient, weight-gradient), TFLOP/s per shape.  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lang2seg_amd import ops as O

SHAPES = [
    # name, n_img, H, W, Cin, Cout, k, stride, pad
    ('l2 3x3', 1, 75, 125, 128, 128, 3, 1, 1),
    ('l2 1x1 out', 1, 75, 125, 128, 512, 1, 1, 0),
    ('l2 1x1 in', 1, 75, 125, 512, 128, 1, 1, 0),
    ('l3 3x3', 1, 38, 63, 256, 256, 3, 1, 1),
    ('l3 1x1 out', 1, 38, 63, 256, 1024, 1, 1, 0),
    ('l3 1x1 in', 1, 38, 63, 1024, 256, 1, 1, 0),
    ('rpn 3x3', 1, 38, 63, 1024, 512, 3, 1, 1),
    ('l4r 3x3', 256, 7, 7, 512, 512, 3, 1, 1),
    ('l4r 1x1 out', 256, 7, 7, 512, 2048, 1, 1, 0),
    ('l4r 1x1 in', 256, 7, 7, 2048, 512, 1, 1, 0),
    ('l4r down', 256, 7, 7, 1024, 2048, 1, 1, 0),
    ('l4m 3x3', 1, 38, 63, 512, 512, 3, 1, 1),
    ('l4m 1x1 out', 1, 38, 63, 512, 2048, 1, 1, 0),
]


def timeit(fn, iters=20):
    """GPU time per launch; the launches are replayed from a launch tape (one C call), so the ctypes/Python cost per
    call (~5-10 us, more than the small kernels take) is not in the number."""
    st = torch.cuda.current_stream()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    h = O.tape_begin([st])
    for _ in range(iters):
        fn()
    O.tape_end(h)
    torch.cuda.synchronize()
    O.tape_run(h, [st])
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    O.tape_run(h, [st])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    dt = 1
    tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    xm = int(sys.argv[2]) if len(sys.argv) > 2 else -1          # xcd_mode: -1 = auto, 0 = M-chunks, 1 = N-chunks
    print('%-14s %8s %8s %8s | %8s %8s %8s (us | TFLOP/s)' % ('shape', 'fwd', 'dgrad', 'wgrad', 'fwd', 'dgrad', 'wgrad'))
    for name, n, H, W, Cin, Cout, k, s, p in SHAPES:
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        M = n * OH * OW
        x = torch.randn(n * H * W, Cin, device='cuda').bfloat16()
        w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.05).bfloat16()
        wt = (torch.randn(Cin, k * k * Cout, device='cuda') * 0.05).bfloat16()
        y = torch.empty(M, Cout, device='cuda', dtype=torch.bfloat16)
        dy = torch.randn(M, Cout, device='cuda').bfloat16()
        dx = torch.empty(n * H * W, Cin, device='cuda', dtype=torch.bfloat16)
        dw = torch.zeros(Cout, k * k * Cin, device='cuda')
        bias = torch.randn(Cout, device='cuda')
        flop = 2.0 * M * Cout * k * k * Cin
        algo = int(sys.argv[3]) if len(sys.argv) > 3 else None
        tf = timeit(lambda: O.conv_igemm(x, w, y, n, H, W, Cin, OH, OW, Cout, k, k, s, p, bias=bias, add=y, relu=True, tile=tile, xcd_mode=xm, algo=algo))
        td = timeit(lambda: O.conv_igemm(dy, wt, dx, n, OH, OW, Cout, H, W, Cin, k, k, 1, k - 1 - p, ref=x, tile=tile, xcd_mode=xm, algo=algo))
        ws = torch.empty(64 << 18, device='cuda')
        tw = timeit(lambda: O.conv_wgrad(dy, x, dw, n, H, W, Cin, OH, OW, Cout, k, k, s, p, ws=ws))
        print('%-14s %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f' % (name, tf * 1e6, td * 1e6, tw * 1e6, flop / tf / 1e12, flop / td / 1e12, flop / tw / 1e12))


if __name__ == '__main__':
    main()
