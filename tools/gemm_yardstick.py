#!/usr/bin/env python
"""Yardstick only (not the product path): what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) takes on the layer4@RoIs 1x1 shapes, bf16,
no epilogue, beside this repo's fused launches (bias + residual + ReLU / ReLU-mask epilogues) from tools/conv_bench.py.  GPU only."""
import torch, sys


def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    for name, M, K, N in [('l4r 1x1 out', 12544, 512, 2048), ('l4r 1x1 in', 12544, 2048, 512), ('l4r down', 12544, 1024, 2048), ('l4r 1x1 in0', 12544, 1024, 512),
                          ('l4r 3x3 as gemm', 12544, 4608, 512), ('l3 1x1 out', 2394, 256, 1024), ('l3 1x1 in', 2394, 1024, 256)]:
        x = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
        r = torch.randn(M, N, device='cuda').bfloat16()
        us = t(lambda: torch.mm(x, w.t()))
        us2 = t(lambda: torch.relu_(torch.addmm(r, x, w.t())))
        print('%-18s M %5d K %4d N %4d  mm %6.1f us %6.0f TF   addmm+relu %6.1f us' % (name, M, K, N, us, 2.0 * M * K * N / us / 1e6, us2))


if __name__ == '__main__':
    main()
