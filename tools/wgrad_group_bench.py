#!/usr/bin/env python
"""The grouped weight-gradient launches of the step's 'heads' stage (layer4 on the RoIs + on the map: the two launches that carry 0.93 ms of
kernel time inside the step) and of a third of layer3, ALONE on the chip, replayed from a tape.  GPU only.
    python tools/wgrad_group_bench.py [--lib build/ab_<rev>/liblang2seg_hip.so]"""
import sys, os, argparse, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default='')
    ap.add_argument('--row3-dma', type=int, default=-1, help='1 / 0: the LDS-DMA filter-row tile for the large 3x3 problems on / off')
    ap.add_argument('--form', type=int, default=-1, help='pipeline form of the LDS-DMA filter-row kernel (0..3)')
    ap.add_argument('--plan', type=int, default=-1, help='0: contiguous stream-K ranges, 1: XCD-lockstep plan')
    ap.add_argument('--only', default='', help='substring of the stage names to run')
    ap.add_argument('--wgs', type=int, default=0, help='workgroups of its stream-K launch')
    ap.add_argument('--check', type=int, default=0, help='1: compare the layer4 3x3 weight gradients of the two filter-row kernels')
    ap.add_argument('--dma1', type=int, default=-1, help='1 / 0: the LDS-DMA 256x256 tile for the large 1x1 problems on / off')
    ap.add_argument('--minm', type=int, default=0, help='pixels from which a 3x3 problem takes the LDS-DMA filter-row tile')
    ap.add_argument('--min-wg', type=int, default=0, help='WgradQueue.MIN_WG (workgroups a grouped launch should have before its problems stop splitting their pixels)')
    args = ap.parse_args()
    from lang2seg_amd import _lib
    knobs = [(n, v) for n, v, on in (('wgrad_1x1_dma', args.dma1, args.dma1 >= 0), ('wgrad_row3_min_m', args.minm, args.minm > 0),
                                      ('row3_plan_mode', args.plan, args.plan >= 0), ('row3_form', args.form, args.form >= 0),
                                      ('wgrad_row3_dma', args.row3_dma, args.row3_dma >= 0), ('wgrad_row3_dma_wgs', args.wgs, args.wgs > 0)) if on]
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    elif knobs or args.check:                  # tunables exist only in the tools build of the library (csrc/knobs.h)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from build_tools_lib import build
        _lib.LIB_PATH = build()
    from lang2seg_amd import ops as O, _lib as L_
    for n, v in knobs:
        L_.tools_set(n, v)
    from lang2seg_amd.nets.network import WgradQueue
    from lang2seg_amd._lib import BF16

    class Net(object):
        dt = BF16; device = 'cuda'; _rec_key = None
        def fork_wgrad(self, alt=False, fixed=None): return contextlib.nullcontext()
        def wgrad_ws(self):
            if not hasattr(self, '_ws'): self._ws = torch.empty(32 << 20, dtype=torch.float32, device='cuda')
            return self._ws
    net = Net()
    if args.min_wg:
        WgradQueue.MIN_WG = args.min_wg
    bf = lambda *s: (torch.randn(*s, device='cuda') * 0.1).bfloat16()
    launches = []

    class Hook(object):
        def __init__(self, tag, v, flop, k): self.t = (tag, v, flop, k)
        def __enter__(self): launches.append(self.t)
        def __exit__(self, *a): pass

    def stage(name, probs):
        if args.only and args.only not in name:
            return
        q = WgradQueue(net)
        q.on_launch = lambda tag, v, flop, k: Hook(tag, v, flop, k)
        keep = []
        def fill():
            for (Cin, Cout, k, segs) in probs:
                dw = torch.zeros(Cout, k * k * Cin, device='cuda'); keep.append(dw)
                for (n, H, W) in segs:
                    g, x = bf(n * H * W, Cout), bf(n * H * W, Cin); keep.extend([g, x])
                    q.add(dw, g, x, n, H, W, Cin, H, W, Cout, k, 1, k // 2)
        st = torch.cuda.current_stream()
        fill(); q.flush(name); torch.cuda.synchronize()
        launches.clear()
        fill()
        h = O.tape_begin([st]); q.flush(name); O.tape_end(h)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            O.tape_run(h, [st]); torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); O.tape_run(h, [st]); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        t = sorted(ts)[len(ts) // 2]
        flop = sum(l[2] for l in launches)
        print('%-28s %7.1f us  %6.1f GFLOP  %5.0f TFLOP/s   launches: %s' % (name, t, flop / 1e9, flop / t / 1e6, [(l[1], round(l[2] / 1e9)) for l in launches]))

    R, MAP = (256, 7, 7), (1, 38, 63)
    l4_3x3 = [(512, 512, 3, [R, MAP])] * 3
    if args.check:
        torch.manual_seed(0)
        res = []
        data = [(bf(n * H * W, 512), bf(n * H * W, 512), n, H, W) for (n, H, W) in (R, MAP)]
        for on in (0, 1):
            L_.tools_set('wgrad_row3_dma', on)
            q = WgradQueue(net)
            dws = [torch.ones(512, 9 * 512, device='cuda') for _ in range(3)]
            for dw in dws:
                for g, x, n, H, W in data:
                    q.add(dw, g, x, n, H, W, 512, H, W, 512, 3, 1, 1)
            q.flush('check%d' % on); torch.cuda.synchronize()
            res.append(dws)
        for i in range(3):
            a, b = res[0][i], res[1][i]
            print('problem %d: max |old - new| %.3e, max |old| %.3e, equal to each other across problems: %s' % (
                i, float((a - b).abs().max()), float(a.abs().max()), torch.equal(res[1][i], res[1][0])))
        L_.tools_set('wgrad_row3_dma', args.row3_dma if args.row3_dma >= 0 else 1)
    l4_1x1 = [(1024, 512, 1, [R, MAP]), (2048, 512, 1, [R, MAP]), (2048, 512, 1, [R, MAP]), (512, 2048, 1, [R, MAP]), (512, 2048, 1, [R, MAP]),
              (512, 2048, 1, [R, MAP]), (1024, 2048, 1, [R, MAP])]
    stage('layer4 3x3 (filter rows)', l4_3x3)
    stage('layer4 1x1', l4_1x1)
    stage('layer4 all (the heads stage)', l4_3x3 + l4_1x1)
    l3 = []
    for _ in range(8):
        l3 += [(1024, 256, 1, [MAP]), (256, 256, 3, [MAP]), (256, 1024, 1, [MAP])]
    stage('layer3, 8 blocks', l3)


if __name__ == '__main__':
    main()
