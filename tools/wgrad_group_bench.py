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
    ap.add_argument('--min-wg', type=int, default=0, help='WgradQueue.MIN_WG (workgroups a grouped launch should have before its problems stop splitting their pixels)')
    args = ap.parse_args()
    if args.lib:
        from lang2seg_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from lang2seg_amd import ops as O
    from lang2seg_amd.nets.network import WgradQueue
    from lang2seg_amd._lib import BF16

    class Net(object):
        dt = BF16; device = 'cuda'; _rec_key = None
        def fork_wgrad(self, alt=False, fixed=None): return contextlib.nullcontext()
        def wgrad_ws(self):
            if not hasattr(self, '_ws'): self._ws = torch.empty(16 << 20, dtype=torch.float32, device='cuda')
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
