#!/usr/bin/env python
"""A/B experiments around bench.py's measured run: everything here patches class attributes of the package or tunables of the TOOLS build of
the library (csrc/knobs.h, tools/build_tools_lib.py) and then calls bench.main() with the remaining arguments; the JSON line is marked
('ab': ...), knock-outs are marked INVALID.  bench.py itself has none of these switches, and the product library has no tunables.

    python tools/ab.py --wgrad-wgs 96 -- --steps 100 --warmup 10 --no-cpu-baseline --extras 0 --mixed-shapes 0
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# flag -> tunable of the tools library (l2s_tools_set)
KNOBS = {'dma256': 'dma256_auto', 'pdma_wgs': 'pdma_wgs', 'wgrad_row3_dma': 'wgrad_row3_dma', 'wgrad_wgs': 'wgrad_row3_dma_wgs', 'wgrad_wide': 'wgrad_row3_wide',
         'wgrad_minm': 'wgrad_row3_min_m', 'wgrad_1x1_dma': 'wgrad_1x1_dma', 'wgrad_cap': 'wgrad_grid_cap', 'sgd_blocks': 'sgd_blocks', 'row3_plan': 'row3_plan_mode'}
# flag -> (class, attribute, type)
ATTRS = {'fuse_roialign': ('Network', 'fuse_roialign', bool), 'wgrad_overwrite': ('Network', 'wgrad_overwrite', bool), 'cap_map_prio': ('Network', 'cap_map_prio', int),
         'join_l1': ('Network', 'join_before_layer1', bool), 'stem_mfma': ('Network', 'stem_mfma', bool), 'cap_persist': ('Network', 'cap_persistent', bool),
         'layer1_fused': ('Network', 'layer1_fused', bool),
         'wgrad_min_wg': ('WgradQueue', 'MIN_WG', int), 'wgrad_v5_stream': ('WgradQueue', 'V5_STREAM', str), 'wgrad_small_tile': ('WgradQueue', 'SMALL_M_TILE', int),
         'wgrad_v4_fill': ('WgradQueue', 'V4_FILL', int), 'defer': ('SGD', 'defer', bool), 'layer2_side': ('SGD', 'layer2_side', bool),
         'dp_skip_stages': ('GradReducer', 'SKIP_STAGES', str), 'dp_g16': ('GradReducer', 'shard_g16', bool)}


def main():
    argv = sys.argv[1:]
    rest = []
    if '--' in argv:
        i = argv.index('--'); argv, rest = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    for k in KNOBS:
        ap.add_argument('--' + k.replace('_', '-'), type=int, default=None, help='tools-library tunable %s' % KNOBS[k])
    for k, (c, a, t) in ATTRS.items():
        ap.add_argument('--' + k.replace('_', '-'), type=str if t is str else int, default=None, help='%s.%s' % (c, a))
    ap.add_argument('--conv-algo', type=int, default=0, help='l2s_conv_desc.algo for every convolution (ops.CONV_ALGO)')
    ap.add_argument('--rpn-early', type=int, default=None, help='bit 0 = Network.rpn_bwd_early, bit 1 = rpn_wgrad_early')
    ap.add_argument('--sgd-early', type=int, default=None, help='optim.SGD.early of the run\'s optimiser')
    ap.add_argument('--lib', default='', help='another build of the library (tools/ab_build.sh <rev>)')
    ap.add_argument('--tape', type=int, default=1); ap.add_argument('--graph', type=int, default=0); ap.add_argument('--main-prio', type=int, default=0)
    ap.add_argument('--dp-backend', default='nccl', help="'gloo': the reducer's collectives staged through the host (ranks may share a GPU; functional run only)")
    ap.add_argument('--force-dp', type=int, default=0, help='build the data-parallel reducer even for one rank (fixed costs of each form)')
    ap.add_argument('--dp-bucket-update', type=int, default=0)
    ap.add_argument('--knockout', default='', help='EXPERIMENT: leave parts of the step out (wgrad,cap); the line is marked invalid')
    ap.add_argument('--dp-skip-allreduce', type=int, default=0, help='EXPERIMENT: 1 = no collective, 2 = no reducer calls, 3 = no reducer')
    a = ap.parse_args(argv)
    knobs = {KNOBS[k]: getattr(a, k) for k in KNOBS if getattr(a, k) is not None}

    class H(bench.Hooks):
        lib = a.lib
        force_dp, dp_skip_allreduce, dp_bucket_update, knockout = bool(a.force_dp), a.dp_skip_allreduce, bool(a.dp_bucket_update), a.knockout
        tape, graph, main_prio = bool(a.tape), bool(a.graph), bool(a.main_prio)
        dp_backend = a.dp_backend
        note = ' '.join(argv)

        def before_net(self):
            from lang2seg_amd import _lib, ops
            from lang2seg_amd.nets.network import Network, WgradQueue
            from lang2seg_amd.optim import SGD
            from lang2seg_amd.parallel import GradReducer
            cls = {'Network': Network, 'WgradQueue': WgradQueue, 'SGD': SGD, 'GradReducer': GradReducer}
            for k, (c, at, t) in ATTRS.items():
                v = getattr(a, k)
                if v is not None:
                    setattr(cls[c], at, t(v))
            if a.rpn_early is not None:
                Network.rpn_bwd_early = bool(a.rpn_early & 1); Network.rpn_wgrad_early = bool(a.rpn_early & 2)
            if a.conv_algo:
                ops.CONV_ALGO = a.conv_algo
            if a.pdma_wgs:
                Network.roi_pdma = True
            for name, v in knobs.items():
                _lib.tools_set(name, v)

        def after_optim(self, optim):
            if a.sgd_early is not None:
                optim.early = bool(a.sgd_early)
    if knobs and not a.lib:
        from tools.build_tools_lib import build
        H.lib = build()
    return bench.main(rest, H())


if __name__ == '__main__':
    sys.exit(main())
