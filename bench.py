#!/usr/bin/env python
"""bench.py — train images/sec of the lang2seg cycle train step (ResNet-101 C4 Mask R-CNN + bi-LSTM +
7 spatial dynamic filters + att2in2 caption-cycle loss) on N MI355X of one node.

One "step" = one `train_step` = one 600x1000 image x one 20-token expression: forward, losses,
backward, SGD (the unit behind the reference's `speed: s / iter`, train_val_cycle.py:371,404-411).
Workload = BASELINE.json configs[2] (`train_cycle.sh`, bf16) with the refcocog token/vocab sizes the
headline metric is quoted on.  Inputs are synthetic and resident in HBM before the
timed region; weights are the reference initialisers with a fixed seed.

N > 1: one process per GPU, per-GPU batch 1, RCCL all-reduce of the flat gradient buffer overlapped with
backward (weak scaling).  The ranks come either from the driver (`python -m torch.distributed.run ... bench.py
--gpus N`: RANK / WORLD_SIZE in the environment) or, when `--gpus N` is given without them, from this script
itself: it starts `torch.distributed.run` as a CHILD process before anything here touches a GPU and relays
rank 0's line.

Prints ONE JSON line (rank 0).  `value` times K pipelined steps (loss read back once, after the region, as
`train_net` does between display iterations); the JSON also carries the fully synchronous `train_step` rate
(the reference reads its losses back every step, NET:704-710), the rate with the 7.2 MB image re-uploaded
every step, the roofline of the dominant launch, of the whole 3x3 stack and of the time-dominant group of
convolution launches, and the CPU restatement timed on this box's host cores."""
import argparse
import json
import re
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work: forward 321.29 GMAC, backward 2x except frozen stem+layer1 (9.40 GMAC)
STEP_FLOP = 2.0 * (321.29e9 + 2.0 * (321.29e9 - 9.40e9))
PEAK_BF16 = 2.5e15      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


VARIANT_CHOICES = ('cycle', 'baseline', 'spatial', 'cycle_response', 'vgg')
# algorithmic work per image-step (BASELINE.md section 2 / SURVEY.md section 8d), TFLOP: forward + 2 x (forward - frozen prefix)
VARIANT_TFLOP = {'cycle': 1.89, 'baseline': 1.68, 'spatial': 1.68, 'cycle_response': 2.10, 'vgg': 1.09}
# expression length / vocabulary: refcocog (umd) for the headline and config 4, refcoco(+) (unc) for configs 1, 2 and 5
VARIANT_TV = {'cycle': (20, 3349), 'cycle_response': (20, 3349), 'baseline': (10, 1999), 'spatial': (10, 1999), 'vgg': (10, 1999)}
VARIANT_DESC = {'cycle': 'ResNet-101 C4 + 7 spatial dynamic filters + att2in2 cycle loss',
                'baseline': 'ResNet-101 C4 + 1 dynamic filter', 'spatial': 'ResNet-101 C4 + 7 spatial dynamic filters',
                'cycle_response': 'ResNet-101 C4 + 7 spatial dynamic filters (sigmoid gating, response loss) + att2in2 cycle loss on the map before and after the gating',
                'vgg': 'VGG16 conv5_3 Faster R-CNN (no mask branch) + 7 spatial dynamic filters (sigmoid gating, response loss)'}
VARIANT_SCRIPT = {'cycle': 'train_cycle.sh', 'baseline': 'train_baseline.sh', 'spatial': 'train_spatial.sh', 'cycle_response': 'train_cycle_response.sh',
                  'vgg': 'train_vgg.sh'}


class Hooks(object):
    """What tools/ab.py (A/B experiments, knock-outs, forced one-rank data parallel) can change around the measured run.  bench.py itself
    runs with the defaults below: the product configuration, nothing patched."""
    lib = ''                  # another build of the C-ABI library (tools/build_tools_lib.py); the line is marked
    force_dp = False          # build the data-parallel reducer even for one rank
    dp_skip_allreduce = 0     # EXPERIMENT: 1 = no collective, 2 = no reducer calls, 3 = no reducer; the line is marked invalid
    dp_bucket_update = False
    knockout = ''             # EXPERIMENT: leave parts of the step out (wgrad,cap); the line is marked invalid
    tape, graph, main_prio = True, False, False
    dp_backend = 'nccl'       # 'gloo' (tools/ab.py --dp-backend gloo): the reducer's buffers staged through the host - several ranks may then share one GPU;
                              # a functional run of the N > 1 path, its figures say nothing about RCCL (the line is marked)
    note = ''                 # what was changed, for the JSON line

    def before_net(self):     # class attributes / library tunables, before the network exists
        pass

    def after_net(self, net):
        pass

    def after_optim(self, optim):
        pass


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--variant', default='cycle', choices=VARIANT_CHOICES,
                    help='network variant = BASELINE.json config: cycle (the headline, config 3), baseline (1), spatial (2), cycle_response (4), vgg (5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-steps', default='3,10', help='W,K: warm-up and timed steps of the CPU restatement (BASELINE.md section 3: 3 + 10, ~2-3 min on the GPU box)')
    ap.add_argument('--dp-wire', default='', choices=['', 'fp32', 'bf16'], help='gradient buckets on the wire: fp32 (282 MB / step) or packed to bf16 (141 MB); default: bf16 for N > 1')
    ap.add_argument('--dp-algo', default='', choices=['', 'allreduce', 'rs_ag'], help='one all-reduce per bucket, or reduce-scatter + all-gather; default: rs_ag for N > 1')
    ap.add_argument('--dp-shard-update', type=int, default=-1, help='with rs_ag: every rank updates only its slice of a bucket and the WEIGHTS are all-gathered; default: on for N > 1')
    ap.add_argument('--mixed-shapes', type=int, default=1, help='extra leg: a stream of six different (image size, token count) shapes replayed from pre-recorded tapes')
    ap.add_argument('--height', type=int, default=600)
    ap.add_argument('--width', type=int, default=1000)
    ap.add_argument('--extras', type=int, default=1, help='0: only the headline timing (no synchronous / PCIe-inclusive / per-launch legs)')
    ap.add_argument('--launcher-check', action='store_true', help='only bring the ranks up (gloo without GPUs), count them with an all-reduce, print the line')
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`--gpus N` without a launcher: start the N ranks as children (never re-exec a process that touched a GPU; this one has not)."""
    port = int(os.environ.get('MASTER_PORT', 29400 + os.getpid() % 500))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [l for l in r.stdout.decode(errors='replace').splitlines() if l.strip().startswith('{')]
    if r.returncode != 0 or not lines:
        sys.stderr.write('bench.py: the %d-rank child run failed (exit code %d)\n' % (args.gpus, r.returncode))
        return r.returncode or 1
    sys.stdout.write(lines[-1] + '\n')
    sys.stdout.flush()
    return 0


def cpu_model():
    try:
        for l in open('/proc/cpuinfo'):
            if l.startswith('model name'):
                return l.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(H, W, T, V, warm, timed, variant='cycle'):
    """The oracle's fp32 CPU restatement of the same step on this box's host cores (baseline only, the checker is not shipped)."""
    import copy
    import numpy as np
    import torch
    from oracle import weights as OW, synth as OS, net as ON
    opt = OW.default_opt(vocab_size=V, seq_length=T)
    if variant == 'vgg':
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=3, variant=variant) if variant != 'cycle' else OW.make_state_dict(opt, seed=3)
    blob = OS.make_blob(H, W, T, V, seed=1234)
    net = ON.OracleNet(sd, opt, copy.deepcopy(ON.DEFAULT_CFG), variant=variant)
    rng = np.random.RandomState(3)
    # a BOUNDED sample: up to `warm` warm-up and `timed` timed steps (3 + 10, ~2 min on an idle 128-thread host), cut short
    # when the box's host is busy - warm-up stops after 30 s, the timed leg after 100 s (at least two steps) - so that the run stays within minutes
    t_w, n_w = time.time(), 0
    for _ in range(warm):
        net.train_step(blob, dict(rng=rng)); n_w += 1
        if time.time() - t_w > 30.0:
            break
    ts, t_t = [], time.time()
    for _ in range(timed):
        t0 = time.time()
        net.train_step(blob, dict(rng=rng))
        ts.append(time.time() - t0)
        if len(ts) >= 2 and time.time() - t_t > 100.0:
            break
    warm, timed = n_w, len(ts)
    dt = float(np.mean(ts))
    return {'value': 1.0 / dt, 'unit': 'img/s', 'cores': torch.get_num_threads(), 'kind': 'port', 'cpu': cpu_model(),
            's_per_step': dt, 's_per_step_all': [round(t, 3) for t in ts],
            'sample': '%d warm-up + %d timed train steps of the %s variant (%dx%d image, %d tokens, 256 RoIs, fp32, torch-CPU restatement in oracle/), mean %.1f s / step'
                      % (warm, timed, variant, H, W, T, dt)}


def _baseline_metric():
    """the metric string of BASELINE.json, verbatim"""
    try:
        return json.load(open(os.path.join(ROOT, 'BASELINE.json')))['metric']
    except Exception:
        return 'train images/sec (cycle loss on), 600×1000 input, at 1/2/4/8 MI355X'


PROFILE_ROUND = 'r06'


def src_hash():
    """sha256[:16] over the kernel sources + the build flags: figures taken from a committed profile are only reported while the library they
    were measured on is the one being benchmarked (tools/prof_step.sh writes the same hash next to the profile)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'lang2seg_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode()); h.update(open(os.path.join(d, f), 'rb').read())
    try:
        sys.path.insert(0, ROOT)
        from __graft_entry__ import HIPCC_FLAGS
        h.update(' '.join(HIPCC_FLAGS).encode())
    except Exception:
        pass
    return h.hexdigest()[:16]


def _profile_meta():
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', '%s_step_kernel_stats.meta.json' % PROFILE_ROUND)))
    except Exception:
        return None


def _profile_is_current():
    m = _profile_meta()
    return bool(m) and m.get('src_hash') == src_hash()


def _profile_kernel_time_ms():
    """sum of the committed rocprofv3 per-step kernel table (profiles/<round>_step_kernel_stats.csv), ms per step; None while the table
    was taken on other kernel sources"""
    if not _profile_is_current():
        return None
    try:
        import csv
        rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', '%s_step_kernel_stats.csv' % PROFILE_ROUND))))
        return sum(float(r['total_us_per_step']) for r in rows) * 1e-3
    except Exception:
        return None


def _pmc_traffic(which):
    """PMC counters cannot be read inside the timed run; the committed same-round measurement of the same launch is reported
    (tools/pmc_traffic.sh -> profiles/<round>_pmc_traffic_<which>.json; round 3's where this round has none: the kernels it names did not change)."""
    for name in ('%s_pmc_traffic_%s.json' % (PROFILE_ROUND, which), 'r03_pmc_traffic_%s.json' % which):
        f = os.path.join(ROOT, 'profiles', name)
        try:
            d = json.load(open(f))
            return float(d['traffic_bytes']), float(d['algorithmic_bytes']), d.get('kernel'), name
        except Exception:
            continue
    return None, None, None, None


def _rocprof_avgs(names):
    """average in-step kernel durations (us) of the given kernel templates from the committed rocprofv3 summary of the same command
    (tools/prof_step.sh -> profiles/r03_step_kernel_stats.csv): execution time only, without the wait for a free slot that the
    HIP-event figures include"""
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', '%s_step_kernel_stats.csv' % PROFILE_ROUND)
    if not _profile_is_current():
        return None                                    # the committed profile was taken on another build of the kernels: not this run's numbers
    out = {}
    try:
        import csv
        rows = list(csv.DictReader(open(f)))
    except Exception:
        return None
    for n in names:
        m = re.match(r'(\w+)<([\d, ]+)>', n) or re.match(r'(\w+)', n)
        base, args_ = m.group(1), (m.group(2).replace(' ', '').split(',') if m.lastindex and m.lastindex > 1 else [])
        tot = cnt = 0.0
        for r in rows:
            k = r['kernel'].replace(' ', '')
            if not k.startswith(base + '<') and not k.startswith(base + '('):
                continue
            ka = re.findall(r'-?\d+', k.split('(')[0].split('<', 1)[1]) if '<' in k.split('(')[0] else []
            if base == 'igemm_ring_kernel':
                ok = ka[0:2] == args_[:2] and 'true' not in k.split('(')[0]
            else:
                ok = all(a in ka for a in args_)
            if ok:
                tot += float(r['total_us_per_step']); cnt += float(r['launches_per_step'])
        if cnt:
            out[n] = round(tot / cnt, 2)
    return out or None


def _rocprof_group(counts, gflop):
    """the group's execution time per step as rocprofv3 saw it: launches of each kernel template in the group (counted on the measurement
    tape) x that template's average in-step duration in the committed profile; and the fraction of peak that gives"""
    if not counts:
        return None
    avg = _rocprof_avgs(list(counts)) or {}
    if any(k not in avg for k in counts):
        return None
    ms = sum(n * avg[k] for k, n in counts.items()) * 1e-3
    ach = gflop / ms                                   # GFLOP per ms = TFLOP/s
    return {'launches': counts, 'ms_per_step': ms, 'achieved': ach, 'frac': ach / (PEAK_BF16 / 1e12), 'unit': 'TFLOP/s',
            'profile': 'profiles/%s_step_kernel_stats.csv' % PROFILE_ROUND, 'profile_src_hash': (_profile_meta() or {}).get('src_hash'),
            'note': 'execution time only (rocprofv3 --kernel-trace of the same command on the same kernel sources); the HIP-event figure also contains the wait for free CU slots'}


class LaunchTimer(object):
    """HIP events around every convolution launch of eager steps, recorded on the stream the launch goes to (the weight-gradient
    launches run on their own streams; torch.cuda.Event.record() uses the current stream, which ConvOp switches before launching)."""

    def __init__(self, net, torch):
        self.torch, self.on, self.recs = torch, False, []
        from lang2seg_amd import ops
        self.O = ops
        ops.TRACK_PLAN = True
        self.plans = {}               # group -> kernel names the dispatcher chose (l2s_conv_plan_name)
        self.tape_pairs = []          # (id0, id1) timing events the launch tape records around the dominant launch in every replayed step
        self.tape_all = False         # measurement tape: timing events around EVERY convolution / grouped weight-gradient launch
        self.tape_recs = []           # (group, kind, k, flop, id0, id1) of that tape
        self.tape_ms = None           # [(group, kind, k, flop, ms)]: mean over the replays read back
        self.tape_plan = []           # kernel (l2s_conv_plan_name) of every convolution record of that tape; '' for the grouped weight gradients
        for c in net.convs:
            self._wrap(c)
        net.wgq.on_launch = self.wgrad_hook
        # VGG's conv1_1 (3 -> 64 channels, its own kernel, not a ConvOp): timed like the convolutions, group 'vgg.conv1_1 fwd'
        orig_c3 = ops.conv3x3_c3

        def c3_timed(img, w, bias, y, H, W, _o=orig_c3):
            if not self.on and self.tape_all:
                a = self.O.tape_time_event(); _o(img, w, bias, y, H, W); b_ = self.O.tape_time_event()
                if a >= 0 and b_ >= 0:
                    self.tape_recs.append(('vgg.conv1_1', 'fwd', 3, 2.0 * H * W * 64 * 27, a, b_)); self.tape_plan.append('conv3x3_c3_kernel')
                    self.plans.setdefault('vgg.conv1_1 fwd', set()).add('conv3x3_c3_kernel')
                return
            _o(img, w, bias, y, H, W)
        ops.conv3x3_c3 = c3_timed

    def _group(self, conv, n, IH, IW):
        k = conv.wkey or (conv.group[0] if conv.group else '?')
        if k.startswith('resnet.layer'):
            li = k.split('.')[1]
            if li == 'layer4':
                return 'layer4@RoIs' if n > 1 else 'layer4@map'
            return li
        if k.startswith('rpn') or k == 'rpn_head_w':
            return 'rpn'
        if k.startswith('vgg.features.'):              # VGG16: one group per convolution (conv1_2 ... conv5_3; conv1_1 is its own 3-channel kernel)
            names = {2: 'conv1_2', 5: 'conv2_1', 7: 'conv2_2', 10: 'conv3_1', 12: 'conv3_2', 14: 'conv3_3', 17: 'conv4_1', 19: 'conv4_2', 21: 'conv4_3',
                     24: 'conv5_1', 26: 'conv5_2', 28: 'conv5_3'}
            return 'vgg.' + names.get(int(k.split('.')[2]), k)
        if k.startswith('vgg.classifier.'):
            return {'0': 'vgg.fc6', '3': 'vgg.fc7'}.get(k.split('.')[2], k)
        return 'heads'

    def _wrap(self, conv):
        T = self.torch
        for kind in ('fwd', 'dgrad'):
            orig = getattr(conv, kind)

            def timed(x, n, IH, IW, *rest, _orig=orig, _kind=kind, **kw):
                if not self.on and self.tape_all:
                    OH, OW = conv.out_hw(IH, IW)
                    flop = 2.0 * n * OH * OW * conv.Np * conv.k * conv.k * conv.Cin
                    self.O.LAST_PLAN = None
                    a = self.O.tape_time_event(); r = _orig(x, n, IH, IW, *rest, **kw); b = self.O.tape_time_event()
                    if a >= 0 and b >= 0:
                        self.tape_recs.append((self._group(conv, n, IH, IW), _kind, conv.k, flop, a, b))
                        self.tape_plan.append(self.O.LAST_PLAN or '?')
                        self.plans.setdefault('%s %s' % (self._group(conv, n, IH, IW), _kind), set()).add(self.O.LAST_PLAN or '?')
                    return r
                if not self.on:
                    if _kind == 'fwd' and conv.k == 3 and n > 1 and self._group(conv, n, IH, IW) == 'layer4@RoIs':
                        # the step that records the tape: bracket this launch with timing events that every replay records again
                        a = self.O.tape_time_event(); r = _orig(x, n, IH, IW, *rest, **kw); b = self.O.tape_time_event()
                        if a >= 0 and b >= 0:
                            self.tape_pairs.append((a, b))
                        return r
                    return _orig(x, n, IH, IW, *rest, **kw)
                OH, OW = conv.out_hw(IH, IW)
                flop = 2.0 * n * OH * OW * conv.Np * conv.k * conv.k * conv.Cin
                e0 = T.cuda.Event(enable_timing=True); e1 = T.cuda.Event(enable_timing=True)
                self.O.LAST_PLAN = None
                e0.record(); r = _orig(x, n, IH, IW, *rest, **kw); e1.record()
                self.recs.append((self._group(conv, n, IH, IW), _kind, conv.k, flop, e0, e1))
                self.plans.setdefault('%s %s' % (self._group(conv, n, IH, IW), _kind), set()).add(self.O.LAST_PLAN or '?')
                return r
            setattr(conv, kind, timed)

    def wgrad_hook(self, tag, variant, flop, k):
        """context around one grouped weight-gradient launch (all weight gradients of a backward stage with one tile variant)"""
        names = ('64x64/tap', '128x128/tap', '64x64/row3', '128x64/row3', '256x256/tap', '128x128/row3-dma', '256x256/tap-dma')
        if not self.on and self.tape_all:
            lt = self

            class _T(object):
                def __enter__(s):
                    s.a = lt.O.tape_time_event()

                def __exit__(s, *a_):
                    b = lt.O.tape_time_event()
                    if s.a >= 0 and b >= 0:
                        lt.tape_recs.append(('%s [%s]' % (tag, names[variant]), 'wgrad', 3 if variant in (2, 3, 5) else k, flop, s.a, b))
                        lt.tape_plan.append('')
            return _T()
        if not self.on:
            return None
        T, recs = self.torch, self.recs

        class _C(object):
            def __enter__(s):
                s.e0 = T.cuda.Event(enable_timing=True); s.e0.record()

            def __exit__(s, *a):
                e1 = T.cuda.Event(enable_timing=True); e1.record()
                recs.append(('%s [%s]' % (tag, names[variant]), 'wgrad', 3 if variant in (2, 3, 5) else k, flop, s.e0, e1))
        return _C()

    def summary(self, steps):
        """(per-group table, 3x3 stack, time-dominant group); times are per step, bias column sums ride with the wgrad launches"""
        groups, s3 = {}, [0.0, 0.0, 0]
        src = [(g_, kd, k, fl, ms_) for g_, kd, k, fl, ms_ in self.tape_ms] if self.tape_ms else [(g_, kd, k, fl, e0.elapsed_time(e1)) for g_, kd, k, fl, e0, e1 in self.recs]
        if self.tape_ms:
            steps = 1
        for grp, kind, k, flop, ms in src:
            g = groups.setdefault('%s %s' % (grp, kind), [0.0, 0.0, 0])
            g[0] += flop; g[1] += ms; g[2] += 1
            if k == 3:
                s3[0] += flop; s3[1] += ms; s3[2] += 1
        counts = {}
        if self.tape_ms:
            for (g_, kd, _, _, _, _), pl in zip(self.tape_recs, self.tape_plan):
                if pl:
                    c = counts.setdefault('%s %s' % (g_, kd), {}); c[pl] = c.get(pl, 0) + 1
        self.group_kernel_counts = counts
        tab = {name: {'launches_per_step': v[2] / steps, 'ms_per_step': v[1] / steps, 'gflop_per_step': v[0] / steps / 1e9, 'tflops': v[0] / (v[1] * 1e-3) / 1e12,
                      'frac': v[0] / (v[1] * 1e-3) / PEAK_BF16, 'kernels': sorted(self.plans.get(name, []))} for name, v in groups.items() if v[1] > 0}
        dom = max(tab, key=lambda n: tab[n]['ms_per_step']) if tab else None
        stack = None
        if s3[1] > 0:
            ach = s3[0] / (s3[1] * 1e-3) / 1e12
            stack = {'achieved': ach, 'peak': PEAK_BF16 / 1e12, 'unit': 'TFLOP/s', 'frac': ach / (PEAK_BF16 / 1e12),
                     'launches_per_step': s3[2] / steps, 'ms_per_step': s3[1] / steps, 'gflop_per_step': s3[0] / steps / 1e9,
                     'note': 'every 3x3 convolution of the step (forward and data-gradient launches; the weight gradients as the grouped '
                             'filter-row launches of each backward stage; layer2/3/4 + RPN): summed algorithmic FLOPs / summed HIP-event time '
                             'of the launches (%s)' % ('events on the launch tape, pipelined replayed steps' if self.tape_ms else 'eager multi-stream steps')}
        return tab, stack, dom


def main(argv=None, hooks=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    hooks = hooks or Hooks()
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args, argv)
    rank = int(os.environ.get('RANK', 0)); world = int(env_world or 1); local = int(os.environ.get('LOCAL_RANK', 0))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)' % (args.gpus, world))

    # Native libraries print to fd 1 (RCCL writes a five-line version banner there): the contract is ONE JSON line on stdout, so
    # fd 1 is pointed at stderr for the whole run and the JSON line goes to a saved duplicate of the real stdout.
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + '\n').encode())

    import numpy as np
    import torch
    import torch.distributed as dist
    if hooks.lib:
        from lang2seg_amd import _lib as _L
        _L.LIB_PATH = os.path.abspath(hooks.lib)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')

    if args.launcher_check:
        # bring the ranks up exactly as a measured run does and count them; gloo where there is no GPU (CPU test of the launcher path)
        on_gpu = torch.cuda.device_count() >= world
        if on_gpu:
            torch.cuda.set_device(local)
        dist.init_process_group('nccl' if on_gpu else 'gloo', rank=rank, world_size=world)
        c = torch.ones(1, device='cuda' if on_gpu else 'cpu')
        dist.all_reduce(c)
        if rank == 0:
            emit({'launcher_check': True, 'n_gpus': world, 'ranks_seen': int(c.item()), 'backend': 'nccl' if on_gpu else 'gloo'})
        dist.destroy_process_group()
        return 0

    torch.cuda.set_device(local if hooks.dp_backend == 'nccl' else local % max(torch.cuda.device_count(), 1))
    use_dp = world > 1 or hooks.force_dp
    if use_dp:
        # no device_id: binding the process group to the device eagerly costs 7 % of the step on this stack even when no collective
        # is ever issued (112 vs 121 img/s at one rank); the communicator is created by the first all-reduce
        dist.init_process_group(hooks.dp_backend, rank=rank, world_size=world)
    from lang2seg_amd.model.config import cfg
    from lang2seg_amd.nets.resnet_v1 import resnetv1
    from lang2seg_amd.optim import SGD  # noqa: F401
    from lang2seg_amd.model.train_val import make_optimizer
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader

    if hooks.main_prio:
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    variant = args.variant
    T, V = VARIANT_TV[variant]
    step_flop = STEP_FLOP if variant == 'cycle' else VARIANT_TFLOP[variant] * 1e12
    cfg.COMPUTE_DTYPE = args.dtype
    opt = dict(vocab_size=V, word_embedding_size=512, word_vec_size=512, rnn_hidden_size=512, bidirectional=1, word_drop_out=0.5,
               rnn_drop_out=0.2, rnn_num_layers=1, rnn_type='lstm', variable_lengths=1, C4_feat_dim=1024, cap_loss_weight=1.0,
               caption_model='att2in2', input_encoding_size=512, rnn_size=512, num_layers=1, drop_prob_lm=0.5, seq_length=T,
               fc_feat_size=4096, att_feat_size=4096, att_hid_size=512)
    np.random.seed(cfg.RNG_SEED)
    hooks.before_net()
    if variant == 'vgg':
        from lang2seg_amd.nets.vgg16 import vgg16
        opt['C4_feat_dim'] = 512
        net = vgg16(opt, batch_size=1)
    else:
        net = resnetv1(opt, batch_size=1, num_layers=101, variant=variant)
    net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    net.train()
    net.rank_seed = rank * 1000003
    net.use_graph = bool(hooks.graph) and world == 1
    net.use_tape = bool(hooks.tape)          # N > 1: the tape is cut at the gradient-bucket hand-offs (Network.tape_step)
    net.knockout = frozenset(x for x in hooks.knockout.split(',') if x)
    hooks.after_net(net)
    experiment = bool(net.knockout) or bool(hooks.dp_skip_allreduce)
    dp_desc = 'dp%d' % world
    if use_dp and hooks.dp_skip_allreduce != 3:
        from lang2seg_amd.parallel import GradReducer
        # N > 1 defaults (round 4): bf16 buckets, reduce-scatter + all-gather (direct exchanges over all seven xGMI links), sharded update
        dp_wire = args.dp_wire or ('bf16' if world > 1 else 'fp32')
        dp_algo = args.dp_algo or ('rs_ag' if world > 1 else 'allreduce')
        dp_shard = (args.dp_shard_update if args.dp_shard_update >= 0 else int(world > 1)) and dp_algo == 'rs_ag'
        dp_bucket = (not dp_shard) and hooks.dp_bucket_update and not hooks.dp_skip_allreduce
        net.dp = GradReducer(net, world, skip_allreduce=hooks.dp_skip_allreduce, wire=dp_wire, algo=dp_algo, timing=True, rank=rank,
                             shard_update=True if dp_shard else None, bucket_update=True if dp_bucket else None)
        dp_desc = 'dp%d (%s buckets, %s%s)' % (world, dp_wire, dp_algo, ', sharded update' if dp_shard else (', update per bucket' if dp_bucket else ''))
    # the optimiser of this variant's own solver (train_val*.py construct_graph: param groups, lr x 10 rule, config_vgg for VGG)
    optim = make_optimizer(net, None, world)
    hooks.after_optim(optim)
    loader = SyntheticLoader(num_images=4, sents_per_image=1, H=args.height, W=args.width, T=T, vocab_size=V, rank=rank)
    blobs = [loader.getBatch('train') for _ in range(4)]
    for b in blobs:
        net.upload_blob(b, 0)          # inputs resident in HBM before the timed region

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def rank_max(dt):
        if world > 1:
            tt = torch.tensor([dt], device='cuda' if hooks.dp_backend == 'nccl' else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return dt

    lt = LaunchTimer(net, torch)
    for i in range(args.warmup):
        net.train_step_async(blobs[i % 4], 0, optim)
    barrier()
    # ---- the headline: K pipelined steps, one loss read-back after the region ----
    t0 = time.time()
    for i in range(args.steps):
        loss = net.train_step_async(blobs[i % 4], 0, optim)
    barrier()
    dt = rank_max(time.time() - t0)
    lv = loss.cpu().numpy()
    # ---- the dominant launch inside the pipelined replay: HIP events on its stream, recorded by the tape in every step; the events now hold
    # the LAST step of the timed region; four more pipelined pairs of steps give further samples (outside the timed region) ----
    dom_ms = []
    if lt.tape_pairs:
        dom_ms += [lt.O.time_event_elapsed(a, b) for a, b in lt.tape_pairs]
        for rep in range(4):
            for i in range(2):
                net.train_step_async(blobs[i % 4], 0, optim)
            barrier()
            dom_ms += [lt.O.time_event_elapsed(a, b) for a, b in lt.tape_pairs]
    # ---- per-launch times of EVERY convolution / grouped weight-gradient launch inside pipelined replayed steps: a second tape with a timing
    # event before and after each of those launches (outside the timed region: ~400 event records per step) ----
    if args.extras and getattr(net, 'use_tape', False) and world == 1:
        lt.tape_all = True
        while getattr(net, '_tapes', None):
            net._evict_tape()
        net.train_step_async(blobs[0], 0, optim)                 # records the measurement tape
        barrier()
        acc = None
        NREP = 4
        for rep in range(NREP):
            for i in range(2):
                net.train_step_async(blobs[0], 0, optim)
            barrier()
            v = np.array([lt.O.time_event_elapsed(a, b) for _, _, _, _, a, b in lt.tape_recs])
            acc = v if acc is None else acc + v
        lt.tape_all = False
        if acc is not None and len(acc):
            lt.tape_ms = [(g_, kd, k, fl, float(m)) for (g_, kd, k, fl, _, _), m in zip(lt.tape_recs, acc / NREP)]
        while getattr(net, '_tapes', None):
            net._evict_tape()
    ranks_seen = 1
    extras = {}
    if use_dp:
        c = torch.ones(1, device='cuda' if hooks.dp_backend == 'nccl' else 'cpu'); dist.all_reduce(c); ranks_seen = int(c.item())
        assert ranks_seen == world, (ranks_seen, world)
        if getattr(net, 'dp', None) is not None:
            rep = net.dp.report()                   # HIP events of the LAST timed step: per-bucket exchange time, and what the main stream waited for
            if rep is not None:
                extras['dp'] = rep
    if args.extras:
        # ---- the reference's unit as it stands: train_step() reads the losses back every step (NET:704-710: seven .data[0]) ----
        # (the measurement leg above evicted every launch tape: two untimed steps record the step's tape again - until round 6 that recording,
        # an eagerly issued step with two device syncs, sat inside this leg's timed region and cost it ~4 %: sync came out BELOW dropin)
        for i in range(2):
            net.train_step_async(blobs[i % 4], 0, optim)
        barrier()
        t0 = time.time()
        for i in range(args.steps):
            net.train_step(blobs[i % 4], 0, optim)
        barrier()
        dts = rank_max(time.time() - t0)
        extras['sync_train_step'] = {'ms_per_step': dts / args.steps * 1e3, 'value': world * args.steps / dts, 'unit': 'img/s',
                                     'note': 'Network.train_step: loss read-back (host sync) after every step'}
        # ---- PCIe-inclusive: the 7.2 MB fp32 image goes host -> device again before every step (NET:633-636 does so per sentence) ----
        hosts = [torch.from_numpy(np.ascontiguousarray(b['data'], dtype=np.float32)).pin_memory() for b in blobs]
        devd = [b['_device']['data'] for b in blobs]
        # Double-buffered, as a loader thread does it: the image of step i + 1 crosses PCIe on a copy stream BESIDE step i, and step i + 1 waits
        # for its event.  (Round 6, tools/h2d_overlap.py: the same copy issued on the main stream in front of its step costs 3-7 % - 210-218
        # against 225.5 img/s without any upload - because it sits between the previous step's last launch and the frozen prefix that normally
        # runs beside that step's tail; prefetched it costs 0.5 %: 224.5-224.8.)
        cps = torch.cuda.Stream()
        evs, rd = [torch.cuda.Event() for _ in range(4)], [None] * 4
        barrier()
        with torch.cuda.stream(cps):
            devd[0].copy_(hosts[0], non_blocking=True); evs[0].record(cps)
        t0 = time.time()
        for i in range(args.steps):
            j = i % 4
            torch.cuda.current_stream().wait_event(evs[j])
            net.train_step_async(blobs[j], 0, optim)
            rd[j] = torch.cuda.Event(); rd[j].record()          # behind the step that read buffer j (its static-input copy is its first launch)
            jn = (i + 1) % 4
            if rd[jn] is not None:
                cps.wait_event(rd[jn])                          # the upload may not overtake the last reader of the buffer it overwrites (three steps back)
            with torch.cuda.stream(cps):
                devd[jn].copy_(hosts[jn], non_blocking=True); evs[jn].record(cps)
        barrier()
        dth = rank_max(time.time() - t0)
        extras['pcie_inclusive'] = {'ms_per_step': dth / args.steps * 1e3, 'value': world * args.steps / dth, 'unit': 'img/s',
                                    'h2d_bytes_per_step': int(hosts[0].numel() * 4),
                                    'note': 'the 7.2 MB fp32 image of the NEXT step uploaded from pinned host memory on a copy stream beside every pipelined step (double-buffered)'}
        # ---- the reference's train_step as it stands, both at once: the image goes host -> device (NET:633-636) AND the losses come back as
        # Python floats after every step (NET:704-710) ----
        barrier()
        t0 = time.time()
        for i in range(args.steps):
            devd[i % 4].copy_(hosts[i % 4], non_blocking=True)
            net.train_step(blobs[i % 4], 0, optim)
        barrier()
        dtd = rank_max(time.time() - t0)
        extras['dropin_train_step'] = {'ms_per_step': dtd / args.steps * 1e3, 'value': world * args.steps / dtd, 'unit': 'img/s',
                                       'note': 'Network.train_step with the 7.2 MB image uploaded from pinned host memory before the step and the loss floats read back after it'}
        # ---- mixed shapes: six different (image size, token count) shapes, every tape recorded before the timed region ----
        if args.mixed_shapes and world == 1 and args.height == 600 and args.width == 1000:
            shapes = [(600, 800, 8), (600, 900, 12), (600, 1000, 20), (800, 600, 5), (600, 904, 9), (600, 800, 14)]
            mb = []
            for i, (h, w, t) in enumerate(shapes):
                ld = SyntheticLoader(num_images=1, sents_per_image=1, H=h, W=w, T=t, vocab_size=V, seed=100 + i)
                b = ld.getBatch('train'); net.upload_blob(b, 0); mb.append(b)
            order = np.random.RandomState(0).randint(0, len(mb), 60)
            for i in list(range(len(mb))) + list(order[:12]):
                net.train_step_async(mb[i], 0, optim)
            barrier()
            t0 = time.time()
            for i in order[12:]:
                net.train_step_async(mb[i], 0, optim)
            barrier()
            dtm = time.time() - t0
            extras['mixed_shapes'] = {'value': (len(order) - 12) / dtm, 'unit': 'img/s', 'ms_per_step': dtm / (len(order) - 12) * 1e3, 'shapes': shapes,
                                      'note': 'a random stream of six (height, width, tokens) shapes replayed from their launch tapes; every shape was recorded before the timed region'}
        # ---- per-launch HIP events (eager steps: events cannot bracket launches inside a replayed tape) ----
        net.use_graph = False; net.use_tape = False
        net.train_step_async(blobs[0], 0, optim)
        barrier()
        lt.on = True
        NE = 5
        for i in range(NE):
            net.train_step_async(blobs[i % 4], 0, optim)
        barrier()
        lt.on = False
    # ---- the exact-f32 verification mode beside the bf16 headline: the mode in which losses / seg-logits match the reference
    # to 1e-4 and the proposal lists to the box (tests/test_train_step_gpu.py f32 legs); its own network, same inputs, pipelined like the headline
    if args.extras and world == 1 and args.dtype == 'bf16' and not experiment:
        try:
            cfg.COMPUTE_DTYPE = 'f32'
            if variant == 'vgg':
                net32 = vgg16(opt, batch_size=1)
            else:
                net32 = resnetv1(opt, batch_size=1, num_layers=101, variant=variant)
            net32.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
            net32.train(); net32.use_tape = True; net32.use_graph = False
            op32 = make_optimizer(net32, None, 1)
            n32 = max(3, min(args.steps, 10))
            for i in range(3):
                l32 = net32.train_step_async(blobs[i % 4], 0, op32)
            barrier()
            t0 = time.time()
            for i in range(n32):
                l32 = net32.train_step_async(blobs[i % 4], 0, op32)
            barrier()
            d32 = time.time() - t0
            if np.isfinite(l32.cpu().numpy()[:7]).all():
                extras['f32_train_step'] = {'ms_per_step': d32 / n32 * 1e3, 'value': n32 / d32, 'unit': 'img/s', 'steps': n32,
                                            'note': 'the same pipelined step in the exact-f32 verification mode (f32 activations / weights, v_mfma_f32_16x16x4_f32): the mode '
                                                    'the 1e-4 parity claim is tested in; not the headline'}
            del net32, op32
        except Exception as e:                                    # (a side leg must not cost the run its headline line)
            extras['f32_train_step'] = {'error': '%s: %s' % (type(e).__name__, e)}
        finally:
            cfg.COMPUTE_DTYPE = args.dtype
    # a run whose network went non-finite measured nothing (NaN activations are silently zeroed by the next ReLU)
    if not (np.isfinite(lv[:7]).all() and bool(torch.isfinite(net.P.param).all())):
        raise RuntimeError('non-finite losses or parameters after the run: %s' % lv[:7])
    if rank == 0:
        ms = dt / args.steps * 1e3
        val = world * args.steps / dt
        R = int(cfg.TRAIN.BATCH_SIZE)
        out = {
            'metric': _baseline_metric(), 'value': val, 'unit': 'img/s', 'n_gpus': world, 'ranks_seen': ranks_seen,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%s step: %s, %dx%d image, %d-token expression (V=%d), 12000->2000 proposals, 256 RoIs, per-GPU batch 1'
                                   % (VARIANT_SCRIPT[variant], VARIANT_DESC[variant], args.height, args.width, T, V),
                       'variant': variant, 'parallelism': dp_desc if use_dp else 'dp%d' % world, 'step_tflop': step_flop / 1e12,
                       'weights': 'random (reference initialisers; trunk BN gains scaled so activations stay O(1)), fixed seed',
                       'timed': 'K pipelined train steps (launch tape), losses read back once after the region'},
            'step_tflops_per_gpu': step_flop / (ms * 1e-3) / 1e12, 'step_frac_of_bf16_peak': step_flop / (ms * 1e-3) / PEAK_BF16,
            'final_losses': [float(x) for x in lv[:7]],
        }
        out.update(extras)
        if 'sync_train_step' in extras:
            out['sync_train_step_value'] = extras['sync_train_step']['value']      # Network.train_step as the reference calls it (seven floats read back per step)
        if 'value' in extras.get('f32_train_step', {}):
            out['f32_train_step_value'] = extras['f32_train_step']['value']        # exact-f32 verification mode (parity 1e-4), beside the bf16 headline
        if 'dropin_train_step' in extras:
            out['dropin_train_step_value'] = extras['dropin_train_step']['value']  # ... with the image uploaded before every step as well: the reference's unit as it stands
        if hooks.lib or hooks.note:
            out['ab'] = 'A/B run (tools/ab.py): %s %s' % (hooks.note, hooks.lib)
        if hooks.dp_backend != 'nccl':
            out['experiment'] = 'INVALID as a measurement: collectives staged through the host on a %s group (functional run of the N > 1 path)' % hooks.dp_backend
        if experiment:
            out['experiment'] = 'INVALID as a measurement: knockout=%s dp_skip_allreduce=%d' % (sorted(net.knockout), hooks.dp_skip_allreduce)
        if args.extras:
            tab, stack, dom = lt.summary(NE)
            dk = [r for r in lt.recs if r[0] == 'layer4@RoIs' and r[1] == 'fwd' and r[2] == 3]
            kflop = 2.0 * (R * 49) * 512 * 4608
            kms_eager = float(np.mean([e0.elapsed_time(e1) for _, _, _, _, e0, e1 in dk])) if dk else float('nan')
            kms = float(np.mean(dom_ms)) if dom_ms else kms_eager
            ach = kflop / (kms * 1e-3) / 1e12
            if not (dk or dom_ms):                       # a variant without layer4 on the RoIs (VGG): the best group of the table stands in
                bn = max(tab, key=lambda n_: tab[n_]['frac']) if tab else None
                if bn:
                    kflop = tab[bn]['gflop_per_step'] * 1e9 / tab[bn]['launches_per_step']; kms = tab[bn]['ms_per_step'] / tab[bn]['launches_per_step']
                    ach = tab[bn]['tflops']
            bt, balg, bkern, bfile = _pmc_traffic('best')
            dt_, dalg, dkern, dfile = _pmc_traffic('dominant')
            tnote = ('HBM/fabric-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of one launch of this kernel '
                     '(tools/pmc_traffic.sh -> profiles/%s; read side doubled per the gfx950 FETCH_SIZE correction); algorithmic bytes %s')
            best = {
                'kernel': ('igemm_dma_kernel<256,128> on layer4@RoIs conv3x3 forward (M=%d,N=512,K=4608)' % (R * 49)) if (dk or dom_ms) else
                          ('%s: %s' % (bn, ' / '.join(tab[bn]['kernels'])) if bn else None),
                'achieved': ach, 'peak': PEAK_BF16 / 1e12, 'unit': 'TFLOP/s', 'frac': ach / (PEAK_BF16 / 1e12), 'traffic': bt,
                'traffic_note': tnote % (bfile, balg),
                'avg_launch_ms': kms, 'launches_timed': len(dom_ms) if dom_ms else len(dk), 'gflop_per_launch': kflop / 1e9,
                'timing': ('HIP events recorded by the launch tape on the main stream right before and after the launch, inside the pipelined '
                           'replayed steps (last step of the timed region + 4 later steps)') if dom_ms else 'HIP events around the launch in eager steps',
                'eager_avg_launch_ms': kms_eager,
            }
            # `roofline` describes the TIME-DOMINANT group of launches (the largest summed time per step), `roofline.best` the best kernel
            if dom:
                g = tab[dom]
                per_launch_ms = g['ms_per_step'] / g['launches_per_step']
                rp = _rocprof_group(getattr(lt, 'group_kernel_counts', {}).get(dom), g['gflop_per_step'])
                # `achieved` / `frac`: the group's FLOPs over the HIP-event time of its launches in THIS run (inside the pipelined step: the
                # intervals also hold the wait for free CU slots); `rocprof`: the same launches x the committed rocprofv3 execution time of each
                # kernel template (profiles/, only while its source hash is this build's); `kernel_time_ms_per_step`: the sum of that table.
                out['roofline'] = {
                    'bound': 'mfma', 'kernel': '%s: %d launches per step of %s' % (dom, round(g['launches_per_step']), ' / '.join(g['kernels'])),
                    'achieved': g['tflops'], 'peak': PEAK_BF16 / 1e12, 'unit': 'TFLOP/s', 'frac': g['frac'],
                    'frac_source': 'HIP events of this run, on the launch tape inside pipelined replayed steps (they include the wait for free CU slots); '
                                   'rocprof.frac = the same launches by the committed rocprofv3 execution times',
                    'kernel_time_ms_per_step': _profile_kernel_time_ms(),
                    'launches_per_step': g['launches_per_step'], 'ms_per_step': g['ms_per_step'], 'gflop_per_step': g['gflop_per_step'],
                    'avg_launch_ms': per_launch_ms, 'rocprof_avg_launch_us': _rocprof_avgs(g['kernels']), 'traffic': dt_,
                    'rocprof': rp,
                    'traffic_note': (tnote % (dfile, dalg)) + '; measured on the group\'s 3x3 launch (%s)' % dkern,
                    'timing': ('summed algorithmic FLOPs / summed HIP-event time of the group\'s launches; the events are on the launch tape, right before and after '
                               'each launch on the stream it goes to, read after pipelined replayed steps (a second tape recorded after the timed region: the '
                               'headline tape carries no per-launch events).  An interval includes the wait for free CU slots behind the other streams\' '
                               'workgroups; rocprof_avg_launch_us = execution time alone, from profiles/%s_step_kernel_stats.csv' % PROFILE_ROUND) if lt.tape_ms else
                              ('summed algorithmic FLOPs / summed HIP-event time of the group\'s launches in %d eager multi-stream steps' % NE),
                    'best': best, 'stack3x3': stack, 'groups': tab,
                }
            else:
                out['roofline'] = dict(best, bound='mfma', best=best, stack3x3=stack, groups=tab)
        else:
            out['roofline'] = None
        if world == 1 and not args.no_cpu_baseline:
            w, k = [int(x) for x in args.cpu_baseline_steps.split(',')]
            out['cpu_baseline'] = cpu_baseline(args.height, args.width, T, V, w, k, variant)
        emit(out)
    if use_dp:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
