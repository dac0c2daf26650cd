"""SolverWrapper / train_net — the solver layer of the reference
(pyutils/mask-faster-rcnn/lib/model/train_val_cycle.py, "TV"): same loop structure (TV:327-434: per image
`loader.getBatch`, random sentence order, one `net.train_step` per sentence, LR step at STEPSIZE+1,
display every cfg.TRAIN.DISPLAY iters as `speed: s / iter`, snapshot every SNAPSHOT_ITERS), same snapshot
files (`<prefix>_iter_N.pth` state dict in the reference's key/shape format + `.pkl` sidecar with numpy /
python RNG state, loader cursors and iter, TV:57-104) and the same resume rules (TV:106-165,227-310:
newest snapshot, name+shape matched copy incl. the `[:, :-1]` partial rule, LR rescaled by passed steps).
Data parallel (not in the reference): every rank runs the same loop on its shard; rank 0 writes the weights and the
reference-format sidecar, every other rank writes its own cursor / RNG sidecar `<prefix>_iter_N.rank<r>.pkl` (the
shards are `split_ix[rank::world]`: their lengths, permutations and per-rank RNG streams differ)."""
import glob
import os
import pickle
import random
import time

import numpy as np
import torch

from .config import cfg
from ..optim import SGD


class Timer(object):
    """utils/timer.py:20-35: device-synchronised tic/toc, running average."""

    def __init__(self):
        self.total, self.calls, self.start, self.diff, self.avg = 0.0, 0, 0.0, 0.0, 0.0

    def tic(self):
        torch.cuda.synchronize()
        self.start = time.time()

    def toc(self, average=True, calls=1):
        """`calls` > 1: the bracketed interval covered that many steps (train_model times a whole display window at once)"""
        torch.cuda.synchronize()
        self.diff = time.time() - self.start
        self.total += self.diff
        self.calls += calls
        self.avg = self.total / self.calls
        return self.avg if average else self.diff

    def average_time(self):
        return self.avg


def scale_lr(optimizer, scale):
    for g in optimizer.param_groups:
        g['lr'] *= scale


def make_optimizer(net, solver_cfg=None, world=1, **kw):
    """The optimiser the reference's solver of this network's variant builds in construct_graph() (train_val.py:186-207 and its five
    siblings, see nets/variants.SOLVERS): torch.optim.SGD with momentum and one param group per tensor - weight decay on non-bias tensors
    (BIAS_DECAY False), lr x (DOUBLE_BIAS + 1) on biases, lr x 10 on rnn_encoder / dynamic_fc / response keys outside the two cycle solvers (or, with
    TRAIN.FROM_FRCN, the detector fine-tuning rule of train_val.py:175-185: lr x GAMMA for everything but the mask branch);
    hyper-parameters from the config module THAT solver imports (config_vgg.py for VGG: WEIGHT_DECAY 5e-4, DOUBLE_BIAS True).  Here the
    groups are the rows of ParamStore's segment table and the update is one fused launch (optim.SGD)."""
    from ..nets.variants import solver_cfg as _scfg
    c = _scfg(net.variant) if solver_cfg is None else solver_cfg
    net.P.build_segments(double_bias=c.TRAIN.DOUBLE_BIAS, bias_decay=c.TRAIN.BIAS_DECAY, from_frcn=bool(c.TRAIN.FROM_FRCN), gamma=c.TRAIN.GAMMA)
    return SGD(net, c.TRAIN.LEARNING_RATE, c.TRAIN.MOMENTUM, c.TRAIN.WEIGHT_DECAY, grad_scale=1.0 / world, **kw)


class SolverWrapper(object):
    def __init__(self, network, loader, output_dir, tbdir, pretrained_model=None, rank=0, world=1, solver_cfg=None):
        from ..nets.variants import solver_cfg as _scfg
        # the config object this variant's solver reads (model/config_vgg.py for VGG, train_val_vgg.py:12); shadows the module's `cfg` below
        self.cfg = _scfg(getattr(network, 'variant', 'cycle')) if solver_cfg is None else solver_cfg
        self.net, self.loader = network, loader
        self.output_dir, self.tbdir, self.pretrained_model = output_dir, tbdir, pretrained_model
        self.rank, self.world = rank, world
        os.makedirs(output_dir, exist_ok=True)
        # experiment switches that change what is trained (bench.py --knockout / --dp-skip-allreduce) have no place in a training run
        if getattr(network, 'knockout', None):
            raise RuntimeError('train_net: network.knockout = %s is a benchmarking experiment, not a training mode' % sorted(network.knockout))
        if getattr(getattr(network, 'dp', None), 'skip_allreduce', 0):
            raise RuntimeError('train_net: the gradient reducer was built with skip_allreduce (ranks would diverge)')

    def _sidecar(self, it, rank=None):
        rank = self.rank if rank is None else rank
        base = os.path.join(self.output_dir, self.cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_{:d}'.format(it))
        return base + ('.pkl' if rank == 0 else '.rank{:d}.pkl'.format(rank))

    def _write_sidecar(self, nfilename, it):
        with open(nfilename, 'wb') as fid:
            pickle.dump(np.random.get_state(), fid, pickle.HIGHEST_PROTOCOL)
            pickle.dump(random.getstate(), fid, pickle.HIGHEST_PROTOCOL)
            for split in ('train', 'val'):
                pickle.dump(self.loader.iterators[split], fid, pickle.HIGHEST_PROTOCOL)
                pickle.dump(self.loader.perm[split], fid, pickle.HIGHEST_PROTOCOL)
            pickle.dump(it, fid, pickle.HIGHEST_PROTOCOL)
            # after the reference's fields (a reader of its format stops above): the device-side RNG step counter, so that a resumed run
            # draws the sampling keys / dropout masks the uninterrupted one would have
            pickle.dump(int(self.net.seed_counter().item()), fid, pickle.HIGHEST_PROTOCOL)

    # ---- TV:57-104 -----------------------------------------------------
    def snapshot(self, it):
        nfilename = self._sidecar(it)
        dp = getattr(self.net, 'dp', None)
        if dp is not None and hasattr(dp, 'gather_master'):
            self.net.join_update()
            dp.gather_master()                           # (a collective: every rank is here) the fp32 masters of the sharded update, together again
        if self.rank != 0:
            self._write_sidecar(nfilename, it)           # this rank's own loader cursor / permutation / RNG streams
            return None, nfilename
        filename = os.path.join(self.output_dir, self.cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_{:d}'.format(it) + '.pth')
        torch.save(self.net.state_dict(), filename)
        print('Wrote snapshot to: {:s}'.format(filename))
        self._write_sidecar(nfilename, it)
        return filename, nfilename

    # ---- TV:106-165 ----------------------------------------------------
    def load_matched(self, saved_state_dict):
        """name+shape matched copy; `param[:, :-1]` partial rule for widened inputs (TV:117-129)."""
        cur = self.net.state_dict()
        n_part = n_miss = 0
        for name, param in cur.items():
            if name in saved_state_dict and tuple(param.shape) == tuple(saved_state_dict[name].shape):
                cur[name] = saved_state_dict[name]
            elif name in saved_state_dict and param.dim() == 4 and tuple(param[:, :-1].shape) == tuple(saved_state_dict[name].shape):
                p = param.clone(); p[:, :-1] = saved_state_dict[name]; cur[name] = p; n_part += 1
            else:
                n_miss += 1
        print('size partially match:', n_part); print('size not match:', n_miss)
        self.net.load_state_dict(cur)

    def _any_rank(self, flag):
        """logical OR of `flag` over the ranks of the run (one small all-reduce; the local value in a single-process run)"""
        import torch.distributed as dist
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return bool(int(t.item()))
        return bool(flag)

    def from_snapshot(self, sfile, nfile):
        print('Restoring model snapshots from {:s}'.format(sfile))
        self.load_matched(torch.load(str(sfile), map_location='cpu'))
        # data parallel: every rank but 0 has its own sidecar (its shard has its own length, permutation and RNG streams); rank 0's file is the
        # reference-format one.  A snapshot written by a single-process or a smaller run lacks some of them.  Whether to go on is decided
        # TOGETHER, before anything else is exchanged: a rank that raised alone would leave the others waiting in their first collective
        # until the RCCL timeout.
        own = str(nfile)[:-len('.pkl')] + '.rank{:d}.pkl'.format(self.rank)
        missing = self.rank != 0 and not os.path.exists(own)
        if self._any_rank(missing) and not self.cfg.TRAIN.ALLOW_RESHARD_RESUME:
            raise ValueError('%s: a rank of this %d-rank run has no sidecar in the snapshot (%s; written by a run with fewer ranks?).  Resuming '
                             'would replay different data and random streams on that rank than the run that wrote the snapshot; set '
                             'TRAIN.ALLOW_RESHARD_RESUME True to continue with freshly seeded cursors there.'
                             % (nfile, self.world, ('missing here: ' + own) if missing else 'present on this rank'))
        if self.rank != 0:
            if missing:
                with open(nfile, 'rb') as fid:
                    for _ in range(6):
                        pickle.load(fid)
                    it_ = pickle.load(fid)
                    try:                                   # rank 0's device RNG step counter: every rank draws from the same counter stream
                        self.net.seed_counter().fill_(int(pickle.load(fid)))
                    except EOFError:
                        pass
                print('WARNING: rank {:d} resumes at iteration {:d} WITHOUT its own sidecar: loader cursor, permutation and host RNG '
                      'streams are freshly seeded (TRAIN.ALLOW_RESHARD_RESUME)'.format(self.rank, it_))
                return it_
            nfile = own
        with open(nfile, 'rb') as fid:
            np_state, py_state = pickle.load(fid), pickle.load(fid)
            cursors = {}
            for split in ('train', 'val'):
                it_, perm_ = pickle.load(fid), pickle.load(fid)
                n_here = len(self.loader.split_ix[split]) if split in getattr(self.loader, 'split_ix', {}) else len(perm_)
                if len(perm_) != n_here:
                    raise ValueError('%s: the %s permutation has %d entries, this rank\'s shard has %d images (snapshot written '
                                     'with a different world size?)' % (nfile, split, len(perm_), n_here))
                cursors[split] = (it_, perm_)
            np.random.set_state(np_state)
            random.setstate(py_state)
            for split, (it_, perm_) in cursors.items():
                self.loader.iterators[split] = it_
                self.loader.perm[split] = perm_
            last_snapshot_iter = pickle.load(fid)
            try:
                self.net.seed_counter().fill_(int(pickle.load(fid)))
            except EOFError:                      # a sidecar written by the reference
                pass
        return last_snapshot_iter

    # ---- TV:167-225 ----------------------------------------------------
    def construct_graph(self):
        torch.manual_seed(self.cfg.RNG_SEED)
        random.seed(self.cfg.RNG_SEED)
        np.random.seed(self.cfg.RNG_SEED + self.rank)
        if not hasattr(self.net, 'P'):
            self.net.create_architecture(81, tag='default', anchor_scales=self.cfg.ANCHOR_SCALES, anchor_ratios=self.cfg.ANCHOR_RATIOS)
        lr = self.cfg.TRAIN.LEARNING_RATE
        self.optimizer = make_optimizer(self.net, self.cfg, self.world)
        return lr, self.optimizer

    def find_previous(self):
        sfiles = glob.glob(os.path.join(self.output_dir, self.cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_*.pth'))
        sfiles.sort(key=os.path.getmtime)
        red = [os.path.join(self.output_dir, self.cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_{:d}.pth'.format(s + 1)) for s in self.cfg.TRAIN.STEPSIZE]
        sfiles = [s for s in sfiles if s not in red]
        nfiles = [f for f in glob.glob(os.path.join(self.output_dir, self.cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_*.pkl')) if '.rank' not in os.path.basename(f)]
        nfiles.sort(key=os.path.getmtime)
        red = [r.replace('.pth', '.pkl') for r in red]
        nfiles = [n for n in nfiles if n not in red]
        assert len(nfiles) == len(sfiles)
        return len(sfiles), nfiles, sfiles

    def initialize(self):
        # TV:249-281.  pretrained_model=None is the caller's explicit choice to start from the weights the network holds
        # (tools: --from_scratch; tests); a path that does not exist raises, as torch.load does in the reference (TV:262) — a
        # mistyped path must not silently train from the random initialisers.
        if self.pretrained_model is None:
            print('No pretrained model given: training starts from the weights the network holds')
        elif not os.path.exists(self.pretrained_model):
            raise FileNotFoundError('pretrained model not found: %s (pass pretrained_model=None / --from_scratch to train from '
                                    'the initialisers)' % self.pretrained_model)
        else:
            print('Loading initial model weights from {:s}'.format(self.pretrained_model))
            self.load_matched(torch.load(self.pretrained_model, map_location='cpu'))
            print('Loaded.')
        return self.cfg.TRAIN.LEARNING_RATE, 0, list(self.cfg.TRAIN.STEPSIZE), [], []

    def restore(self, sfile, nfile):
        last = self.from_snapshot(sfile, nfile)
        lr_scale, stepsizes = 1, []
        for s in self.cfg.TRAIN.STEPSIZE:
            if last > s:
                lr_scale *= self.cfg.TRAIN.GAMMA
            else:
                stepsizes.append(s)
        scale_lr(self.optimizer, lr_scale)
        if self.rank != 0:                       # the files this rank owns (rank 0 owns the weights and the reference-format sidecar)
            own = str(nfile)[:-len('.pkl')] + '.rank{:d}.pkl'.format(self.rank)
            return self.cfg.TRAIN.LEARNING_RATE * lr_scale, last, stepsizes, [own if os.path.exists(own) else None], [None]
        return self.cfg.TRAIN.LEARNING_RATE * lr_scale, last, stepsizes, [nfile], [sfile]

    def remove_snapshot(self, np_paths, ss_paths):
        for paths in (np_paths, ss_paths):
            while len(paths) > self.cfg.TRAIN.SNAPSHOT_KEPT:
                f = paths.pop(0)
                if f is not None and os.path.exists(str(f)):        # every rank removes the files it wrote
                    os.remove(str(f))

    # ---- TV:327-434 ----------------------------------------------------
    def train_model(self, max_iters):
        lr, self.optimizer = self.construct_graph()
        lsf, nfiles, sfiles = self.find_previous()
        if lsf == 0:
            lr, last_snapshot_iter, stepsizes, np_paths, ss_paths = self.initialize()
        else:
            lr, last_snapshot_iter, stepsizes, np_paths, ss_paths = self.restore(str(sfiles[-1]), str(nfiles[-1]))
        it = last_snapshot_iter + 1
        stepsizes.append(max_iters); stepsizes.reverse()
        next_stepsize = stepsizes.pop()
        self.net.train(); self.net.cuda()
        # TRAIN.USE_TAPE: record every (image size, token counts) on a launch tape while it first executes and replay it afterwards
        # (Network.tape_step; least recently used tapes and their activation plans are evicted).  Measured: a stream of six mixed
        # shapes 120-121 img/s against 116 img/s issued eagerly (tools/mixed_shape_bench.py); one fixed shape 122 vs 102-110 (600x1000),
        # 127.5 vs 112-116 (600x800), 135 vs 101 (480x640) (bench.py --tape 1/0).
        self.net.use_tape = bool(getattr(self.cfg.TRAIN, 'USE_TAPE', True))
        timer = Timer()
        pending = 0                     # steps issued since the timer was started
        while it < max_iters + 1:
            blobs = self.loader.getBatch('train', self.net._batch_size)
            sent_num = blobs['gt_boxes'].shape[0]
            arr = np.random.permutation(sent_num)
            for idx in range(sent_num):
                if pending == 0:
                    timer.tic()
                if it == next_stepsize + 1:
                    self.snapshot(it)
                    lr *= self.cfg.TRAIN.GAMMA
                    scale_lr(self.optimizer, self.cfg.TRAIN.GAMMA)
                    next_stepsize = stepsizes.pop()
                # The reference reads its 7 losses back after every step (NET:704-710) and brackets every step with a device-synchronising
                # timer (TV:371,404; utils/timer.py:20-35), but only shows both every DISPLAY iterations (TV:404-411).  Here the steps of a
                # display window are issued back to back without a host sync; the losses are fetched and the window is timed when they are shown
                # (`speed` stays the running average of seconds per iteration).
                show = it % self.cfg.TRAIN.DISPLAY == 0
                snap = it % self.cfg.TRAIN.SNAPSHOT_ITERS == 0
                if show or not hasattr(self.net, 'train_step_async'):
                    vals = self.net.train_step(blobs, int(arr[idx]), self.optimizer)  # 6, 7 or 8 floats depending on the network variant
                else:
                    self.net.train_step_async(blobs, int(arr[idx]), self.optimizer)
                pending += 1
                if show or snap or it >= max_iters:
                    timer.toc(calls=pending)
                    pending = 0
                if show and self.rank == 0:
                    short = dict(rpn_cross_entropy='rpn_loss_cls', cross_entropy='loss_cls')
                    names = self.net._loss_names()
                    print('iter: %d / %d, total loss: %.6f' % (it, max_iters, vals[-1]))
                    for n, v in zip(names[:-1], vals[:-1]):
                        print(' >>> %s: %.6f' % (short.get(n, n), v))
                    print(' >>> lr: %f' % lr)
                    print('speed: {:.3f}s / iter'.format(timer.average_time()))
                if it % self.cfg.TRAIN.SNAPSHOT_ITERS == 0:
                    last_snapshot_iter = it
                    ss, nn = self.snapshot(it)
                    np_paths.append(nn); ss_paths.append(ss)
                    self.remove_snapshot(np_paths, ss_paths)
                it += 1
                if it >= max_iters + 1:
                    break
        if last_snapshot_iter != it - 1:
            self.snapshot(it - 1)


def train_net(network, loader, output_dir, tb_dir, pretrained_model=None, max_iters=40000, rank=0, world=1, solver_cfg=None):
    """TV:477-490.  `solver_cfg`: the config object of the variant's solver (default: nets/variants.solver_cfg(network.variant))."""
    sw = SolverWrapper(network, loader, output_dir, tb_dir, pretrained_model=pretrained_model, rank=rank, world=world, solver_cfg=solver_cfg)
    print('Solving...')
    sw.train_model(max_iters)
    print('done solving')
    return sw
