"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI.  The reference has no distributed code at all; the natural sharding
is one (image, expression) pair per rank per step (train_val_cycle.py:362-403), with a sum
all-reduce of the flat fp32 gradient buffer averaged over ranks.

The flat buffer is laid out in reverse execution order (nets/params.py), so `ready(stage)` can launch
the all-reduce of a finished bucket on a side stream while earlier layers are still back-propagating;
`finish()` joins before the optimiser."""
import numpy as np
import torch
import torch.distributed as dist



def bucket_bounds(P):
    """prefix of the flat parameter / gradient buffer that is final after each backward stage (the buffer is laid out in
    reverse execution order, nets/params.py).  Used by the gradient reducer and by the optimiser's early partial updates."""
    def end_of(pred):
        e = 0
        for k in P.trainable:
            if pred(k):
                e = max(e, P.offsets[k] + int(np.prod(P.shapes[k])))
        return (e + 63) // 64 * 64
    # layer4 weights get gradient from both the caption pass and the RoI pass -> final only after 'heads'
    bounds = {
        'caption': end_of(lambda k: k.startswith('caption_model.')),
        'heads': end_of(lambda k: k.startswith(('caption_model.', 'resnet.layer4.', 'cls_score', 'bbox_pred', 'mask_'))),
        'language': end_of(lambda k: not k.startswith(('resnet.layer3.', 'resnet.layer2.', 'resnet.layer1.', 'vgg.features.'))),
        'layer3': end_of(lambda k: not k.startswith(('resnet.layer2.', 'resnet.layer1.'))),
        'layer2': end_of(lambda k: not k.startswith('resnet.layer1.')),
        'layer1': P.total,
    }
    # layer3 is the largest stage (23 blocks, ~100 MB of gradients): it is handed over in three pieces so that its all-reduce
    # overlaps with the rest of its own backward pass.  'layer3:b' = everything down to (and including) block b.
    for b in (16, 8):
        bounds['layer3:%d' % b] = end_of(lambda k, b=b: not k.startswith(('resnet.layer2.', 'resnet.layer1.')) and
                                         not (k.startswith('resnet.layer3.') and int(k.split('.')[2]) < b))
    return bounds


def replay_segments(stages, run_segment, dp):
    """data-parallel replay of a launch tape cut at the bucket hand-offs (Network.tape_step): segment i, then the bucket that segment
    i completed goes to the reducer (torch.distributed cannot be recorded), ..., the last hand-off is the join before the optimiser, and
    the final segment holds the update.  `run_segment(i)` issues the launches of segment i."""
    for i, stage in enumerate(stages):
        run_segment(i)
        if stage == 'finish':
            dp.finish()
        else:
            dp.ready(stage)
    run_segment(len(stages))


def shard_plan(P, lo, hi, min_shadow_run=1 << 18):
    """the sub-buckets of the gradient bucket [lo, hi) for the sharded update: [(a, b, on_wire)] with on_wire = 'shadow' where EVERY tensor
    of [a, b) is read through the dtype shadow only (ParamStore.shadow_only: the bf16 convolution weights) - its all-gather carries the
    shadow, half the bytes, and the fp32 masters of the other ranks' slices stay behind until gather_master() - and 'master' elsewhere: the
    all-gather carries the fp32 master weights and every rank rewrites the shadow of the range from them.  'master' is right for any
    tensor, so a shadow-only run shorter than `min_shadow_run` elements joins its neighbours instead of paying for its own collective."""
    if not hasattr(P, 'shadow_only_runs'):
        return [(lo, hi, 'master')]
    cut = []
    for a, b, so in P.shadow_only_runs():
        a, b = max(a, lo), min(b, hi)
        if b > a:
            cut.append([a, b, 'shadow' if (so and b - a >= min_shadow_run) else 'master'])
    out = []
    for c in cut:
        if out and out[-1][2] == c[2]:
            out[-1][1] = c[1]
        else:
            out.append(c)
    return [tuple(c) for c in out]


class GradReducer(object):
    """Bucketed gradient exchange on a side stream.

    wire:  'fp32' - the bucket is reduced in place (282 MB per step for the ResNet cycle network);
           'bf16' - the bucket is packed to bf16 (one cast launch), reduced, and unpacked back into the fp32 gradient buffer: half the
                    bytes on every xGMI link (141 MB).  Every partial sum of the collective is rounded to bf16, so the gradients agree
                    with the fp32 exchange to ~2^-8 relative per rank added; the master weights, momentum and update stay fp32.
    algo:  'allreduce'  - one all-reduce per bucket (RCCL picks ring / tree);
           'rs_ag'      - reduce-scatter + all-gather per bucket: each rank reduces 1/world of the bucket, which RCCL can run as direct
                          exchanges over all seven xGMI links of the fully connected node instead of a ring that is bound by one link
                          (SURVEY.md section 8e).  Same sums as the all-reduce up to the order of the additions.
    Queue / CU budget: the exchange owns ONE extra stream (`self.side`); RCCL's kernels use its default channel count (a few workgroups
    per channel) - no CU mask, no priority (both measured harmful on this stack).
    `timing=True` records HIP events around every bucket and around the wait in finish(); `report()` returns per-bucket durations and
    the exposed wait (what the main stream actually stalled for) of the last step."""
    STAGES = ['caption', 'heads', 'language', 'layer3', 'layer2', 'layer1']
    # hand-offs that are NOT taken (Network.dp_ready): their gradients ride with the next stage's bucket - fewer, larger collectives and fewer
    # cuts of the launch tape.  A comma list or a tuple; '' = every hand-off (seven buckets per step).
    SKIP_STAGES = ('caption', 'layer3:16')     # five buckets per step: caption + heads | language | layer3 blocks 22-8 | layer3 blocks 7-0 | layer2 (round 6)
    shard_g16 = True             # bf16 wire: the sharded update reads the reduce-scattered bf16 shard directly (False: cast back to f32 first)

    def __init__(self, net, world, backend_stream=True, skip_allreduce=0, wire='fp32', algo='allreduce', timing=False, shard_update=None, rank=None,
                 bucket_update=None):
        assert wire in ('fp32', 'bf16') and algo in ('allreduce', 'rs_ag')
        assert shard_update is None or algo == 'rs_ag', 'the sharded update rides on reduce-scatter + all-gather'
        assert shard_update is None or bucket_update is None
        self.net, self.world = net, world
        # Update per bucket (round 4, unsharded): as soon as a bucket's gradients are summed, the optimiser updates that bucket on the reducer's
        # stream (`bucket_update.update_range(lo, hi, full=True)`: weights, momentum, dtype shadow, gradient clear) - data parallel's form of the
        # single-process early partial updates (optim.SGD.early): only the last, smallest bucket's update trails the step.  True = "the optimiser
        # built for this network binds itself" (optim.SGD.__init__).
        self.bucket_update = bucket_update
        # Sharded update (round 4): with algo = 'rs_ag' the all-gather does not have to carry GRADIENTS.  Each rank keeps the slice of the bucket
        # the reduce-scatter left it with, runs the optimiser on that slice only (`shard_update.update_range(lo, hi)`: fp32 master weights and
        # momentum of the slice; the 1 / world averaging is the optimiser's grad_scale) and the ranks all-gather the updated WEIGHTS - the
        # same bytes as gathering fp32 gradients, the update's HBM traffic divided by `world`, and the update of a bucket overlaps with
        # the rest of the backward pass instead of trailing the step.  Momentum of the other slices is never read on this rank.
        self.shard_update = shard_update
        # Round 5: what the all-gather of a sharded bucket carries is the dtype SHADOW the kernels read (bf16(scale * w) in the benchmarked mode), not
        # the fp32 master weights: half the bytes on the wire, and the shadow-only pass over the whole buffer at the end of the step is gone -
        # every rank writes the shadow of its slice in the update and receives the others'.
        # Round 6: that is only right for tensors NO kernel reads as fp32 master.  The language encoder, the captioner, every
        # bias, the dynamic-filter FCs, mask_pred / mask_up_sampling are read from ParamStore.param directly, and in round 5 the other ranks'
        # slices of those stayed at their initial values on every rank (the ranks diverged; the world-1 GPU test and the shadow-only gloo check
        # could not see it).  Now every bucket is cut into sub-buckets by what its tensors are read through (shard_plan): 'shadow' sub-buckets
        # (the bf16 convolution weights: 47 of 73 M elements of the cycle network) gather the shadow as before, 'master' sub-buckets gather the
        # fp32 masters and rewrite their shadow locally.  Per step on the wire: 94 MB of shadow + 104 MB of masters (280 MB if every all-gather
        # carried masters).  The fp32 masters of a 'shadow' slice then live on its owner ONLY; gather_master() (a collective: every rank calls
        # it, model/train_val.py before a snapshot, Network.state_dict through it) brings them together again.
        self.gather_shadow = shard_update is not None and getattr(net.P, 'shadow', None) is not None
        self._host_staged = None     # device buffers on a gloo process group: collectives staged through the host (decided at the first collective)
        self._parts = {}             # sub-bucket lo -> (m, per): the partition of every sharded sub-bucket
        self._plans = {}             # bucket (lo, hi) -> shard_plan(...)
        self._stale = []             # (lo, m, per) of the sub-buckets whose all-gather carried the shadow since the last gather_master()
        self.master_stale = False    # other ranks' slices of P.param are behind (until gather_master())
        self.rank = rank if rank is not None else (dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0)
        # experiment only (bench.py --dp-skip-allreduce): 1 = keep the stream structure but issue no collective, 2 = do nothing.
        # Ranks diverge with either, so model/train_val.py refuses a reducer built this way.
        self.skip_allreduce = int(skip_allreduce)
        self.wire, self.algo, self.timing = wire, algo, bool(timing)
        P = net.P
        self.bounds = bucket_bounds(P)
        self.done = 0
        self.on_gpu = P.grad.is_cuda
        self.side = torch.cuda.Stream() if self.on_gpu else None
        self._pack = None            # bf16 staging buffer (whole flat length: buckets are slices of it)
        self._shard = None           # reduce-scatter output (largest bucket / world)
        self._wshard = None          # this rank's updated weights of a bucket, the all-gather's input
        self._events = []            # (stage, start, end) of the last step
        self._wait_events = None

    # ---- the three collectives.  On RCCL (backend "nccl") they run on the device, on the current stream.  With device buffers on a "gloo" process
    # group they are STAGED THROUGH THE HOST (a blocking copy out, the collective on CPU tensors, a copy back): no xGMI, no overlap - a debug
    # transport that lets two real ranks share ONE GPU (tests/test_train_step_gpu.py::test_two_ranks_one_gpu_end_to_end; TRAIN.DP_BACKEND = 'gloo')
    def _staged(self):
        if self._host_staged is None:
            self._host_staged = bool(self.on_gpu and dist.is_available() and dist.is_initialized() and dist.get_backend() == 'gloo')
        return self._host_staged

    def _rs(self, out, inp):
        if self._staged():
            h = inp.float().cpu()                              # (gloo sums fp32; a bf16 wire is rounded to bf16 again on the way back, as every RCCL partial sum is)
            o = torch.empty(out.numel(), dtype=torch.float32)
            dist.reduce_scatter_tensor(o, h, op=dist.ReduceOp.SUM)
            out.copy_(o.to(out.dtype))
        else:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)

    def _ag(self, out, inp):
        if self._staged():
            h = inp.cpu().contiguous()
            hv = h.view(torch.uint8)                             # (bit patterns: a gather must not re-round anything, and gloo has no bf16)
            o = torch.empty(out.numel() * out.element_size(), dtype=torch.uint8)
            dist.all_gather_into_tensor(o, hv)
            out.copy_(o.view(h.dtype).to(out.device))
        else:
            dist.all_gather_into_tensor(out, inp)

    def _ar(self, t):
        if self._staged():
            h = t.float().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h.to(t.dtype))
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    # ---- one bucket, on the current stream (the side stream on the GPU) ----
    def _cast(self, src, dst):
        if self.on_gpu:
            from . import ops as O
            O.cast(src, dst)                         # the library's cast kernel (bf16 <-> f32, round to nearest even)
        else:
            dst.copy_(src)                           # gloo tests on the CPU box

    def _collective(self, buf):
        """sum over ranks of `buf` (a contiguous 1-D tensor), in place"""
        W = self.world
        n = buf.numel()
        if self.algo == 'rs_ag' and W > 1 and n >= W:
            m = n // W * W                           # the part that splits evenly; a tail of < world elements goes through all_reduce
            per = m // W
            if self._shard is None or self._shard.numel() < per or self._shard.dtype != buf.dtype:
                big = max(per, (max(self.bounds.values()) + W - 1) // W)
                self._shard = torch.empty(big, dtype=buf.dtype, device=buf.device)
            sh = self._shard[:per]
            self._rs(sh, buf[:m])
            self._ag(buf[:m], sh)
            if m < n:
                self._ar(buf[m:])
        else:
            self._ar(buf)

    def _exchange_sharded(self, seg, lo, hi, on_wire='shadow'):
        """bucket [lo, hi): reduce-scatter the gradients, update this rank's slice, all-gather the weights (the < 4 world elements that do not
        split evenly are all-reduced and updated by every rank)"""
        W, r, P = self.world, self.rank, self.net.P
        n = hi - lo
        if self.wire == 'bf16' and self._pack is None:
            self._pack = torch.empty(P.total, dtype=torch.bfloat16, device=seg.device)
        m = n // (4 * W) * (4 * W)                          # slices start at multiples of four elements (the update kernel's vector width)
        per = m // W
        if m:
            if self.wire == 'bf16':
                src = self._pack[lo:lo + m]
                self._cast(seg[:m], src)
            else:
                src = seg[:m]
            if self._shard is None or self._shard.numel() < per or self._shard.dtype != src.dtype:
                big = max(per, (max(self.bounds.values()) + W - 1) // W)
                self._shard = torch.empty(big, dtype=src.dtype, device=src.device)
            sh = self._shard[:per]
            self._rs(sh, src)
            own = seg[r * per:(r + 1) * per]
            # bf16 wire on the device: the update reads the reduce-scattered shard as it is (l2s_sgd_momentum_range_g16) - no cast back into
            # the f32 gradient buffer, whose slice keeps this rank's local gradients until the next step overwrites them
            g16 = sh if (self.wire == 'bf16' and self.on_gpu and self.shard_g16) else None
            if g16 is None:
                self._cast(sh, own) if self.wire == 'bf16' else own.copy_(sh)
            kw = dict(grad_bf16=g16) if g16 is not None else {}
            self._parts[lo] = (m, per)
            if self.gather_shadow and on_wire == 'shadow':
                self.shard_update.update_range(lo + r * per, lo + (r + 1) * per, shadow=True, **kw)
                wsl = P.shadow[lo:lo + m]
                self.master_stale = W > 1
                self._stale.append((lo, m, per))
            else:
                # (masters on the wire: this rank's slice gets its shadow from the update itself, the other ranks' slices from a ranged rewrite behind the gather)
                self.shard_update.update_range(lo + r * per, lo + (r + 1) * per, shadow=bool(self.gather_shadow), **kw)
                wsl = P.param[lo:lo + m]
            if self.on_gpu:
                # in place: this rank's slice already sits where the gathered buffer wants it (RCCL's in-place all-gather: send = recv + rank * count)
                self._ag(wsl, wsl[r * per:(r + 1) * per])
            else:
                if self._wshard is None or self._wshard.numel() < per or self._wshard.dtype != wsl.dtype:
                    self._wshard = torch.empty(max(per, (max(self.bounds.values()) + W - 1) // W), dtype=wsl.dtype, device=wsl.device)
                mine = self._wshard[:per]
                mine.copy_(wsl[r * per:(r + 1) * per])
                self._ag(wsl, mine)
            if self.gather_shadow and on_wire != 'shadow':
                if r > 0:
                    self.shard_update.refresh_shadow_range(lo, lo + r * per)              # the shadow of the masters the other ranks sent
                if r < W - 1:
                    self.shard_update.refresh_shadow_range(lo + (r + 1) * per, lo + m)
        if m < n:
            tail = seg[m:]
            if self.wire == 'bf16':
                tb = self._pack[lo + m:hi]
                self._cast(tail, tb); self._ar(tb); self._cast(tb, tail)
            else:
                self._ar(tail)
            # (every rank updates the tail from the all-reduced gradients: master and shadow are both current everywhere)
            self.shard_update.update_range(lo + m, hi, shadow=bool(self.gather_shadow))

    def gather_master(self):
        """sharded update with the shadow on the wire: all-gather the fp32 master weights of every bucket (each rank contributes the slices it owns).
        A COLLECTIVE - every rank calls it at the same point (before a snapshot, an evaluation on the master weights, a state_dict()); cheap to call
        when nothing is stale."""
        if not (self.gather_shadow and self.master_stale) or self.world <= 1:
            self.master_stale = False
            return
        P, W, r = self.net.P, self.world, self.rank
        if self.on_gpu:
            torch.cuda.synchronize()
        for lo, m, per in sorted(set(self._stale)):
            wsl = P.param[lo:lo + m]
            mine = wsl[r * per:(r + 1) * per].clone()
            self._ag(wsl, mine)
        if self.on_gpu:
            torch.cuda.synchronize()
        self.master_stale = False
        del self._stale[:]

    def stale_master_ranges(self):
        """[(lo, hi)] of the flat buffer whose fp32 masters this rank does NOT hold current (other ranks' slices of the sub-buckets whose
        all-gather carried the shadow) - until gather_master()"""
        out = []
        for lo, m, per in sorted(set(self._stale)):
            for j in range(self.world):
                if j != self.rank:
                    out.append((lo + j * per, lo + (j + 1) * per))
        return out

    def _exchange(self, seg, lo, hi):
        if self.shard_update is not None:
            if (lo, hi) not in self._plans:
                self._plans[(lo, hi)] = shard_plan(self.net.P, lo, hi) if self.gather_shadow else [(lo, hi, 'master')]
            for a, b, on_wire in self._plans[(lo, hi)]:
                self._exchange_sharded(seg[a - lo:b - lo], a, b, on_wire)
            return
        if self.wire == 'bf16':
            if self._pack is None:
                self._pack = torch.empty(self.net.P.total, dtype=torch.bfloat16, device=seg.device)
            pk = self._pack[lo:hi]
            self._cast(seg, pk)
            self._collective(pk)
            self._cast(pk, seg)
        else:
            self._collective(seg)
        if self.bucket_update is not None and self.bucket_update is not True:
            self.bucket_update.update_range(lo, hi, full=True)

    def skips(self, stage):
        sk = self.SKIP_STAGES
        return stage in (tuple(x for x in sk.split(',') if x) if isinstance(sk, str) else tuple(sk))

    def ready(self, stage):
        if self.skip_allreduce == 2 or self.skips(stage):
            return
        end = min(self.bounds[stage], self.net.P.total)
        if end <= self.done:
            return
        lo = self.done
        seg = self.net.P.grad[lo:end]
        if self.on_gpu:
            self.side.wait_stream(torch.cuda.current_stream())
            if getattr(self.net, 'use_streams', False) and hasattr(self.net, '_streams'):
                for name in ('wg', 'wg2', 'lang', 'cap'):           # gradients are also produced on the side streams: the reducer
                    self.side.wait_stream(self.net._streams[name])  # waits for them, the main stream does not have to
            with torch.cuda.stream(self.side):
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record()
                if not self.skip_allreduce:
                    self._exchange(seg, lo, end)
                if self.timing:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record()
                    self._events.append((stage, end - lo, e0, e1))
        else:                                   # CPU/gloo path (tests)
            self._exchange(seg, lo, end)
        self.done = end

    def finish(self):
        if self.skip_allreduce == 2:
            return
        P = self.net.P
        if self.done < P.total:
            self.ready('layer1')
        if self.on_gpu:
            main = torch.cuda.current_stream()
            # who waits for the last bucket: the main stream - or, when the buckets were updated here (whole, or this rank's slices) and the
            # optimiser's tail runs on the weight-gradient stream (optim.SGD.side), that stream: the next step's frozen prefix then starts beside the last update and the main
            # stream joins before its first trainable layer (Network.join_update), as in the single-process step
            upd = self.bucket_update if self.bucket_update not in (None, True) else self.shard_update
            if upd not in (None, True) and getattr(upd, 'side_active', False) and getattr(self.net, 'use_streams', False):
                main = self.net.streams()['wg']
            if self.timing:
                a = torch.cuda.Event(enable_timing=True); a.record(main)
            main.wait_stream(self.side)
            if self.timing:
                b = torch.cuda.Event(enable_timing=True); b.record(main)
                self._wait_events = (a, b)
                self._last_events, self._events = self._events, []
        self.done = 0
        # average over ranks: folded into the optimiser's grad_scale by the caller

    def report(self):
        """after a device sync: per-bucket exchange time and the exposed wait of the last finished step (timing=True)"""
        if not self.timing or self._wait_events is None:
            return None
        esz = 2 if self.wire == 'bf16' else 4
        return {'wire': self.wire, 'algo': self.algo,
                'buckets': [{'stage': st, 'mbytes': n * esz / 1e6, 'ms': e0.elapsed_time(e1)} for st, n, e0, e1 in self._last_events],
                'exposed_wait_ms': self._wait_events[0].elapsed_time(self._wait_events[1])}
