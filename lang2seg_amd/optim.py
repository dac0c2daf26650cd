"""SGD with momentum over the flat parameter buffer — the optimiser the reference builds at
pyutils/mask-faster-rcnn/lib/model/train_val_cycle.py:194-220 (one param group per tensor, weight decay on
non-bias tensors, optional DOUBLE_BIAS), fused into one HIP launch that also rewrites the dtype
shadow weights; the data-gradient (transposed) weight copies are refreshed right after."""
import bisect

import torch

from . import ops as O


class SGD(object):
    def __init__(self, net, lr, momentum=0.9, weight_decay=1e-4, grad_scale=1.0, keep_grad=False):
        self.net = net
        self.lr, self.momentum, self.weight_decay, self.grad_scale = float(lr), float(momentum), float(weight_decay), float(grad_scale)
        self.param_groups = [self.__dict__]        # scale_lr()-style code can do group['lr'] *= gamma
        # decided once, before the first step: the edges of these modes are part of every recorded launch tape
        self.side_active = bool(self.side and getattr(net, 'use_streams', False) and hasattr(net, 'flush_wgrads'))
        if self.side_active:
            net.update_on_wg = True
        self._early = bool(type(self)._early and not self.defer)      # (the deferred heads stage, when asked for, excludes the early partial updates)
        self.defer_active = bool(self.defer and self.side_active and not self._early and hasattr(net, 'wgq') and getattr(net, 'dp', None) is None)
        net.defer_heads = self.defer_active
        # the plain side-stream update records when the LAST segments (layer2) are done: the next step's layer2 waits for that only
        net.update_split = bool(self.side_active and getattr(net, 'dp', None) is None and not self.defer_active)
        # layer2's weight gradients + the update of what the early partial updates left, on 'wg2': only when what is left IS layer2 alone, i.e.
        # layer2 is the last trainable stage (FIXED_BLOCKS = 1).  With FIXED_BLOCKS = 0 the tail also holds layer1, whose weight gradients run
        # on 'wg': an update on 'wg2' would race them.  (FIXED_BLOCKS >= 2 / VGG: no 'layer2' flush at all.)
        net.layer2_side = bool(net.update_split and self.layer2_side and getattr(net.P, 'fixed_blocks', 1) == 1 and not getattr(net.P, 'is_vgg', False))
        # keep_grad=False: the update kernel zeroes every gradient it consumes (optimizer.zero_grad(), TV:383, folded in), and
        # forward_backward no longer clears the buffer (it is zero when the network is built, and every update leaves it zero; a second
        # backward pass without an update in between is refused); keep_grad=True leaves the step's gradients in P.grad (tests read them there)
        dp = getattr(net, 'dp', None)
        if dp is not None and getattr(dp, 'shard_update', None) is True:
            dp.shard_update = self                   # the reducer calls update_range() on this rank's slice of every bucket
        if dp is not None and getattr(dp, 'bucket_update', None) is True:
            dp.bucket_update = self                  # ... or on every whole bucket, right behind its all-reduce
        self.in_reducer = dp is not None and (getattr(dp, 'shard_update', None) is self or getattr(dp, 'bucket_update', None) is self)
        self.clear_grad = not keep_grad and getattr(dp, 'shard_update', None) is None   # (a sharded update touches a slice only)
        net.update_clears_grad = self.clear_grad

    def zero_grad(self):
        pass                                        # gradients are zeroed at the start of forward_backward

    # Early partial updates: the flat buffer is in reverse execution order, so once a backward stage has finished a prefix of
    # it is final (parallel.bucket_bounds).  `partial(stage)` applies the update to that prefix on the idle transpose stream
    # while the earlier layers are still back-propagating (the update is HBM-bound, the convolutions are not); `step()` then
    # only has the last layers left.  Single-process only: with a gradient reducer the prefix still has to be all-reduced.
    # Round 1 measured it neutral (122.4 / 123.6 img/s with it vs 123.1 without) - but that form made the MAIN queue wait for the
    # transpose stream at every hand-off, i.e. for the weight-gradient launches the partial update itself waits for.  Since round 3 the
    # hand-off only forks, and whatever is left at the end of the step follows the last weight gradients on their stream (SGD.side).
    # Round 4: ON by default (one rank).  With the update kernel as one persistent workgroup per CU (shallow memory queues) and gradients
    # that are overwritten instead of cleared, the partial updates beside the layer3 / layer2 backward leave 0.13 ms of update at the
    # end of the step instead of 0.29 ms: 199.3 -> 202.7 img/s, same box x 3 (profiles/r04_sgd_early_ab.txt).
    _early = True
    _seg_done = 0

    @property
    def early(self):
        return self._early

    @early.setter
    def early(self, v):                            # set by tests / tools before the first step (the deferred heads stage excludes it)
        self._early = bool(v)
        if self._early and getattr(self, 'defer_active', False):
            self.defer_active = False
            self.net.defer_heads = False

    def _launch(self, s0, s1, table=None):
        P = self.net.P
        if s1 > s0:
            O.sgd_momentum(P.param, P.grad, P.mom, (P.segs_dev if table is None else table)[s0 * P.seg_size:], s1 - s0, P.rowscale, self.lr,
                           self.momentum, self.weight_decay, self.grad_scale, shadow=P.shadow, clear_grad=self.clear_grad)

    def partial(self, stage):
        net = self.net
        P = net.P
        if not hasattr(self, '_bounds'):
            from .parallel import bucket_bounds
            self._bounds = bucket_bounds(P)
            assert all(a <= b for a, b in zip(P.seg_ends, P.seg_ends[1:]))
        hi = bisect.bisect_right(P.seg_ends, self._bounds[stage])
        if hi <= self._seg_done:
            return
        # the marks of the previous complete step hold for this prefix only if its marked tensors were written again (same schedule: always)
        stale = P.stale_marked(self._seg_done, hi, getattr(net, '_fresh', ()))
        if stale:
            P.mark_overwritten(P._ow_key - frozenset(stale))
        S = net.streams()
        tr = S['tr']
        net.sfork(torch.cuda.current_stream(), tr)
        for name in ('wg', 'wg2', 'lang', 'cap'):                # gradients are also produced on the side streams
            net.sfork(S[name], tr)
        with torch.cuda.stream(tr):
            self._launch(self._seg_done, hi)
            net._mark('partial update %s (tr)' % stage)
        self._seg_done = hi

    # Update on the weight-gradient stream: the main queue does not wait for the last grouped weight-gradient launch and the ~0.28 ms
    # HBM-bound update; the next step's frozen prefix (stem, layer1: small dependent launches) starts beside them and the main queue
    # joins before its first trainable layer (Network.join_update).  No extra stream: the update simply follows the weight gradients
    # it depends on in stream order.
    # Measured (bench.py, 200 steps, A/B alternating in one box): 165.8 vs 163.0 img/s.  SGD.side = False (before construction) restores the update on the caller's stream.
    side = True

    # Deferred heads stage (round 4).  The grouped weight-gradient launches of the heads stage (att_embed, layer4 on the RoIs and on the
    # map, RoI / mask heads, RPN: ~0.7 ms of chip-filling kernels) used to start at the caption join and run beside the layer3 / layer2
    # data-gradient chain - 140 short dependent launches that then waited for CU slots (0.66 ms alone, 1.67 ms beside them).  Nothing reads
    # those gradients before the update, and nothing reads the updated weights before the NEXT step has finished its backbone forward.  So
    # the step ends with: [weight gradients of layer3 / layer2] -> update of everything OUTSIDE ParamStore.defer_range -> mark
    # SLOT_UPDATE_REST -> [the held-back weight gradients] -> update of defer_range -> transposes, all on the weight-gradient stream; the
    # next step's layer2 waits for the mark only (Network.join_update(full=False)), its dynamic filters for the whole stream
    # (Network.join_deferred()).  The launches are the same and so is their order per tensor: weights are bit-identical to the undeferred
    # step's (tests/test_train_step_gpu.py::test_deferred_heads_bit_identical).  There is no host-side pending state: the tail is enqueued
    # by this call, and every reader outside the step (state_dict, TEST mode, snapshots) joins the whole weight-gradient stream.
    layer2_side = True   # with the split update: layer2's weight gradients + update on 'wg2' instead of behind the backlog of 'wg' (Network.layer2_side)
    defer = False   # measured (round 4, same-box A/B x3): 180.7 img/s with it, 184.1 without - see DESIGN.md section 4.6

    def _mark_overwritten(self):
        # gradients the grouped weight gradients of this step wrote whole are not cleared by the update (ParamStore.mark_overwritten): marked
        # at the end of a step, behind every weight gradient, for this step's last launch and the steps that follow (early partial updates
        # check their prefix against the marks, partial()); not with the deferred heads stage, whose gradients arrive after this point.
        net = self.net
        if self.clear_grad and not self.defer_active and getattr(net, 'wgrad_overwrite', False):
            net.P.mark_overwritten(getattr(net, '_fresh', ()))

    def step(self):
        P = self.net.P
        net = self.net
        net._bwd_pending = 0
        net._pass_without_step = False
        if getattr(getattr(net, 'dp', None), 'shard_update', None) is not None:
            # the reducer updated this rank's slice of every bucket and gathered the others' (parallel.GradReducer): what is left is the dtype
            # shadow of the gathered weights and the data-gradient copies - on the weight-gradient stream where the tail lives there (the reducer
            # made that stream wait for the last all-gather, GradReducer.finish; the next step joins it before its first trainable layer)
            gathered = bool(getattr(net.dp, 'gather_shadow', False))      # the all-gathers carried the shadow itself: nothing to rewrite
            if self.side_active and net.use_streams:
                net.flush_wgrads('final')
                with torch.cuda.stream(net.streams()['wg']):
                    if not gathered:
                        self.refresh_shadow()
                    net.refresh_weights()
                    net._mark('update done (wg)')
                return
            if hasattr(net, 'join_wgrad'):
                net.join_wgrad()
            if not gathered:
                self.refresh_shadow()
            net.refresh_weights()
            return
        if getattr(getattr(net, 'dp', None), 'bucket_update', None) is self:
            # every bucket was updated behind its all-reduce (weights, momentum, shadow, gradient clear): the data-gradient copies are left.
            # With the tail on the weight-gradient stream the reducer made THAT stream wait for the last bucket (GradReducer.finish)
            if self.side_active and net.use_streams:
                net.flush_wgrads('final')
                self._mark_overwritten()
                with torch.cuda.stream(net.streams()['wg']):
                    net.refresh_weights()
                    net._mark('update done (wg)')
                return
            if hasattr(net, 'join_wgrad'):
                net.join_wgrad()
            self._mark_overwritten()
            net.refresh_weights()
            return
        if self.side_active and net.use_streams:
            net.flush_wgrads('final')
            self._mark_overwritten()
            S = net.streams()
            # what the early partial updates left is layer2 alone (the last segments of the buffer): its gradients come from the main queue and
            # the two weight-gradient streams, so that launch does not wait for the language / caption / transpose streams - they are joined
            # behind it, before the transposes.  The next step's layer2 waits for SLOT_UPDATE_L2 only (Network.join_update(layer2_only=True)).
            early_tail = bool(self._seg_done and net.update_split and not self.defer_active and getattr(net.wgq, 'V5_STREAM', 'wg') != 'tr')
            side2 = bool(early_tail and getattr(net, '_layer2_on_side', False))     # layer2's weight gradients went to 'wg2' (WgradQueue.flush)
            net._layer2_on_side = False
            if not side2:
                net.sfork(S['wg2'], S['wg'])
            if getattr(net.wgq, 'V5_STREAM', 'wg') == 'tr':
                net.sfork(S['tr'], S['wg'])               # (A/B: a weight-gradient launch on the transpose stream)
            O.event_record(net.SLOT_WGRADS, S['wg'])          # behind every weight-gradient launch of this step on 'wg' (/ 'wg2' unless side2)
            if early_tail:
                T = S['wg2'] if side2 else S['wg']            # side2: behind layer2's weight gradients, beside the backlog of 'wg'
                net.sfork(torch.cuda.current_stream(), T)
                with torch.cuda.stream(T):
                    self._launch(self._seg_done, P.nseg)
                    O.event_record(net.SLOT_UPDATE_L2, T)
                    net._mark('update of layer2 done')
                for k in ('wg2', 'lang', 'cap', 'tr'):
                    net.sfork(S[k], S['wg'])
                net.sfork(torch.cuda.current_stream(), S['wg'])
                with torch.cuda.stream(S['wg']):
                    net.refresh_weights()
                    net._mark('update done (wg)')
                self._seg_done = 0
                return
            for k in ('wg2', 'lang', 'cap'):
                net.sfork(S[k], S['wg'])
            if self._seg_done or getattr(net.wgq, 'V5_STREAM', 'wg') == 'tr':
                net.sfork(S['tr'], S['wg'])               # early partial updates of this step (SGD.early) ran on the transpose stream
            net.sfork(torch.cuda.current_stream(), S['wg'])
            with torch.cuda.stream(S['wg']):
                if self.defer_active:
                    self._launch(0, P.n_rest, P.segs_split_dev)
                    O.event_record(net.SLOT_UPDATE_REST, S['wg'])
                    net._mark('update of the rest done (wg)')
                    net.wgq.flush_deferred('heads')
                    net._mark('deferred weight gradients done (wg)')
                    self._launch(P.n_rest, P.nseg, P.segs_split_dev)
                else:
                    self._launch(self._seg_done, P.nseg)  # whatever the partial updates left: the last backward stages
                    O.event_record(net.SLOT_UPDATE_L2, S['wg'])
                net.refresh_weights()
                net._mark('update done (wg)')
            self._seg_done = 0
            return
        if hasattr(self.net, 'join_wgrad'):
            self.net.join_wgrad()                   # weight-gradient stream -> current stream
        if self._seg_done:
            self.net.sfork(self.net.streams()['tr'], torch.cuda.current_stream())
        self._mark_overwritten()
        self._launch(self._seg_done, P.nseg)
        self._seg_done = 0
        self.net.refresh_weights()

    # ---- data parallel, sharded update (parallel.GradReducer(shard_update=...)): a rank updates only ITS slice of a reduce-scattered bucket ----
    def update_range(self, lo, hi, full=False, shadow=False, grad_bf16=None):
        """the update on the elements [lo, hi) of the flat buffer, on the current stream (gradients there are final and summed over ranks).
        full: also rewrite the dtype shadow and clear the gradients consumed (a whole bucket updated on this rank: GradReducer.bucket_update);
        otherwise weights and momentum only (a rank's slice: the shadow follows the all-gather, the clear the next step's memset)."""
        P = self.net.P
        c_lo, c_hi = P.chunk_range(lo, hi)
        if c_hi <= c_lo:
            return
        if full and self.clear_grad:
            # (the overwrite marks are those of the previous complete step: a marked tensor that was not written again is cleared here)
            s0, s1 = bisect.bisect_right(P.seg_ends, lo), bisect.bisect_left(P.seg_offs, hi)
            stale = P.stale_marked(s0, s1, getattr(self.net, '_fresh', ()))
            if stale:
                P.mark_overwritten(P._ow_key - frozenset(stale))
        if grad_bf16 is not None:
            # the slice's summed gradients are the reduce-scattered bf16 shard itself (element o = grad_bf16[o - lo]): no cast back into P.grad
            assert not full
            O.sgd_momentum_range_g16(P.param, grad_bf16, lo, P.mom, P.segs_dev, P.nseg, P.rowscale, self.lr, self.momentum, self.weight_decay, self.grad_scale,
                                     P.shadow if shadow else None, lo, hi, c_lo, c_hi)
            return
        # shadow (a rank's slice whose dtype shadow goes on the wire instead of its weights, GradReducer.gather_shadow): written with the update
        O.sgd_momentum_range(P.param, P.grad, P.mom, P.segs_dev, P.nseg, P.rowscale, self.lr, self.momentum, self.weight_decay, self.grad_scale,
                             P.shadow if (full or shadow) else None, int(bool(full and self.clear_grad)), lo, hi, c_lo, c_hi)

    def refresh_shadow_range(self, lo, hi):
        """dtype shadow of the elements [lo, hi) from the parameters there (a sub-bucket whose fp32 masters the ranks have just gathered)"""
        P = self.net.P
        c_lo, c_hi = P.chunk_range(lo, hi)
        if c_hi > c_lo:
            O.sgd_momentum_range(P.param, P.grad, P.mom, P.segs_dev, P.nseg, P.rowscale, 0.0, 1.0, 0.0, 0.0, P.shadow, 2, lo, hi, c_lo, c_hi)

    def refresh_shadow(self):
        """dtype shadow of every tensor from the (gathered) parameters: shadow = dtype(rowscale * param), no update"""
        P = self.net.P
        O.sgd_momentum_range(P.param, P.grad, P.mom, P.segs_dev, P.nseg, P.rowscale, 0.0, 1.0, 0.0, 0.0, P.shadow, 2, 0, P.total, 0, -1)

    def state_dict(self):
        return {'lr': self.lr, 'momentum': self.momentum, 'weight_decay': self.weight_decay}
