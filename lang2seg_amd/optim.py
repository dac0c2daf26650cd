"""SGD with momentum over the flat parameter buffer — the optimiser the reference builds at
pyutils/mask-faster-rcnn/lib/model/train_val_cycle.py:194-220 (one param group per tensor, weight decay on
non-bias tensors, optional DOUBLE_BIAS), fused into one HIP launch that also rewrites the dtype
shadow weights; the data-gradient (transposed) weight copies are refreshed right after."""
import bisect

import torch

from . import ops as O


class SGD(object):
    def __init__(self, net, lr, momentum=0.9, weight_decay=1e-4, grad_scale=1.0):
        self.net = net
        self.lr, self.momentum, self.weight_decay, self.grad_scale = float(lr), float(momentum), float(weight_decay), float(grad_scale)
        self.param_groups = [self.__dict__]        # scale_lr()-style code can do group['lr'] *= gamma
        # decided once, before the first step: the edges of this mode are part of every recorded launch tape
        self.side_active = bool(self.side and getattr(net, 'use_streams', False) and hasattr(net, 'flush_wgrads'))
        if self.side_active:
            net.update_on_wg = True

    def zero_grad(self):
        pass                                        # gradients are zeroed at the start of forward_backward

    # Early partial updates: the flat buffer is in reverse execution order, so once a backward stage has finished a prefix of
    # it is final (parallel.bucket_bounds).  `partial(stage)` applies the update to that prefix on the idle transpose stream
    # while the earlier layers are still back-propagating (the update is HBM-bound, the convolutions are not); `step()` then
    # only has the last layers left.  Single-process only: with a gradient reducer the prefix still has to be all-reduced.
    # Round 1 measured it neutral (122.4 / 123.6 img/s with it vs 123.1 without) - but that form made the MAIN queue wait for the
    # transpose stream at every hand-off, i.e. for the weight-gradient launches the partial update itself waits for.  Since round 3 the
    # hand-off only forks, and whatever is left at the end of the step follows the last weight gradients on their stream (SGD.side).
    early = False                                  # set by tests / tools before the first step
    _seg_done = 0

    def _launch(self, s0, s1):
        P = self.net.P
        if s1 > s0:
            O.sgd_momentum(P.param, P.grad, P.mom, P.segs_dev[s0 * P.seg_size:], s1 - s0, P.rowscale, self.lr, self.momentum,
                           self.weight_decay, self.grad_scale, shadow=P.shadow)

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

    def step(self):
        P = self.net.P
        net = self.net
        if self.side_active and net.use_streams:
            net.flush_wgrads('final')
            S = net.streams()
            for k in ('wg2', 'lang', 'cap'):
                net.sfork(S[k], S['wg'])
            if self._seg_done:
                net.sfork(S['tr'], S['wg'])               # early partial updates of this step (SGD.early) ran on the transpose stream
            net.sfork(torch.cuda.current_stream(), S['wg'])
            with torch.cuda.stream(S['wg']):
                self._launch(self._seg_done, P.nseg)      # whatever the partial updates left: the last backward stages
                net.refresh_weights()
                net._mark('update done (wg)')
            self._seg_done = 0
            return
        if hasattr(self.net, 'join_wgrad'):
            self.net.join_wgrad()                   # weight-gradient stream -> current stream
        if self._seg_done:
            self.net.sfork(self.net.streams()['tr'], torch.cuda.current_stream())
        self._launch(self._seg_done, P.nseg)
        self._seg_done = 0
        self.net.refresh_weights()

    def state_dict(self):
        return {'lr': self.lr, 'momentum': self.momentum, 'weight_decay': self.weight_decay}
