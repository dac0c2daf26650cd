"""SGD with momentum over the flat parameter buffer — the optimiser the reference builds at
pyutils/mask-faster-rcnn/lib/model/train_val_cycle.py:194-220 (one param group per tensor, weight decay on
non-bias tensors, optional DOUBLE_BIAS), fused into one HIP launch that also rewrites the dtype
shadow weights; the data-gradient (transposed) weight copies are refreshed right after."""
from . import ops as O


class SGD(object):
    def __init__(self, net, lr, momentum=0.9, weight_decay=1e-4, grad_scale=1.0):
        self.net = net
        self.lr, self.momentum, self.weight_decay, self.grad_scale = float(lr), float(momentum), float(weight_decay), float(grad_scale)
        self.param_groups = [self.__dict__]        # scale_lr()-style code can do group['lr'] *= gamma

    def zero_grad(self):
        pass                                        # gradients are zeroed at the start of forward_backward

    def step(self):
        P = self.net.P
        if hasattr(self.net, 'join_wgrad'):
            self.net.join_wgrad()                   # weight-gradient stream -> current stream
        O.sgd_momentum(P.param, P.grad, P.mom, P.segs_dev, P.nseg, P.rowscale, self.lr, self.momentum, self.weight_decay,
                       self.grad_scale, shadow=P.shadow)
        self.net.refresh_weights()

    def state_dict(self):
        return {'lr': self.lr, 'momentum': self.momentum, 'weight_decay': self.weight_decay}
