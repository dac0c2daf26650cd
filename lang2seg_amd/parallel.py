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


class GradReducer(object):
    STAGES = ['caption', 'heads', 'language', 'layer3', 'layer2', 'layer1']

    def __init__(self, net, world, backend_stream=True, skip_allreduce=0):
        self.net, self.world = net, world
        # experiment only (bench.py --dp-skip-allreduce): 1 = keep the stream structure but issue no collective, 2 = do nothing.
        # Ranks diverge with either, so model/train_val.py refuses a reducer built this way.
        self.skip_allreduce = int(skip_allreduce)
        P = net.P
        self.bounds = bucket_bounds(P)
        self.done = 0
        self.on_gpu = P.grad.is_cuda
        self.side = torch.cuda.Stream() if self.on_gpu else None

    def ready(self, stage):
        if self.skip_allreduce == 2:
            return
        end = min(self.bounds[stage], self.net.P.total)
        if end <= self.done:
            return
        seg = self.net.P.grad[self.done:end]
        if self.on_gpu:
            self.side.wait_stream(torch.cuda.current_stream())
            if getattr(self.net, 'use_streams', False) and hasattr(self.net, '_streams'):
                for name in ('wg', 'wg2', 'lang', 'cap'):           # gradients are also produced on the side streams: the reducer
                    self.side.wait_stream(self.net._streams[name])  # waits for them, the main stream does not have to
            with torch.cuda.stream(self.side):
                if not self.skip_allreduce:
                    dist.all_reduce(seg, op=dist.ReduceOp.SUM)
        else:                                   # CPU/gloo path (tests)
            dist.all_reduce(seg, op=dist.ReduceOp.SUM)
        self.done = end

    def finish(self):
        if self.skip_allreduce == 2:
            return
        P = self.net.P
        if self.done < P.total:
            self.ready('layer1')
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.side)
        self.done = 0
        # average over ranks: folded into the optimiser's grad_scale by the caller
