"""Global configuration, same keys / defaults / override API as the reference's
pyutils/mask-faster-rcnn/lib/model/config.py:19-290 (`cfg`, `cfg_from_file` :358-364,
`cfg_from_list` :367-387).  EasyDict is not available in the image, so a minimal
attribute-dict provides the same access patterns (cfg.TRAIN.BATCH_SIZE, cfg['TRAIN'])."""
import numpy as np


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        dict.__setitem__(self, k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


__C = AttrDict()
cfg = __C

__C.TRAIN = AttrDict()
__C.TRAIN.LEARNING_RATE = 1e-4
__C.TRAIN.MOMENTUM = 0.9
__C.TRAIN.WEIGHT_DECAY = 0.0001
__C.TRAIN.GAMMA = 0.1
__C.TRAIN.STEPSIZE = [360000]
__C.TRAIN.DISPLAY = 20
__C.TRAIN.DOUBLE_BIAS = False
__C.TRAIN.TRUNCATED = False
__C.TRAIN.BIAS_DECAY = False
__C.TRAIN.USE_GT = False
__C.TRAIN.ASPECT_GROUPING = False
__C.TRAIN.SNAPSHOT_KEPT = 120
__C.TRAIN.SUMMARY_INTERVAL = 180
__C.TRAIN.SCALES = (600,)
__C.TRAIN.MAX_SIZE = 1000
__C.TRAIN.IMS_PER_BATCH = 1
__C.TRAIN.BATCH_SIZE = 256
__C.TRAIN.FG_FRACTION = 0.25
# (this implementation) size the mask head for BATCH_SIZE RoIs instead of FG_FRACTION * BATCH_SIZE: exact in the no-background case of
# proposal_target_layer.py:155-158 (all sampled RoIs foreground) at 4x the mask-head work; off, that case trains the mask branch on the
# first FG_FRACTION * BATCH_SIZE foreground RoIs only
__C.TRAIN.MASK_SLOTS_ALL = False
__C.TRAIN.FG_THRESH = 0.5
__C.TRAIN.BG_THRESH_HI = 0.5
__C.TRAIN.BG_THRESH_LO = 0.0
__C.TRAIN.USE_FLIPPED = True
__C.TRAIN.BBOX_REG = True
__C.TRAIN.BBOX_THRESH = 0.5
__C.TRAIN.SNAPSHOT_ITERS = 5000
__C.TRAIN.SNAPSHOT_PREFIX = 'res101_mask_rcnn'
# replay steps from per-shape launch tapes in train_net (see model/train_val.py)
__C.TRAIN.USE_TAPE = True
# data parallel: resume from a snapshot that lacks a rank's own sidecar (written by a run with fewer ranks) with freshly seeded loader
# cursors / RNG streams on that rank instead of refusing (model/train_val.py from_snapshot)
__C.TRAIN.ALLOW_RESHARD_RESUME = False
# data parallel (one process per GPU, RCCL over xGMI; parallel.GradReducer): gradient buckets on the wire in bf16 or fp32, one all-reduce per
# bucket or reduce-scatter + all-gather, and (with rs_ag) each rank updating only its slice of a bucket before the weights are gathered
__C.TRAIN.DP_BACKEND = 'nccl'          # torch.distributed backend of tools/train*.py: 'nccl' = RCCL over xGMI; 'gloo' = the reducer's buffers staged through the host (a debug
                                       # transport: two ranks can then share one GPU, tests/test_train_step_gpu.py::test_train_entry_point_two_ranks_one_gpu)
__C.TRAIN.DP_WIRE = 'bf16'
__C.TRAIN.DP_ALGO = 'rs_ag'
__C.TRAIN.DP_SHARD_UPDATE = True
__C.TRAIN.DP_BUCKET_UPDATE = False    # unsharded data parallel: each gradient bucket is updated right behind its all-reduce (parallel.GradReducer.bucket_update); one rank: slower
# RoIAlign + layer4[0].conv1 + layer4[0].downsample as ONE launch, one workgroup per RoI (l2s_roialign_block0_fwd, bf16): bit-identical to
# the three launches it replaces and measured slower (170 vs 121 us: every workgroup streams all 5.2 MB of weights), so it is opt-in
__C.TRAIN.FUSE_ROIALIGN = False
__C.TRAIN.BBOX_NORMALIZE_TARGETS = True
__C.TRAIN.BBOX_INSIDE_WEIGHTS = (1.0, 1.0, 1.0, 1.0)
__C.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED = True
__C.TRAIN.BBOX_NORMALIZE_MEANS = (0.0, 0.0, 0.0, 0.0)
__C.TRAIN.BBOX_NORMALIZE_STDS = (0.1, 0.1, 0.2, 0.2)
__C.TRAIN.PROPOSAL_METHOD = 'gt'
__C.TRAIN.HAS_RPN = True
__C.TRAIN.RPN_POSITIVE_OVERLAP = 0.7
__C.TRAIN.RPN_NEGATIVE_OVERLAP = 0.3
__C.TRAIN.RPN_CLOBBER_POSITIVES = False
__C.TRAIN.RPN_FG_FRACTION = 0.5
__C.TRAIN.RPN_BATCHSIZE = 256
__C.TRAIN.RPN_NMS_THRESH = 0.7
__C.TRAIN.RPN_PRE_NMS_TOP_N = 12000
__C.TRAIN.RPN_POST_NMS_TOP_N = 2000
__C.TRAIN.RPN_BBOX_INSIDE_WEIGHTS = (1.0, 1.0, 1.0, 1.0)
__C.TRAIN.RPN_POSITIVE_WEIGHT = -1.0
__C.TRAIN.USE_ALL_GT = True
__C.TRAIN.FROM_FRCN = False

__C.TEST = AttrDict()
__C.TEST.SCALES = (600,)
__C.TEST.MAX_SIZE = 1000
__C.TEST.NMS = 0.3
__C.TEST.SVM = False
__C.TEST.BBOX_REG = True
__C.TEST.HAS_RPN = True
__C.TEST.PROPOSAL_METHOD = 'gt'
__C.TEST.RPN_NMS_THRESH = 0.7
__C.TEST.RPN_PRE_NMS_TOP_N = 6000
__C.TEST.RPN_POST_NMS_TOP_N = 300
__C.TEST.MODE = 'nms'
__C.TEST.RPN_TOP_N = 5000

__C.RESNET = AttrDict()
__C.RESNET.MAX_POOL = False
__C.RESNET.FIXED_BLOCKS = 1

__C.PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])
__C.RNG_SEED = 3
__C.EXP_DIR = 'res101'
__C.USE_GPU_NMS = True
__C.POOLING_MODE = 'crop'
__C.POOLING_SIZE = 7
__C.POOLING_ALIGN = False
__C.ANCHOR_SCALES = [4, 8, 16, 32]
__C.ANCHOR_RATIOS = [0.5, 1, 2]
__C.MASK_SIZE = 14

# --- additions of this implementation (not in the reference) ---
# NMS comparator: 'ge' = the reference's CPU path (nms.c:59, the parity target), 'gt' = its CUDA kernel (nms_kernel.cu:63)
__C.NMS_CMP = 'ge'
# activation storage: 'bf16' (MFMA bf16, fp32 accumulate) or 'f32' (exact-f32 MFMA verification mode)
__C.COMPUTE_DTYPE = 'bf16'


def _merge_a_into_b(a, b):
    """config.py:325-355."""
    if not isinstance(a, dict):
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old_type = type(b[k])
        if old_type is not type(v):
            if isinstance(b[k], np.ndarray):
                v = np.array(v, dtype=b[k].dtype)
            elif isinstance(b[k], AttrDict) and isinstance(v, dict):
                pass
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(b[k]), type(v), k))
        if isinstance(v, dict):
            try:
                _merge_a_into_b(v, b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename, target=None):
    """Load a yaml config file and merge it into the default options (config.py:358-364).  `target`: another cfg object
    (model/config_vgg.py binds its own)."""
    import yaml
    with open(filename, 'r') as f:
        yaml_cfg = yaml.safe_load(f)
    _merge_a_into_b(yaml_cfg, __C if target is None else target)


def cfg_from_list(cfg_list, target=None):
    """Set config keys via list, e.g. from the command line (config.py:367-387)."""
    from ast import literal_eval
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key_list = k.split('.')
        d = __C if target is None else target
        for subkey in key_list[:-1]:
            assert subkey in d
            d = d[subkey]
        subkey = key_list[-1]
        assert subkey in d
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        assert type(value) == type(d[subkey]), 'type {} does not match original type {}'.format(type(value), type(d[subkey]))
        d[subkey] = value
