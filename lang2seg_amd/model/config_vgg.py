"""The configuration object of the VGG16 solver - the reference keeps a second module, pyutils/mask-faster-rcnn/lib/model/config_vgg.py,
that tools/train_vgg.py:24 merges the yaml / command line into and that model/train_val_vgg.py:12 (the solver) reads, while the network
and its layers (nets/network_vgg.py:29, layer_utils/*.py) keep reading model/config.py.  It differs from model/config.py in four values
(config_vgg.py:28,40,100,267); everything else, and the override API, is shared."""
import copy

from . import config as _base

cfg = copy.deepcopy(_base.cfg)
cfg.TRAIN.WEIGHT_DECAY = 0.0005
cfg.TRAIN.DOUBLE_BIAS = True
cfg.TRAIN.SNAPSHOT_PREFIX = 'vgg16_faster_rcnn'
cfg.EXP_DIR = 'vgg16'


def cfg_from_file(filename):
    _base.cfg_from_file(filename, cfg)


def cfg_from_list(cfg_list):
    _base.cfg_from_list(cfg_list, cfg)
