#!/usr/bin/env python
"""Entry point of `experiments/scripts/train_vgg.sh` (reference: tools/train_vgg.py): the VGG16 / Faster R-CNN variant
(lang2seg_amd/nets/vgg16.py), C4_feat_dim = 512."""
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.abspath(__file__)))
from opt import parse_opt
from train_common import main

if __name__ == '__main__':
    main(parse_opt(), variant='vgg')
