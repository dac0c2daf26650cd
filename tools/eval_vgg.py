#!/usr/bin/env python
"""Evaluation entry point of the VGG16 / Faster R-CNN network (reference: tools/eval_vgg.py): box accuracy only."""
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.abspath(__file__)))
from eval_common import main, parse_args

if __name__ == '__main__':
    main(parse_args(), variant='vgg')
