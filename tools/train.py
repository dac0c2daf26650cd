#!/usr/bin/env python
"""Entry point of `experiments/scripts/train_baseline.sh` (reference: tools/train.py): the 'baseline' network variant (lang2seg_amd/nets/variants.py)."""
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.abspath(__file__)))
from opt import parse_opt
from train_common import main

if __name__ == '__main__':
    main(parse_opt(), variant='baseline')
