#!/usr/bin/env python
"""Shared body of the training entry points tools/train*.py (reference: tools/train.py, train_spatial.py, train_response.py,
train_cycle_2.py:37-111, train_cycle_response.py): build loader + the variant's resnetv1 and call train_net.
Launch with torchrun for data-parallel training (one process per GPU)."""
import os
import os.path as osp
import random
import sys

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, osp.join(ROOT, 'tools'))

import numpy as np
import torch

from opt import parse_opt


def main(args, variant='cycle'):
    from lang2seg_amd.model.config import cfg, cfg_from_file, cfg_from_list
    from lang2seg_amd.model.train_val import train_net
    from lang2seg_amd.nets.resnet_v1 import resnetv1
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); local = int(os.environ.get('LOCAL_RANK', 0))
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % max(ndev, 1))                 # (TRAIN.DP_BACKEND gloo: more ranks than devices share them)
    torch.manual_seed(args['seed']); random.seed(args['seed'])
    T = 20 if args['dataset'] == 'refcocog' else 10
    V = 3349 if args['dataset'] == 'refcocog' else 1999
    # the reference's dataset files (train_cycle_2.py:50-53) when they are present, else the synthetic stand-in of the same contract
    data_json = osp.join(ROOT, 'cache/prepro', args['dataset'] + '_' + args['splitBy'], 'data.json')
    data_h5 = osp.join(ROOT, 'cache/prepro', args['dataset'] + '_' + args['splitBy'], 'data.h5')
    if osp.exists(data_json):
        from lang2seg_amd.loaders.cycle_loader import CycleLoader, GtMRCNLoader
        cls = CycleLoader if variant in ('cycle', 'cycle_response') else GtMRCNLoader      # tools/train.py etc. use GtMRCNLoader
        loader = cls(data_json, data_h5, image_root=osp.join(ROOT, 'pyutils/mask-faster-rcnn/data/coco/images/train2014'))
        if world > 1:                                            # one (image, expression) stream per rank: rank-strided image list
            for k in loader.split_ix:
                loader.split_ix[k] = loader.split_ix[k][rank::world]
                loader.perm[k] = np.arange(len(loader.split_ix[k]))
    elif args['synthetic']:
        loader = SyntheticLoader(num_images=args['synthetic_images'], sents_per_image=3, T=T, vocab_size=V, rank=rank)
    else:
        raise FileNotFoundError('%s not found (pass --synthetic 1 to run on the synthetic stand-in)' % data_json)
    opt = dict(args)
    opt['vocab_size'] = loader.vocab_size
    opt['C4_feat_dim'] = 512 if variant == 'vgg' else 1024
    opt['use_att'] = True
    opt['seq_length'] = loader.label_length
    opt['dataset_splitBy'] = args['dataset'] + '_' + args['splitBy']
    opt['checkpoint_root'] = ROOT
    if variant in ('cycle', 'cycle_response') and opt.get('start_from') is not None:
        # train_cycle_2.py:69-76: the caption model saved under <dataset_splitBy>/<start_from> must agree with the command line
        # (the weights themselves are loaded by the network's constructor, caption_models/__init__.py:45-51)
        from lang2seg_amd.utils.caption_ckpt import check_infos
        check_infos(opt, root=ROOT)
    else:
        opt['start_from'] = None                      # the other variants have no captioner (tools/train.py etc. never read it)
    if args['cfg_file'] and osp.exists(osp.join(ROOT, args['cfg_file'])):
        cfg_from_file(osp.join(ROOT, args['cfg_file']))
    if args['set_cfgs']:
        cfg_from_list(args['set_cfgs'])
    if variant == 'vgg':
        # tools/train_vgg.py:24 merges the yaml / --set list into model/config_vgg.py's cfg, the object its solver reads
        # (train_val_vgg.py:12: WEIGHT_DECAY 5e-4, DOUBLE_BIAS True by default there), while nets/network_vgg.py:29 and the layers read
        # model/config.py - which the reference therefore leaves at its defaults for this entry point.  Here both objects receive the
        # overrides: with the shipped vgg16.yml / train_vgg.sh the values the network reads are the defaults either way, and a
        # network-side override (TRAIN.RPN_BATCHSIZE ...) is honoured instead of silently dropped.
        from lang2seg_amd.model import config_vgg
        if args['cfg_file'] and osp.exists(osp.join(ROOT, args['cfg_file'])):
            config_vgg.cfg_from_file(osp.join(ROOT, args['cfg_file']))
        if args['set_cfgs']:
            config_vgg.cfg_from_list(args['set_cfgs'])
    cfg.COMPUTE_DTYPE = args['dtype']
    if world > 1:
        torch.distributed.init_process_group(cfg.TRAIN.DP_BACKEND)   # 'nccl': lazy communicator (see bench.py)
    if variant == 'vgg':
        from lang2seg_amd.nets.vgg16 import vgg16
        net = vgg16(opt, batch_size=1)
    else:
        net = resnetv1(opt, batch_size=1, num_layers=101, variant=variant)
    net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    net.rank_seed = rank * 1000003
    if world > 1:
        from lang2seg_amd.parallel import GradReducer
        net.dp = GradReducer(net, world, wire=cfg.TRAIN.DP_WIRE, algo=cfg.TRAIN.DP_ALGO, rank=rank,
                             shard_update=True if (cfg.TRAIN.DP_SHARD_UPDATE and cfg.TRAIN.DP_ALGO == 'rs_ag') else None,   # (the optimiser binds itself to the reducer)
                             bucket_update=True if (cfg.TRAIN.DP_BUCKET_UPDATE and not (cfg.TRAIN.DP_SHARD_UPDATE and cfg.TRAIN.DP_ALGO == 'rs_ag')) else None)
    output_dir = osp.join(ROOT, opt['dataset_splitBy'], 'output_{}'.format(args['output_postfix']))
    tb_dir = osp.join(ROOT, opt['dataset_splitBy'], 'tb_{}'.format(args['output_postfix']))
    pretrained = osp.join(ROOT, 'pyutils/mask-faster-rcnn/output/%s/%s_2014_train_minus_refer_valtest+%s_2014_valminusminival/%s/%s_mask_rcnn_iter_%s.pth' % (
        args['net_name'], args['imdb_name'], args['imdb_name'], args['tag'], args['net_name'], args['iters']))
    if args['from_scratch']:
        pretrained = None                                        # explicit: train from the initialisers (train_net raises on a missing file)
    train_net(net, loader, output_dir, tb_dir, pretrained_model=pretrained, max_iters=args['max_iters'], rank=rank, world=world)

