"""Shared body of the evaluation entry points tools/eval*.py (reference: tools/eval.py:40-127, eval_spatial.py, eval_response.py):
load the snapshot `<dataset_splitBy>/output_<postfix>/<prefix>_iter_<model_iter>.pth` into the variant's resnetv1 and run
model.test.eval_split on a split.  Without dataset files in this repository the SyntheticLoader stands in for the loader."""
import argparse
import os
import os.path as osp
import sys

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', default='refcoco'); p.add_argument('--splitBy', default='unc')
    p.add_argument('--split', default='val'); p.add_argument('--id', default='mrcn_cmr_with_st')
    p.add_argument('--output_postfix', default='cycle'); p.add_argument('--model_iter', type=int, default=0)
    p.add_argument('--num_sents', type=int, default=-1); p.add_argument('--verbose', type=int, default=1)
    p.add_argument('--cfg', dest='cfg_file', default='experiments/cfgs/res101.yml')
    p.add_argument('--set', dest='set_cfgs', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--synthetic_images', type=int, default=8); p.add_argument('--dtype', default='bf16')
    p.add_argument('--results_dir', default=None, help='where det_results.txt / mask_results.txt are appended (default: experiments/)')
    p.add_argument('--synthetic', type=int, default=0, help='1: evaluate on the SyntheticLoader when the dataset files are absent')
    p.add_argument('--allow_init_weights', type=int, default=0, help='1: evaluate the initial weights when the snapshot is missing (otherwise an error)')
    return vars(p.parse_args(argv))


def main(args, variant):
    from lang2seg_amd.model.config import cfg, cfg_from_file, cfg_from_list
    from lang2seg_amd.model.test import eval_split, summarize
    from lang2seg_amd.nets.resnet_v1 import resnetv1
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    sys.path.insert(0, osp.join(ROOT, 'tools'))
    from opt import parse_opt
    torch.cuda.set_device(0)
    T = 20 if args['dataset'] == 'refcocog' else 10
    V = 3349 if args['dataset'] == 'refcocog' else 1999
    data_json = osp.join(ROOT, 'cache/prepro', args['dataset'] + '_' + args['splitBy'], 'data.json')       # eval_cycle.py:45-48
    data_h5 = osp.join(ROOT, 'cache/prepro', args['dataset'] + '_' + args['splitBy'], 'data.h5')
    if osp.exists(data_json):
        from lang2seg_amd.loaders.cycle_loader import GtMRCNLoader
        loader = GtMRCNLoader(data_json, data_h5, image_root=osp.join(ROOT, 'pyutils/mask-faster-rcnn/data/coco/images/train2014'))
    elif args['synthetic']:
        loader = SyntheticLoader(num_images=args['synthetic_images'], sents_per_image=3, T=T, vocab_size=V)
    else:
        raise FileNotFoundError('%s not found (pass --synthetic 1 to run on the synthetic stand-in)' % data_json)
    opt = parse_opt([])
    opt.update(vocab_size=loader.vocab_size, C4_feat_dim=1024, seq_length=loader.label_length,
               dataset_splitBy=args['dataset'] + '_' + args['splitBy'])
    if args['cfg_file'] and osp.exists(osp.join(ROOT, args['cfg_file'])):
        cfg_from_file(osp.join(ROOT, args['cfg_file']))
    if args['set_cfgs']:
        cfg_from_list(args['set_cfgs'])
    cfg.COMPUTE_DTYPE = args['dtype']
    if variant == 'vgg':
        from lang2seg_amd.nets.vgg16 import vgg16
        opt['C4_feat_dim'] = 512
        net = vgg16(opt, batch_size=1)
    else:
        net = resnetv1(opt, batch_size=1, num_layers=101, variant=variant)
    net.create_architecture(81, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    ckpt = osp.join(ROOT, opt['dataset_splitBy'], 'output_{}'.format(args['output_postfix']),
                    cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_{:d}.pth'.format(args['model_iter']))
    if osp.exists(ckpt):
        net.load_state_dict(torch.load(ckpt, map_location='cpu'))
        print('loaded', ckpt)
    elif args['allow_init_weights']:
        print('no snapshot at %s: evaluating the initial weights (--allow_init_weights 1)' % ckpt)
    else:
        raise FileNotFoundError('no snapshot at %s (eval.py:66 torch.load would fail; --allow_init_weights 1 evaluates the initialisers)' % ckpt)
    split = args['split'] if args['split'] in loader.split_ix else 'val'
    if variant == 'vgg':                                     # tools/eval_vgg.py: boxes only (model/test_vgg.py)
        from lang2seg_amd.model.test_vgg import eval_split as eval_split_vgg
        acc, n = eval_split_vgg(loader, net, None, split, dict(num_sents=args['num_sents'], verbose=bool(args['verbose'])))
        print('Comprehension on %s\'s %s (%s sents): box acc %.2f%%' % (opt['dataset_splitBy'], args['split'], n, acc * 100))
        return acc, None, None
    opt['split'], opt['id'] = args['split'], args['id']
    acc, eval_seg_iou_list, seg_correct, seg_total, cum_I, cum_U, num_sent = eval_split(
        loader, net, None, split, dict(num_sents=args['num_sents'], verbose=bool(args['verbose'])))
    print('Comprehension on %s\'s %s (%s sents) is %.2f%%' % (opt['dataset_splitBy'], split, num_sent, acc * 100.))
    results_str, prec, iou = summarize(eval_seg_iou_list, seg_correct, seg_total, cum_I, cum_U)
    print('Segmentation results on [%s][%s]' % (opt['dataset_splitBy'], split))
    print(results_str)
    # tools/eval_spatial.py:95-98,121-124: the two running result logs
    res_dir = args.get('results_dir') or osp.join(ROOT, 'experiments')
    os.makedirs(res_dir, exist_ok=True)
    with open(osp.join(res_dir, 'det_results.txt'), 'a') as f:
        f.write('[%s][%s], id[%s]\'s acc is %.2f%%\n' % (opt['dataset_splitBy'], opt['split'], opt['id'], acc * 100.0))
    with open(osp.join(res_dir, 'mask_results.txt'), 'a') as f:
        f.write('[%s][%s]\'s iou is:\n%s' % (opt['dataset_splitBy'], split, results_str))
    return acc, iou, prec
