"""Command-line options of the training entry points — the live subset of the reference's
tools/opt_cycle_2.py:4-128 (same flag names and defaults), plus --synthetic for dataset-less runs."""
import argparse


def parse_opt(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--imdb_name', default='coco_minus_refer'); p.add_argument('--net_name', default='res101')
    p.add_argument('--iters', default=1250000, type=int); p.add_argument('--tag', default='notime')
    p.add_argument('--dataset', default='refcoco'); p.add_argument('--splitBy', default='unc')
    p.add_argument('--output_postfix', default='cycle'); p.add_argument('--id', default='mrcn_cmr_with_st')
    p.add_argument('--with_st', type=int, default=1)
    p.add_argument('--cfg', dest='cfg_file', default='experiments/cfgs/res101.yml')
    p.add_argument('--set', dest='set_cfgs', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--max_iters', type=int, default=800000); p.add_argument('--seed', type=int, default=24)
    # language encoder (lib/layers/lang_encoder.py)
    p.add_argument('--word_embedding_size', type=int, default=512); p.add_argument('--word_vec_size', type=int, default=512)
    p.add_argument('--word_drop_out', type=float, default=0.5); p.add_argument('--bidirectional', type=int, default=1)
    p.add_argument('--rnn_hidden_size', type=int, default=512); p.add_argument('--rnn_type', default='lstm')
    p.add_argument('--rnn_drop_out', type=float, default=0.2); p.add_argument('--rnn_num_layers', type=int, default=1)
    p.add_argument('--variable_lengths', type=int, default=1)
    # caption model (lib/caption_models/AttModel.py)
    p.add_argument('--caption_model', default='att2in2'); p.add_argument('--rnn_size', type=int, default=512)
    p.add_argument('--num_layers', type=int, default=1); p.add_argument('--input_encoding_size', type=int, default=512)
    p.add_argument('--att_hid_size', type=int, default=512); p.add_argument('--fc_feat_size', type=int, default=4096)
    p.add_argument('--att_feat_size', type=int, default=4096); p.add_argument('--drop_prob_lm', type=float, default=0.5)
    p.add_argument('--start_from', default=None); p.add_argument('--cap_loss_weight', type=float, default=1.0)
    # this implementation
    p.add_argument('--synthetic', type=int, default=0, help='1: run on the SyntheticLoader when cache/prepro/<dataset>_<splitBy>/data.json is absent (otherwise that is an error)')
    p.add_argument('--from_scratch', type=int, default=0, help='1: do not load the pretrained Mask R-CNN (a missing file is otherwise an error)')
    p.add_argument('--synthetic_images', type=int, default=64); p.add_argument('--dtype', default='bf16')
    args = p.parse_args(argv)
    return vars(args)
