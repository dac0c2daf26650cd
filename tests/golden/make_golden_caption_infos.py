"""Known answers for the caption warm-start option check (train_cycle_2.py:69-76): the reference repository ships the
`infos-best.pkl` files of its four caption logs (Python-2, protocol-0 pickles).  This script reads them with the reference's own
method — `pickle.load` — in THIS container (it needs numpy + argparse only; run it with a trusted checkout) and commits the
fields the warm start compares, as JSON.  tests/test_host_cpu.py checks lang2seg_amd.utils.caption_ckpt.read_infos (the no-import
reader the product uses) against them whenever /root/reference is present."""
import json
import os
import pickle

REF = '/root/reference'
out = {}
for ds in ('refcoco_unc', 'refcoco+_unc', 'refcocog_umd'):
    for log in ('caption_log_res5_2', 'caption_log_response'):
        f = os.path.join(REF, ds, log, 'infos-best.pkl')
        if not os.path.exists(f):
            continue
        with open(f, 'rb') as fid:
            infos = pickle.load(fid, encoding='latin1')
        o = vars(infos['opt'])
        out['%s/%s' % (ds, log)] = dict(best_val_score=float(infos['best_val_score']), iter=int(infos['iter']),
                                        opt={k: o[k] for k in ('caption_model', 'rnn_type', 'rnn_size', 'num_layers', 'vocab_size', 'seq_length',
                                                               'att_feat_size', 'fc_feat_size', 'input_encoding_size', 'att_hid_size')})
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_caption_infos.json'), 'w'), indent=1, sort_keys=True)
print(sorted(out))
