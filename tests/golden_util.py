"""Helpers shared by the golden-fixture tests."""
import os
import copy
import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(tag):
    return dict(np.load(os.path.join(GOLD, 'ref_%s.npz' % tag), allow_pickle=False))


def check_digest(g, prefix, arr, rtol=1e-4, atol=1e-4, extra_sample=0.0, extra_sum=0.0):
    a = np.asarray(arr, dtype=np.float64).ravel()
    assert list(g[prefix + '.shape']) == list(np.asarray(arr).shape), (prefix, g[prefix + '.shape'], np.asarray(arr).shape)
    stride = int(g[prefix + '.stride'])
    samp = g[prefix + '.sample'].astype(np.float64)
    mine = a[::stride][:samp.size]
    scale = max(1.0, float(np.abs(samp).max()))
    err = np.abs(mine - samp).max()
    assert err <= atol * scale + rtol * scale + extra_sample, (prefix, 'sample max err', err, 'scale', scale)
    dsum = abs(a.sum() - float(g[prefix + '.sum']))
    assert dsum <= (rtol * float(g[prefix + '.abssum']) + atol + extra_sum), (prefix, 'sum', dsum, rtol * float(g[prefix + '.abssum']) + atol + extra_sum)
    return err


def digest_metrics(g, prefix, arr):
    """(cosine, max-normalised error, relative L2 error, 99th-percentile error over the tensor's scale) of `arr` against the fixture's
    strided sample of the same tensor; the tensor's scale = max(largest sample entry, mean |entry| of the whole tensor)"""
    a = np.asarray(arr, dtype=np.float64).ravel()
    assert list(g[prefix + '.shape']) == list(np.asarray(arr).shape), (prefix, g[prefix + '.shape'], np.asarray(arr).shape)
    stride = int(g[prefix + '.stride'])
    samp = g[prefix + '.sample'].astype(np.float64)
    mine = a[::stride][:samp.size]
    nrm = float(np.linalg.norm(samp))
    mean_abs = float(g[prefix + '.abssum']) / max(1.0, float(np.prod(g[prefix + '.shape'])))
    scale = max(float(np.abs(samp).max()), mean_abs, 1e-300)
    p99 = float(np.percentile(np.abs(mine - samp), 99)) / scale
    if nrm == 0.0:
        return 1.0, float(np.abs(mine).max()), float(np.linalg.norm(mine)), p99
    cos = float(mine @ samp / (np.linalg.norm(mine) * nrm + 1e-300))
    return cos, float(np.abs(mine - samp).max() / np.abs(samp).max()), float(np.linalg.norm(mine - samp) / nrm), p99


def variant_of(g):
    return str(g['meta_variant']) if 'meta_variant' in g else 'cycle'


def setup_from_fixture(g):
    """(opt, state_dict, blob, cfg, samp) reproducing the inputs the reference was run on."""
    from oracle import weights as OW, synth as OS, net as ON
    opt = OW.default_opt(vocab_size=int(g['meta_V']), seq_length=int(g['meta_T']))
    if OW.VARIANTS[variant_of(g)].get('backbone') == 'vgg':
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=int(g['meta_seed_w']), head_gain=float(g['meta_head_gain']), variant=variant_of(g))
    blob = OS.make_blob(int(g['meta_H']), int(g['meta_W']), int(g['meta_T']), int(g['meta_V']), seed=int(g['meta_seed_blob']))
    cfg = copy.deepcopy(ON.DEFAULT_CFG)
    for k in g:
        if k.startswith('cfg.'):
            cfg['TRAIN'][k[4:]] = int(g[k])
        if k.startswith('top.'):                     # top-level switches of the reference's cfg (POOLING_ALIGN, ...)
            cfg[k[4:]] = bool(int(g[k]))
    samp = dict(rpn_fg_keys=g['samp.rpn_fg_keys'], rpn_bg_keys=g['samp.rpn_bg_keys'],
                roi_fg_keys=g['samp.roi_fg_keys'], roi_bg_keys=g['samp.roi_bg_keys'])
    return opt, sd, blob, cfg, samp


def setup_from_fixture_test(g):
    """inputs of a TEST-mode fixture (default TEST proposal settings, no sampling keys)."""
    from oracle import weights as OW, synth as OS, net as ON
    opt = OW.default_opt(vocab_size=int(g['meta_V']), seq_length=int(g['meta_T']))
    if OW.VARIANTS[variant_of(g)].get('backbone') == 'vgg':
        opt['C4_feat_dim'] = 512
    sd = OW.make_state_dict(opt, seed=int(g['meta_seed_w']), head_gain=float(g['meta_head_gain']), variant=variant_of(g))
    blob = OS.make_blob(int(g['meta_H']), int(g['meta_W']), int(g['meta_T']), int(g['meta_V']), seed=int(g['meta_seed_blob']))
    cfg = copy.deepcopy(ON.DEFAULT_CFG)
    if 'meta_test_mode' in g:
        cfg['TEST']['MODE'] = str(g['meta_test_mode'])
        if int(g['meta_top_n']):
            cfg['TEST']['RPN_TOP_N'] = int(g['meta_top_n'])
    return opt, sd, blob, cfg, None
