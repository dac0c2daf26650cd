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
        if k.startswith('resnet.'):                  # cfg.RESNET.* of the reference (FIXED_BLOCKS)
            cfg[k[7:]] = int(g[k])
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


def materialize_ref_snapshot(dst_dir, drop_last_cin_of=None):
    """Rebuild the snapshot pair the REFERENCE's SolverWrapper.snapshot() wrote (tests/golden/make_golden.py run_snapshot; committed under
    tests/golden/ref_snapshot/ without the ~230 MB of tensor payloads): the zip's structural records are the reference's bytes, every payload
    record is regenerated from the deterministic synthetic weights and must match the size and CRC-32 the reference's file had.  Returns
    (path of the .pth, path of the .pkl sidecar, manifest).  `drop_last_cin_of`: additionally write `<dst>/partial.pth`, the same state dict
    with that 4-D tensor saved one input channel short (the `[:, :-1]` rule's input, TV:121-124), and return its path as manifest['_partial']."""
    import base64
    import json
    import shutil
    import zipfile
    import zlib
    import torch
    from oracle import weights as OW
    src = os.path.join(GOLD, 'ref_snapshot')
    man = json.load(open(os.path.join(src, 'manifest.json')))
    m = man['meta']
    opt = OW.default_opt(vocab_size=m['V'], seq_length=m['T'])
    sd = OW.make_state_dict(opt, seed=m['seed_w'], head_gain=m['head_gain'], variant=m['variant'])
    by_rec = {e['record']: e for e in man['keys']}
    os.makedirs(dst_dir, exist_ok=True)
    sfile = os.path.join(dst_dir, man['pth'])
    with zipfile.ZipFile(sfile, 'w', zipfile.ZIP_STORED) as z:
        for r in man['records']:
            name = r['name']
            if name in by_rec:
                e = by_rec[name]
                dt = np.dtype(e['dtype'].replace('torch.', ''))
                if e['source'] == 'gen':
                    raw = np.ascontiguousarray(sd[e['key']], dtype=dt).tobytes()
                elif e['source'] == 'zeros':
                    raw = np.zeros(e['shape'], dt).tobytes()
                else:
                    raw = base64.b64decode(e['bytes'])
            else:
                raw = open(os.path.join(src, 'pth.' + name.split('/', 1)[1].replace('/', '.')), 'rb').read()
            assert len(raw) == r['size'] and (zlib.crc32(raw) & 0xFFFFFFFF) == r['crc32'], ('record differs from what the reference wrote', name)
            z.writestr(zipfile.ZipInfo(name), raw)
    nfile = os.path.join(dst_dir, man['pkl'])
    shutil.copy(os.path.join(src, man['pkl']), nfile)
    if drop_last_cin_of:
        full = torch.load(sfile, map_location='cpu')
        full[drop_last_cin_of] = full[drop_last_cin_of][:, :-1].clone()
        man['_partial'] = os.path.join(dst_dir, 'partial.pth')
        torch.save(full, man['_partial'])
    return sfile, nfile, man
