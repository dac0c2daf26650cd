"""Caption-model warm start (`--start_from caption_log_*`), the reference's
lib/caption_models/__init__.py:45-51 (directory + `infos-best.pkl` must exist, `model-best.pth` is loaded into the
captioner with its own, un-prefixed key names, strictly) and tools/train_cycle_2.py:69-76 (the options saved in
`infos-best.pkl` must agree with the command line on caption_model / rnn_type / rnn_size / num_layers).

`infos-best.pkl` is a protocol-0 pickle written by Python 2 (an `argparse.Namespace` under 'opt', numpy index
arrays, history scalars).  It is untrusted input, so it is never handed to `pickle.load`: `read_infos` runs the
pickle machine with `find_class` replaced by a factory of inert stand-in classes — no module is imported and no
foreign callable is ever invoked — and keeps only plain python values.  (`pickletools.genops` cannot be used:
it decodes Python-2 string opcodes as ASCII and stops at the first numpy byte string.)"""
import os
import pickle

NEED_BE_SAME = ('caption_model', 'rnn_type', 'rnn_size', 'num_layers')        # train_cycle_2.py:73


class _Inert(object):
    """stand-in for every global the pickle names: construction and state restoration only store their arguments"""

    def __init__(self, *a, **k):
        pass

    def __new__(cls, *a, **k):
        return object.__new__(cls)

    def __call__(self, *a, **k):          # copy_reg._reconstructor(cls, base, state) and numpy's _reconstruct land here
        for x in a:
            if isinstance(x, type) and issubclass(x, _Inert):
                return x()
        return _Inert()

    def __setstate__(self, state):
        self.__dict__['state'] = state


class _NoImportUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        cls = type(str(name), (_Inert,), {'__module__': str(module)})
        if name in ('_reconstructor', '_reconstruct', 'dtype'):
            return cls()                   # a callable instance: REDUCE calls it with the pickled arguments
        return cls


def read_infos(path):
    """-> dict of the pickled top-level dict with 'opt' turned into a plain dict of python scalars/strings"""
    with open(path, 'rb') as f:
        top = _NoImportUnpickler(f, encoding='latin1').load()
    if not isinstance(top, dict) or 'opt' not in top:
        raise ValueError('%s: not a caption infos file (no "opt")' % path)
    o = top['opt']
    st = o if isinstance(o, dict) else getattr(o, 'state', None)
    if not isinstance(st, dict):
        st = {k: v for k, v in getattr(o, '__dict__', {}).items() if k != 'state'}
    out = {k: v for k, v in top.items() if isinstance(v, (int, float, str, bool, type(None)))}
    out['opt'] = {str(k): v for k, v in st.items() if isinstance(v, (int, float, str, bool, type(None)))}
    return out


def start_dir(opt, root='.'):
    return os.path.join(root, opt['dataset_splitBy'], opt['start_from'])


def check_infos(opt, root='.'):
    """train_cycle_2.py:69-76.  Raises (the reference asserts) when the directory / file is missing or an option disagrees."""
    if opt.get('start_from') is None:
        return None
    d = start_dir(opt, root)
    if not os.path.isdir(d):
        raise FileNotFoundError('%s must be a path' % d)
    f = os.path.join(d, 'infos-best.pkl')
    if not os.path.isfile(f):
        raise FileNotFoundError('infos-best.pkl file does not exist in path %s' % d)
    infos = read_infos(f)
    for k in NEED_BE_SAME:
        if k not in infos['opt'] or infos['opt'][k] != opt[k]:
            raise ValueError("Command line argument and saved model disagree on '%s' (%r vs %r)" % (k, opt.get(k), infos['opt'].get(k)))
    return infos


def load_caption_weights(net, opt, root='.'):
    """caption_models/__init__.py:45-51: `model.load_state_dict(torch.load(<dir>/model-best.pth))` — strict: every captioner
    tensor must be present with its shape and nothing else may be.  The file's keys are the caption model's own
    (`embed.0.weight`, `core.i2h.weight`, ...); the network's state dict holds them under `caption_model.`."""
    import torch
    if opt.get('start_from') is None:
        return False
    d = start_dir(opt, root)
    if not os.path.isdir(d):
        raise FileNotFoundError('%s must be a path' % d)
    if not os.path.isfile(os.path.join(d, 'infos-best.pkl')):
        raise FileNotFoundError('infos-best.pkl file does not exist in path %s' % d)
    f = os.path.join(d, 'model-best.pth')
    try:
        sd = torch.load(f, map_location='cpu', weights_only=True)        # tensors only: no arbitrary unpickling
    except TypeError:
        sd = torch.load(f, map_location='cpu')
    cur = net.state_dict()
    own = {k[len('caption_model.'):]: k for k in cur if k.startswith('caption_model.')}
    missing = sorted(set(own) - set(sd)); unexpected = sorted(set(sd) - set(own))
    if missing or unexpected:
        raise KeyError('caption model-best.pth does not match the captioner: missing %s, unexpected %s' % (missing, unexpected))
    upd = {}
    for k, full in own.items():
        if tuple(sd[k].shape) != tuple(cur[full].shape):
            raise ValueError('caption weight %s: shape %s, expected %s' % (k, tuple(sd[k].shape), tuple(cur[full].shape)))
        upd[full] = sd[k].float()
    cur.update(upd)
    net.load_state_dict(cur)
    print('--------------------------------')
    print('Caption model restored from {}'.format(f))
    return True
