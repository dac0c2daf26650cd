#!/usr/bin/env python
"""Diagnostic: per-tensor agreement of the bf16 step's gradients with the f32 step's on the full-size fixture inputs
(norm ratio, cosine), to localise a tensor whose bf16 gradient is off.  GPU only; uses the fixtures' inputs through tests/golden_util."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from golden_util import load, setup_from_fixture, variant_of
from lang2seg_amd import selftest

tag = sys.argv[1] if len(sys.argv) > 1 else 'full'
g = load(tag)
opt, sd, blob, ocfg, samp = setup_from_fixture(g)
samp['forced_proposals'] = (g['int.proposal_rois'], g['int.proposal_scores'])
over = {k[4:]: int(g[k]) for k in g if k.startswith('cfg.')}
grads = {}
for dt in ('f32', 'bf16'):
    net = selftest.build_net(opt, over, dt, sd, variant=variant_of(g))
    net.parity = selftest.parity_from_samp(samp)
    net.forward_backward(net.upload_blob(blob, 0))
    torch.cuda.synchronize()
    P = net.P
    grads[dt] = {k: P.view(k, P.grad).clone() for k in P.trainable}
    if getattr(P, 'grad_alt', None) is not None:
        grads[dt + '_alt'] = {c.wkey: c.w_grad_alt.clone() for blk in net.layers[4] for c in (blk.c1, blk.c2, blk.c3, blk.down) if c is not None}
    del net
for k in sorted(grads['f32']):
    a, b = grads['f32'][k].double(), grads['bf16'][k].double()
    na, nb = float(a.norm()), float(b.norm())
    if na == 0:
        continue
    cos = float((a * b).sum() / (na * nb + 1e-300))
    flag = '  <<<<' if abs(nb / na - 1) > 0.1 or cos < 0.99 else ''
    if flag or 'layer4' in k:
        extra = ''
        if k in grads.get('f32_alt', {}):
            ea, eb = grads['f32_alt'][k].double(), grads['bf16_alt'][k].double()
            extra = '  alt: ratio %.4f cos %.5f (alt/total f32 %.3f)' % (float(eb.norm() / (ea.norm() + 1e-300)), float((ea * eb).sum() / (ea.norm() * eb.norm() + 1e-300)), float(ea.norm() / na))
        print('%-50s ratio %.4f cos %.5f%s%s' % (k, nb / na, cos, extra, flag))
if len(sys.argv) > 2:
    k = sys.argv[2]
    a, b = grads['f32'][k].double(), grads['bf16'][k].double()
    shp = (a.numel() // int(sys.argv[3]), int(sys.argv[3]))
    a, b = a.view(shp), b.view(shp)
    print('per-column (first 16): ratio of norms', [round(float(b[:, c].norm() / a[:, c].norm()), 4) for c in range(16)])
    print('per-column cos', [round(float((a[:, c] * b[:, c]).sum() / (a[:, c].norm() * b[:, c].norm())), 5) for c in range(16)])
    cn = (b.norm(dim=0) / a.norm(dim=0))
    print('columns with ratio off by > 5%:', torch.nonzero((cn - 1).abs() > 0.05).flatten().tolist()[:40], 'of', shp[1])
    rn = (b.norm(dim=1) / a.norm(dim=1))
    print('rows with ratio off by > 5%:', torch.nonzero((rn - 1).abs() > 0.05).flatten().tolist()[:40], 'of', shp[0])
    g2 = load(tag)
    samp_ref = g2['g.' + k + '.sample']
    print('fixture sample[:8]', samp_ref[:8])
    print('f32 col0[:8]', a[:8, 0].tolist())
    print('bf16 col0[:8]', b[:8, 0].tolist())
