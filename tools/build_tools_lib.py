#!/usr/bin/env python
"""Build the TOOLS variant of the C-ABI library: the same sources as liblang2seg_hip.so compiled with -DL2S_TOOLS, in which the tunables of
csrc/knobs.h are variables behind l2s_tools_set(name, value) and the knock-out builds of the filter-row weight-gradient kernel exist.
Output: lang2seg_amd/lib/liblang2seg_hip_tools.so (git-ignored).  Only tools/ab.py and the tools/*bench*.py / *stamps.py scripts load it;
bench.py, the tests and the package's default path load the product library, which exports no setter."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import CSRC, LIBDIR, HIPCC, HIPCC_FLAGS   # noqa: E402

OUT = os.path.join(LIBDIR, 'liblang2seg_hip_tools.so')


def build():
    objdir = os.path.join(LIBDIR, 'obj_tools')
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(ROOT, 'include', 'lang2seg_hip.h')]

    def cc(f):
        src, obj = os.path.join(CSRC, f), os.path.join(objdir, f[:-4] + '.o')
        if not os.path.exists(obj) or any(os.path.getmtime(s) > os.path.getmtime(obj) for s in [src] + hdrs):
            r = subprocess.run([HIPCC] + HIPCC_FLAGS + ['-DL2S_TOOLS', '-c', src, '-o', obj], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed for %s:\n%s' % (f, r.stderr[-4000:]))
        return obj
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(cc, srcs))
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    return OUT


if __name__ == '__main__':
    print(build())
