#!/usr/bin/env python
"""Scan the gfx950 ISA of every kernel for a result of ds_bpermute / ds_permute / ds_swizzle that is consumed after an
`s_waitcnt lgkmcnt(n > 0)` while a YOUNGER ds_write is still allowed to be outstanding.

The compiler assumes that DS operations of one wave retire in order.  Round 2 found a kernel (cap_att_bwd_step_kernel: two interleaved
wave-shuffle reductions, lane 0 storing both results to LDS) whose second sum was sporadically wrong when other kernels shared the CU:
exactly this instruction pattern, and the error went away when the pattern did.  Plain ds_read before ds_write (every software-pipelined
GEMM) is not reported.

Round 4 added a second pattern: packed fp32 VALU ops (v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32).  The LSTM step kernel's paired fmaf chains
gave sporadically wrong low halves beside other queues' kernels (tools/vgg_corun_fuzz.py); the library is built with the target feature
`packed-fp32-ops` off (__graft_entry__.HIPCC_FLAGS) and this scan checks that none is left in any kernel.

    python tools/scan_lgkm_order.py            # compiles lang2seg_amd/csrc/*.hip to assembly under /tmp and scans them
"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(path):
    hits = {}
    fn, out = None, []
    for line in open(path):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', t)
        if m:
            fn, out = m.group(1), []
            continue
        if not t or t[0] in '.;' or t.split()[0].endswith(':'):
            continue
        op = t.split()[0]
        if re.match(r'v_pk_\w+_f32', op):
            hits[fn + ' [packed fp32 op]'] = hits.get(fn + ' [packed fp32 op]', 0) + 1
        if op.startswith(('ds_bpermute', 'ds_permute', 'ds_swizzle')):
            out.append('P')
        elif op.startswith('ds_read'):
            out.append('R')
        elif op.startswith(('ds_write', 'ds_add', 'ds_or', 'ds_and', 'ds_max', 'ds_min')):
            out.append('W')
        elif op.startswith(('s_load', 's_buffer_load')):
            out.append('S')
        elif op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                if 0 < n < len(out):
                    done, rest = out[:len(out) - n], out[len(out) - n:]
                    if 'P' in done and 'W' in rest:
                        hits[fn] = hits.get(fn, 0) + 1
                out = out[len(out) - n:] if n > 0 else []
    return hits


def main():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    bad = 0
    for src in sorted(glob.glob(os.path.join(ROOT, 'lang2seg_amd', 'csrc', '*.hip'))):
        asm = os.path.join('/tmp', 'l2s_scan_' + os.path.basename(src)[:-4] + '.s')
        sys.path.insert(0, ROOT)
        from __graft_entry__ import HIPCC_FLAGS                      # the flags the library is built with
        subprocess.run([hipcc] + [f for f in HIPCC_FLAGS if f != '-fPIC'] + ['-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only',
                        src, '-o', asm], check=True, capture_output=True)
        for fn, n in scan(asm).items():
            print('%s: %d site(s) in %s' % (os.path.basename(src), n, fn))
            bad += n
    print('sites: %d' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
