#!/usr/bin/env python
"""Per-stream timeline of one train step from a rocprofv3 --kernel-trace CSV of bench.py (tape replay): busy time, span and the
largest idle gaps of every HIP stream, and which stream's kernel is the last to finish before each big gap of the main stream ends.
usage: stream_timeline.py <kernel_trace.csv>"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
lo, hi = sgd[-2] + 1, sgd[-1] + 1
win = rows[lo:hi]
t0 = int(win[0]['Start_Timestamp'])
def short(n): return re.sub(r'\(anonymous namespace\)::|void ', '', n)[:48]
S = collections.defaultdict(list)
for r in win:
    S[r['Stream_Id']].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, short(r['Kernel_Name'])))
print('step wall %.3f ms, %d launches' % ((max(e for v in S.values() for _, e, _ in v)) / 1e6, len(win)))
main = max(S, key=lambda k: len(S[k]))
for sid, v in sorted(S.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in v) / 1e6
    print('stream %-4s %s launches %3d  busy %.3f ms  span %.3f .. %.3f ms  first: %s  last: %s' % (sid, '(main)' if sid == main else '      ', len(v), busy, v[0][0] / 1e6, v[-1][1] / 1e6, v[0][2], v[-1][2]))
print('largest idle gaps of the main stream:')
mv = S[main]
gaps = sorted(((b[0] - a[1], a, b) for a, b in zip(mv, mv[1:])), key=lambda g: -g[0])[:8]
for g, a, b in sorted(gaps, key=lambda g: g[1][1]):
    # what ran on the other streams inside the gap
    other = collections.defaultdict(float)
    for sid, v in S.items():
        if sid == main: continue
        for s, e, n in v:
            ov = min(e, b[0]) - max(s, a[1])
            if ov > 0: other[sid] += ov
    print('  %.1f us idle at %.3f ms  after %-40s before %-40s  other streams busy in it: %s' % (g / 1e3, a[1] / 1e6, a[2], b[2], {k: round(v / 1e3) for k, v in other.items()}))
tot_gap = sum(b[0] - a[1] for a, b in zip(mv, mv[1:]))
small = sum(b[0] - a[1] for a, b in zip(mv, mv[1:]) if b[0] - a[1] < 20000)
print('main stream: busy %.3f ms, gaps %.3f ms (of which < 20 us each: %.3f ms over %d launches)' % (sum(e - s for s, e, _ in mv) / 1e6, tot_gap / 1e6, small / 1e6, len(mv)))
