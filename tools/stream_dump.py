#!/usr/bin/env python
"""Kernel sequence of ONE stream inside the last complete step of a rocprofv3 --kernel-trace CSV of bench.py: start, duration, gap to the
previous launch of that stream; consecutive launches of one kernel are folded.  usage: stream_dump.py <kernel_trace.csv> <rank of the stream
by launch count: 0 = main, 1 = ...> [fold=1]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one step = from one proposal gather (exactly one launch per step, in the middle of it) to the next
sgd = [i for i, r in enumerate(rows) if 'gather_rois_kernel' in r['Kernel_Name']]
win = rows[sgd[-2] + 1:sgd[-1] + 1]
t0 = int(win[0]['Start_Timestamp'])
def short(n): return re.sub(r'\(anonymous namespace\)::|void ', '', n)[:60]
S = collections.defaultdict(list)
for r in win:
    S[r['Stream_Id']].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, short(r['Kernel_Name'])))
order = sorted(S, key=lambda k: -len(S[k]))
for i, k in enumerate(order):
    print('# stream rank %d id %s: %d launches, busy %.3f ms' % (i, k, len(S[k]), sum(e - s for s, e, _ in S[k]) / 1e6))
v = S[order[int(sys.argv[2])]]
fold = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prev_end = v[0][0]
i = 0
while i < len(v):
    j = i
    if fold:
        while j + 1 < len(v) and v[j + 1][2] == v[i][2]:
            j += 1
    busy = sum(e - s for s, e, _ in v[i:j + 1]) / 1e3
    print('%9.1f us  x%-3d busy %7.1f us  span %7.1f us  gap before %6.1f us  %s' % (v[i][0] / 1e3, j - i + 1, busy, (v[j][1] - v[i][0]) / 1e3, (v[i][0] - prev_end) / 1e3, v[i][2]))
    prev_end = v[j][1]
    i = j + 1
