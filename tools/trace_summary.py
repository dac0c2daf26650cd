#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-kernel totals per step and per-queue busy time.

usage: trace_summary.py <kernel_trace.csv> <steps_in_trace_tail> [out.csv]
The last `steps` train steps of the trace are isolated by looking for a kernel with exactly one launch per step: the gather of the kept
proposals in the middle of the step (the SGD kernel, used until round 3, has several launches per step since the early partial updates
are on; the NMS scan, used until round 5, has one per 4096-box stage now)."""
import csv, sys, collections, re

def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'void ', '', n)
    return n[:90]

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sgd = [i for i, r in enumerate(rows) if 'gather_rois_kernel' in r['Kernel_Name']]
if len(sgd) <= steps:
    sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
assert len(sgd) > steps, 'not enough steps in trace'
lo, hi = sgd[-steps - 1] + 1, sgd[-1] + 1
win = rows[lo:hi]
t0 = int(win[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in win)
wall = (t1 - t0) / steps / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
queues = collections.defaultdict(float)
for r in win:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = short(r['Kernel_Name'])
    agg[k][0] += 1; agg[k][1] += d
    queues[r['Queue_Id']] += d
tot = sum(v[1] for v in agg.values())
print('steps %d  wall/step %.3f ms  sum(kernel)/step %.3f ms  launches/step %d' % (steps, wall, tot / steps / 1e3, len(win) // steps))
for q, d in sorted(queues.items(), key=lambda x: -x[1]):
    print('  queue %-6s busy %.3f ms/step' % (q, d / steps / 1e3))
out = [('kernel', 'launches_per_step', 'total_us_per_step', 'avg_us', 'pct')]
for k, (n, d) in sorted(agg.items(), key=lambda x: -x[1][1]):
    out.append((k, '%.1f' % (n / steps), '%.1f' % (d / steps), '%.2f' % (d / n), '%.1f' % (100 * d / tot)))
for r in out[:32]:
    print('%-92s %8s %10s %8s %6s' % r)
if len(sys.argv) > 3:
    csv.writer(open(sys.argv[3], 'w')).writerows(out)
