#!/usr/bin/env python
"""rocprofv3 --kernel-trace csv -> do consecutive lstm_step_fwd_kernel launches overlap in time?  Prints, per queue, the launches whose start
precedes the previous launch's end, and which queue ids the step's kernels ran on."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print('columns:', list(rows[0].keys()))
name = [k for k in rows[0] if 'Kernel_Name' in k or k == 'Kernel_Name'][0]
q = 'Queue_Id'
lst = [r for r in rows if r[name].startswith('lstm_step_fwd')]
lst.sort(key=lambda r: int(r['Start_Timestamp']))
print('%d lstm_step_fwd launches; queues used: %s' % (len(lst), collections.Counter(r[q] for r in lst)))
bad = 0
for a, b in zip(lst, lst[1:]):
    gap = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    if gap < 0:
        bad += 1
        if bad <= 10:
            print('OVERLAP: launch starts %d ns before the previous one ends (queues %s -> %s, stream %s -> %s)' % (-gap, a[q], b[q], a.get('Stream_Id'), b.get('Stream_Id')))
print('overlapping consecutive pairs:', bad)
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(lst, lst[1:])]
import statistics
print('gap ns: min %d median %d' % (min(gaps), statistics.median(gaps)))
byq = collections.defaultdict(collections.Counter)
for r in rows:
    byq[r[q]][r[name].split('(')[0][:40]] += 1
for k, c in byq.items():
    print('queue', k, c.most_common(6))
