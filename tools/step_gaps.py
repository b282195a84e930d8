"""Idle gaps of the GPU (no kernel of any queue running) inside one training step of a rocprofv3 kernel trace, and the kernels on
either side.  usage: python tools/step_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rollout_team_kernel' in r['Kernel_Name'] and 'bwd' not in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
seg = rows[a:b + 1]
t0 = int(seg[0]['Start_Timestamp'])
end = t0
last = None
tot = 0.0
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > end and last is not None:
        gap = (s - end) / 1e3
        tot += gap
        if gap >= min_gap:
            print('%8.1f  gap %6.1f us   after %-40s before %s' % ((end - t0) / 1e3, gap, last['Kernel_Name'][:40], r['Kernel_Name'][:60]))
    if e > end:
        end, last = e, r
print('idle total %.1f us of %.1f' % (tot, (int(seg[-1]['Start_Timestamp']) - t0) / 1e3))
