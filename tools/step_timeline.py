"""One training step out of a rocprofv3 --kernel-trace CSV: start offset, duration, queue and name of every kernel between two
consecutive forward-rollout launches.  usage: python tools/step_timeline.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rollout_team_kernel' in r['Kernel_Name'] and 'bwd' not in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
seg = rows[a:b]
t0 = int(seg[0]['Start_Timestamp'])
print('kernels per step', len(seg), ' span us %.1f' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
busy = 0.0
for r in seg:
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy += d
    if d >= min_us:
        print('%8.1f %7.1f  q%s  grid %s x %s  %s' % (s, d, r['Queue_Id'], r['Grid_Size_X'], r['Workgroup_Size_X'], r['Kernel_Name'][:110]))
print('sum of durations %.1f' % busy)
