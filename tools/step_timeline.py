#!/usr/bin/env python3
"""Timeline of ONE replayed training step from a rocprofv3 kernel trace (r1_kernel_trace.csv): every kernel between two
consecutive k_adam_multi launches with its start / end (us from the previous Adam), duration and hardware queue -- shows which
stream is the tail of the step (main: decoder / encoder backward; side: GP parameter sums + cache backward).

    python tools/step_timeline.py gpurun_out/prof_elbo_cfg2/r1_kernel_trace.csv [--from-us 2300]
"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--from-us', type=float, default=0.0)
    a = ap.parse_args()
    rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r['Start_Timestamp']))
    adam = [i for i, r in enumerate(rows) if 'k_adam_multi' in r['Kernel_Name']]
    i0, i1 = adam[-3], adam[-2]
    t0 = int(rows[i0]['End_Timestamp'])
    seg = rows[i0 + 1:i1 + 1]
    busy = {}
    for r in seg:
        busy[r['Queue_Id']] = busy.get(r['Queue_Id'], 0) + int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print('step: %.1f us, %d kernels; busy per queue: %s' % ((int(rows[i1]['End_Timestamp']) - t0) / 1e3, len(seg),
                                                            ', '.join('q%s %.0f us' % (q, v / 1e3) for q, v in sorted(busy.items()))))
    for r in seg:
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        if s < a.from_us:
            continue
        n = r['Kernel_Name']
        n = n[n.find('gp::'):] if 'gp::' in n else n
        print('%8.1f %8.1f %7.1f  q%s  %s' % (s, e, e - s, r['Queue_Id'], n[:90]))


if __name__ == '__main__':
    main()
