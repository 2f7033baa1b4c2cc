#!/usr/bin/env python3
"""One minibatch step of tools/config4_bench.py out of a rocprofv3 kernel trace: start, end, duration (us) and queue of every kernel between two forward launches.
Usage: tools/c4_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fw = [i for i, r in enumerate(rows) if 'generic_forward_kernel<true>' in r['Kernel_Name']]
i0 = fw[len(fw) // 2]
i1 = next(i for i in fw if int(rows[i]['Start_Timestamp']) > int(rows[i0]['Start_Timestamp']) + 300000)
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:48]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{r.get('Queue_Id', '?')} {n}")
