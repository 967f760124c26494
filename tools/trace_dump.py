"""Every dispatch of the last training step in a rocprofv3 kernel trace CSV, in launch order: python tools/trace_dump.py trace.csv
(index, start offset us, duration us, gap before us, grid / workgroup, kernel name with its template arguments)"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_apply' in r['Kernel_Name']]
last = rows[adam[-2] + 1:adam[-1] + 1] if len(adam) >= 2 else rows
while last and 'pack_weights' in last[0]['Kernel_Name']:
    last.pop(0)
t0 = int(last[0]['Start_Timestamp'])
prev_end = t0
for i, r in enumerate(last):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name']
    nm = re.sub(r'\(anonymous namespace\)::', '', nm)
    nm = re.sub(r'\(.*$', '', nm)
    nm = re.sub(r'^void ', '', nm)
    grid = r.get('Grid_Size_X', r.get('Grid_Size', '?'))
    wg = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))
    print(f"{i:3d} {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:5.1f}  grid {grid:>8s}/{wg:<5s} {nm[:150]}")
    prev_end = e
