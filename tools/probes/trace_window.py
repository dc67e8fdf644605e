"""Timeline window from a rocprofv3 --kernel-trace CSV: python tools/probes/trace_window.py kt_kernel_trace.csv NAME_SUBSTR skip count
Prints start offset, duration, gap to the previous kernel end on the same queue (stream), queue id, grid."""
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(.*', '', n).replace('void ', '').replace('llmc::', '')
    return n[:40]


path, after, skip, count = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
hits = [i for i, r in enumerate(rows) if after in r['Kernel_Name']]
i0 = hits[skip]
win = rows[i0:i0 + count]
t0 = int(win[0]['Start_Timestamp'])
last = {}
print(f'{"kernel":40s} {"q":>4s} {"start_us":>9s} {"dur_us":>8s} {"gap_us":>8s} grid')
for r in win:
    s, e, q = int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    print(f'{short(r["Kernel_Name"]):40s} {q[-4:]:>4s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:8.1f} {r.get("Grid_Size", "")}/{r.get("Workgroup_Size", "")}')
