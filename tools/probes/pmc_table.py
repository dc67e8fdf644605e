"""Per-dispatch table from one rocprofv3 --kernel-trace --pmc run: python tools/probes/pmc_table.py DIR KERNEL_SUBSTR
Joins *_counter_collection.csv with *_kernel_trace.csv on Dispatch_Id; prints duration, every counter, and the derived
effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs /
(GRBM_GUI_ACTIVE / 8)) when those counters are present (guide: MI355X_MICROARCH.md, DVFS give-back)."""
import csv
import glob
import sys
from collections import defaultdict

d, sub = sys.argv[1], sys.argv[2]
cnt = defaultdict(lambda: defaultdict(float))
names = {}
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name']:
            cnt[int(r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
            names[int(r['Dispatch_Id'])] = r['Kernel_Name'][:40]
dur = {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name']:
            dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for i in sorted(cnt):
    c = cnt[i]
    line = f'dispatch {i:5d} {names[i]:40s} dur_us {dur.get(i, float("nan")):10.1f}'
    for k in sorted(c):
        line += f'  {k}={c[k]:.4g}'
    if 'GRBM_GUI_ACTIVE' in c and i in dur:
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        line += f'  | clock_GHz={cyc / dur[i] / 1e3:.3f}'
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            line += f' mfma_busy={c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc:.3f}'
    print(line)
