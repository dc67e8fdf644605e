"""Per-kernel stats table (text) from a rocprofv3 --kernel-trace CSV (same columns as tools/rocpd_stats.py)."""
import csv
import re
import sys
from collections import defaultdict


def main(path, top=40):
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    rows = sorted(((n, len(v), sum(v), sum(v) / len(v), min(v), max(v)) for n, v in agg.items()), key=lambda r: -r[2])
    total = sum(r[2] for r in rows)
    print(f'{"kernel":70s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}')
    for n, cnt, tot, avg, mn, mx in rows[:top]:
        n = re.sub(r'\(anonymous namespace\)::', '', n)
        print(f'{n[:70]:70s} {cnt:7d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}')
    print(f'total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
