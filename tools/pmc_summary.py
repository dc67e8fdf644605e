"""Average rocprofv3 --pmc counters per kernel from *_counter_collection.csv files."""
import csv
import glob
import sys
from collections import defaultdict


def main(root, kernel_filter):
    agg = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(root + '/**/*_counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r.get('Kernel_Name', '')
            if kernel_filter not in k:
                continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        print(k[:100])
        for c, v in sorted(cs.items()):
            print(f'  {c:32s} n={len(v):3d} avg={sum(v)/len(v):.6g}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
