#!/bin/bash
# HBM-side traffic of k_linear_eval4 inside bench.py --workload awq: FETCH_SIZE and WRITE_SIZE in separate passes (see
# tools/pmc_bench.sh). Per subset shape: the loss launches (mode 1) of the last step. Writes <out>/pmc_traffic_awq.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_awq}
mkdir -p $OUT
CMD="python bench.py --workload awq --steps 1 --warmup 0 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/f -o f -- $CMD > $OUT/f.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/w -o w -- $CMD > $OUT/w.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out_dir = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out_dir + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_linear_eval4<1, 2' in r['Kernel_Name'] or 'k_linear_eval4<1, 2>' in r['Kernel_Name']:
            agg[int(r['Dispatch_Id'])][r['Counter_Name']].append(float(r['Counter_Value']))
def per_launch(counter):
    ids = sorted(i for i, c in agg.items() if counter in c)
    return [sum(agg[i][counter]) for i in ids]
fetch, write = per_launch('FETCH_SIZE'), per_launch('WRITE_SIZE')
hit, miss = per_launch('TCC_HIT_sum'), per_launch('TCC_MISS_sum')
N = 128 * 512
shapes = [('q|k|v', 4096, 6144), ('o', 4096, 4096), ('gate|up', 4096, 28672), ('down', 14336, 4096)]   # 20 loss launches each
rows = []
for si, (name, K, R) in enumerate(shapes):
    sl = slice(si * 20, si * 20 + 20)
    f, w = fetch[sl], write[sl]
    if len(f) < 20 or len(w) < 20:
        continue
    b = (2.0 * sum(f) / 20 + sum(w) / 20) * 1024.0            # gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2
    alg = 2.0 * (N * K + R * K + N * R)                       # x/s, Wq, the reference output: each read once (16-bit)
    h, m = sum(hit[sl]), sum(miss[sl])
    rows.append({'subset': name, 'K': K, 'R': R, 'bytes_per_launch': b, 'algorithmic_bytes': alg, 'ratio': b / alg,
                 'l2_hit': h / (h + m) if h + m else None})
res = {'k_linear_eval4': {'per_shape': rows, 'hbm_bytes_per_launch': sum(r['bytes_per_launch'] for r in rows) / max(1, len(rows)),
                          'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --workload awq --steps 1; FETCH_SIZE '
                                  'doubled per the gfx950 correction; fabric-side requests (Infinity-Cache hits included); the 20 loss launches per subset'}}
json.dump(res, open(out_dir + '/pmc_traffic_awq.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/f $OUT/w
