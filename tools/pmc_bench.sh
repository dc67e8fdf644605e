#!/bin/bash
# HBM-side traffic of the dominant kernel (k_syrk4) inside bench.py: FETCH_SIZE and WRITE_SIZE in separate passes
# (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2 — /opt/skills/guides/MI355X_MICROARCH.md), per input width.
# Writes gpurun_out/<round>/pmc_traffic.json; copy it to profiles/rNN_pmc_traffic.json (bench.py reports it).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_bench}
mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/f -o f -- $CMD > $OUT/f.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/w -o w -- $CMD > $OUT/w.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out_dir = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out_dir + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_syrk4<' in r['Kernel_Name']:
            agg[int(r['Dispatch_Id'])][r['Counter_Name']].append(float(r['Counter_Value']))
# dispatches of one pass in launch order: per step ONE launch for the three K = 4096 inputs (one unit queue) and one for K = 14336;
# keep the LAST step's two
def per_launch(counter):
    ids = sorted(i for i, c in agg.items() if counter in c)
    vals = [sum(agg[i][counter]) for i in ids]
    return vals[-2:]
fetch, write = per_launch('FETCH_SIZE'), per_launch('WRITE_SIZE')
hit, miss = per_launch('TCC_HIT_sum'), per_launch('TCC_MISS_sum')
T = 128 * 2048
def alg(K, copies):
    return copies * (2.0 * T * K + 4.0 * K * K)          # X read once (16-bit) + H written once, per Hessian (2NK + 4K^2)
rows = []
for i, (K, copies) in enumerate([(4096, 3), (14336, 1)]):
    if i < len(fetch) and i < len(write):
        b = (2.0 * fetch[i] + write[i]) * 1024.0     # gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (guide, HBM section)
        rows.append({'K': K, 'hessians_in_launch': copies, 'FETCH_SIZE_KB': fetch[i], 'WRITE_SIZE_KB': write[i], 'bytes': b,
                     'algorithmic_bytes': alg(K, copies), 'ratio_vs_algorithmic': b / alg(K, copies),
                     'l2_hit': hit[i] / (hit[i] + miss[i]) if i < len(hit) and i < len(miss) else None})
avg = sum(r['bytes'] for r in rows) / max(1, len(rows))
res = {'k_syrk': {'launches': len(rows), 'per_launch': rows, 'hbm_bytes_per_launch': avg,
                  'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 1; '
                          'FETCH_SIZE doubled per the gfx950 correction; fabric-side requests (Infinity-Cache hits included); '
                          'average over the two k_syrk4 launches of a step (one for the three K=4096 inputs, one for K=14336), per-launch values in per_launch'}}
json.dump(res, open(out_dir + '/pmc_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
