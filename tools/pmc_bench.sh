#!/bin/bash
# HBM traffic of the dominant kernel (k_syrk) inside bench.py: FETCH_SIZE and WRITE_SIZE in separate passes
# (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2 — /opt/skills/guides/MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_bench
mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- $CMD > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- $CMD > $OUT/w.log 2>&1
python - <<'PY'
import csv, glob, json
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob('gpurun_out/pmc_bench/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_syrk<' in k or 'k_syrk2<' in k:
            agg['k_syrk'][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in agg.items():
    fetch = cs.get('FETCH_SIZE', [])
    write = cs.get('WRITE_SIZE', [])
    out[k] = {
        'launches': len(fetch),
        'FETCH_SIZE_KB_avg': sum(fetch) / max(1, len(fetch)),
        'WRITE_SIZE_KB_avg': sum(write) / max(1, len(write)),
        # gfx950: FETCH_SIZE reports half of the bytes of a wide coalesced streaming read -> x2 (guide, HBM section)
        'hbm_bytes_per_launch': (2.0 * sum(fetch) / max(1, len(fetch)) + sum(write) / max(1, len(write))) * 1024.0,
        'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 1; '
                'FETCH_SIZE doubled per the gfx950 correction; average over the 8 k_syrk launches (6x K=4096, 2x K=14336)',
    }
json.dump(out, open('gpurun_out/pmc_bench/traffic.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
