#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_sgemm
mkdir -p $OUT
CMD="python tools/bench_sgemm.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
rows = defaultdict(dict)
for f in glob.glob('gpurun_out/pmc_sgemm/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_sgemm' not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'][:40], r['Grid_Size'], r['Dispatch_Id'])
        rows[key][r['Counter_Name']] = float(r['Counter_Value'])
# group by (kernel, grid): average
agg = defaultdict(lambda: defaultdict(list))
for (k, g, d), cs in rows.items():
    for c, v in cs.items(): agg[(k, g)][c].append(v)
for (k, g), cs in sorted(agg.items()):
    m = {c: sum(v)/len(v) for c, v in cs.items()}
    busy = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0); gui = m.get('GRBM_GUI_ACTIVE', 0)
    print(k, 'grid', g, {c: f'{v:.3g}' for c, v in m.items()})
PY
