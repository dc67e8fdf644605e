#!/bin/bash
# round 5, call K: the reference-pipeline tests with the no-fallback assertion, e2e tests after the scatter inverse permutation, default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_ref_pipeline_gpu.py tests/test_e2e_gpu.py tests/test_gptq_gpu.py tests/test_envelope_gpu.py -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1
tail -12 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench.json 2> $O/bench.err; python - <<PY
import json
try:
    j=json.loads(open('$O/bench.json').read().strip().splitlines()[0]); print('bench', j['value'], j['ms_per_step'], j['ms_per_step_median'], j['roofline']['frac'], j.get('cpu_baseline',{}).get('value'))
except Exception as e: print('bench failed', e); print(open('$O/bench.err').read()[-1500:])
PY
