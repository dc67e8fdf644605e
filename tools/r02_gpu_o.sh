#!/bin/bash
mkdir -p gpurun_out/o
timeout 900 python -m pytest tests/test_gptq_gpu.py -x -q -m gpu -k "chol or factor" > gpurun_out/o/tests.log 2>&1
tail -3 gpurun_out/o/tests.log
timeout 300 python tools/bench_stages.py 2>&1 | grep "down\|gate"
LLMC_K3_NO_PLANES=1 timeout 300 python tools/bench_stages.py 2>&1 | grep "down\|gate"
run() {  # name, args...
  name=$1; shift
  timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/o/$name.json 2> gpurun_out/o/$name.err
  python - "$name" <<'PY'
import json, sys
f = sys.argv[1]
try:
    j = json.load(open(f'gpurun_out/o/{f}.json')); print(f, round(j['value'], 2), round(j['ms_per_step'], 2), round(j['roofline']['frac'], 3))
except Exception as e: print(f, 'fail', e)
PY
}
run bench_planes
LLMC_K3_NO_PLANES=1 run bench_noplanes
