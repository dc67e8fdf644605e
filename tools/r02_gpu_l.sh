#!/bin/bash
mkdir -p gpurun_out/l
LLMC_LIN_BDIR=1 timeout 300 python -m pytest tests/test_awq_gpu.py -x -q -m gpu > gpurun_out/l/tests_bdir.log 2>&1
tail -4 gpurun_out/l/tests_bdir.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --workload awq --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/l/$name.json 2> gpurun_out/l/$name.err
  python - "$name" <<'PY'
import json, sys
f = sys.argv[1]
try:
    j = json.load(open(f'gpurun_out/l/{f}.json')); print(f, round(j['value'], 2), round(j['roofline']['achieved'], 1), round(j['roofline']['whole_search_tflops'], 1))
except Exception as e: print(f, 'fail', e)
PY
}
run lds_b LLMC_LIN_BDIR=0
run bdir LLMC_LIN_BDIR=1
run lds_b_main LLMC_LIN_BDIR=0 LLMC_LIN_ABL=1
run bdir_main LLMC_LIN_BDIR=1 LLMC_LIN_ABL=1
