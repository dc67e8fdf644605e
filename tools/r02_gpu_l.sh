#!/bin/bash
# k_linear_eval4: parity tests + AWQ bench
mkdir -p gpurun_out/l
timeout 300 python -m pytest tests/test_awq_gpu.py tests/test_e2e_gpu.py -x -q -m gpu > gpurun_out/l/tests.log 2>&1
tail -3 gpurun_out/l/tests.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --workload awq --steps 3 --warmup 1 > gpurun_out/l/$name.json 2> gpurun_out/l/$name.err
  python - "$name" <<'PY'
import json, sys
f = sys.argv[1]
try:
    j = json.load(open(f'gpurun_out/l/{f}.json')); print(f, round(j['value'], 2), round(j['roofline']['achieved'], 1), round(j['roofline']['whole_search_tflops'], 1))
except Exception as e: print(f, 'fail', e)
PY
}
run awq4 X=1
run awq4_mainloop LLMC_LIN_ABL=1
