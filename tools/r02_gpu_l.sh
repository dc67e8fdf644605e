#!/bin/bash
mkdir -p gpurun_out/l
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --workload awq --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/l/$name.json 2> gpurun_out/l/$name.err
  python - "$name" <<'PY'
import json, sys
f = sys.argv[1]
try:
    j = json.load(open(f'gpurun_out/l/{f}.json')); print(f, round(j['value'], 2), round(j['roofline']['achieved'], 1), round(j['roofline']['whole_search_tflops'], 1))
except Exception as e: print(f, 'fail', e)
PY
}
run mainloop LLMC_LIN_ABL=1
run mainloop_noBreads LLMC_LIN_ABL=4
run mainloop_constBreads LLMC_LIN_ABL=5
