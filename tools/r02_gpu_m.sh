#!/bin/bash
# GPTQ bench: subset schedules
mkdir -p gpurun_out/m
run() {  # name, args...
  name=$1; shift
  timeout 120 python bench.py --steps 3 --warmup 1 "$@" > gpurun_out/m/$name.json 2> gpurun_out/m/$name.err
  python - "$name" <<'PY'
import json, sys
f = sys.argv[1]
try:
    j = json.load(open(f'gpurun_out/m/{f}.json')); print(f, round(j['value'], 2), round(j['ms_per_step'], 2), round(j['roofline']['frac'], 3))
except Exception as e: print(f, 'fail', e)
PY
}
run shadow32h --order shadow --reserve 32 --wide-helper 1
run shadow0h --order shadow --reserve 0 --wide-helper 1
run shadow32 --order shadow --reserve 32 --wide-helper 0
