#!/bin/bash
# GPTQ bench: subset schedules
mkdir -p gpurun_out/m
run() {  # name, env/args...
  name=$1; shift
  env $ENVV timeout 300 python bench.py --steps 3 --warmup 1 "$@" > gpurun_out/m/$name.json 2> gpurun_out/m/$name.err
  python - "$name" <<'PY'
import json, sys
f = sys.argv[1]
try:
    j = json.load(open(f'gpurun_out/m/{f}.json')); print(f, round(j['value'], 2), round(j['ms_per_step'], 2), round(j['roofline']['frac'], 3))
except Exception as e: print(f, 'fail', e)
PY
}
ENVV="X=1" run chain_prio
ENVV="LLMC_BENCH_PRIO=0" run chain_noprio
ENVV="X=1" run k1first --order k1first
