#!/bin/bash
mkdir -p gpurun_out/o
LLMC_NO_SIDE_STREAM=1 timeout 300 python tools/bench_stages.py > gpurun_out/o/stages_s3_noside.txt 2>&1; grep down gpurun_out/o/stages_s3_noside.txt
LLMC_NO_SIDE_STREAM=1 LLMC_K3_NO_SYRK3=1 timeout 300 python tools/bench_stages.py > gpurun_out/o/stages_gemm3_noside.txt 2>&1; grep down gpurun_out/o/stages_gemm3_noside.txt
run() {  # name, args...
  name=$1; shift
  timeout 120 python bench.py --steps 3 --warmup 1 "$@" > gpurun_out/o/$name.json 2> gpurun_out/o/$name.err
  python - "$name" <<'PY'
import json, sys
f = sys.argv[1]
try:
    j = json.load(open(f'gpurun_out/o/{f}.json')); print(f, round(j['value'], 2), round(j['ms_per_step'], 2), round(j['roofline']['frac'], 3))
except Exception as e: print(f, 'fail', e)
PY
}
run bench_s3
LLMC_K3_NO_SYRK3=1 run bench_gemm3
