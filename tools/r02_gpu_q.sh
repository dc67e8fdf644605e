#!/bin/bash
mkdir -p gpurun_out/q
for ov in 4 3 2 6; do
  timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --overlap $ov > gpurun_out/q/ov$ov.json 2>/dev/null
  python -c "
import json; j=json.load(open('gpurun_out/q/ov$ov.json')); print('overlap $ov', round(j['value'],2), round(j['ms_per_step'],2))"
done
