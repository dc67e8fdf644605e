#!/bin/bash
mkdir -p gpurun_out/q
for q in 4 8 2; do
  GPU_MAX_HW_QUEUES=$q timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/q/hwq$q.json 2>/dev/null
  python -c "
import json; j=json.load(open('gpurun_out/q/hwq$q.json')); print('GPU_MAX_HW_QUEUES=$q', round(j['value'],2), round(j['ms_per_step'],2))"
done
