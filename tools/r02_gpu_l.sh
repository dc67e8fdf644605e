#!/bin/bash
# k_linear_eval4 (k-tiled) + XCD block tile order: parity tests + AWQ bench, row-major 8-wave kernel beside it
mkdir -p gpurun_out/l
timeout 300 python -m pytest tests/test_awq_gpu.py tests/test_export_gpu.py tests/test_e2e_gpu.py -x -q -m gpu > gpurun_out/l/tests.log 2>&1
tail -5 gpurun_out/l/tests.log
timeout 200 python bench.py --workload awq --steps 3 --warmup 1 > gpurun_out/l/awq4.json 2> gpurun_out/l/awq4.err
python - <<'PY'
import json
for f in ['awq4']:
    try:
        j=json.load(open(f'gpurun_out/l/{f}.json')); print(f, j['value'], j['roofline']['achieved'], j['roofline']['whole_search_tflops'])
    except Exception as e: print(f, 'fail', e)
PY
LLMC_AWQ_KT=0 timeout 200 python bench.py --workload awq --steps 3 --warmup 1 > gpurun_out/l/awq8.json 2> gpurun_out/l/awq8.err
python - <<'PY'
import json
for f in ['awq8']:
    try:
        j=json.load(open(f'gpurun_out/l/{f}.json')); print(f, j['value'], j['roofline']['achieved'], j['roofline']['whole_search_tflops'])
    except Exception as e: print(f, 'fail', e)
PY
