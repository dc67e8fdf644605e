#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04i
timeout 600 python -m pytest tests/test_awq_gpu.py -m gpu -x -q -k "linear or ktiled or fake_quant_linear or forward" 2>&1 | tail -3
PYTHONPATH=/root/repo timeout 300 python tools/bench_linear.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04i/linear_paths2.txt
timeout 400 python bench.py --workload awq --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('awq', j['value'], j['ms_per_step'], j['roofline']['frac'])" | tee gpurun_out/r04i/awq_bench2.txt
