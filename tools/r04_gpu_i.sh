#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04i
timeout 300 python -m pytest tests/test_quant_gpu.py -m gpu -x -q -k "act or hist or minmax or sample" 2>&1 | tail -3
timeout 200 python tools/probes/fp8_host_overhead.py 2>&1 | grep -v amdgpu.ids | head -8 | tee gpurun_out/r04i/fp8_host_overhead2.txt
timeout 200 python bench.py --workload fp8 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])" | tee gpurun_out/r04i/fp8_bench3.txt
