#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04i
timeout 600 python -m pytest tests/test_awq_gpu.py tests/test_config3_shapes_gpu.py -m gpu -x -q -k "awq or search or chain or ktiled or AWQ" 2>&1 | tail -3
timeout 400 python bench.py --workload awq --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'])" | tee gpurun_out/r04i/awq_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04i/kta -o kta -- python /root/repo/bench.py --workload awq --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd /root/repo
F=$(ls gpurun_out/r04i/kta/*/*kernel_trace.csv gpurun_out/r04i/kta/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kernel_stats_csv.py $F 8 | tee gpurun_out/r04i/awq_kernel_stats.txt; rm -rf gpurun_out/r04i/kta
