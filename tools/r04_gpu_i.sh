#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04i
timeout 120 python tools/probes/fp8_cast_ab.py 2>&1 | grep -v "amdgpu.ids" | tail -10 | tee gpurun_out/r04i/fp8_cast_ab4.txt
timeout 600 python -m pytest tests/test_fp8_fast_gpu.py tests/test_awq_gpu.py tests/test_fp8_block_gpu.py -m gpu -x -q -k "fp8 or fast" 2>&1 | tail -5
for i in 1; do
echo "--- bench fp8 default"; timeout 200 python bench.py --workload fp8 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
echo "--- bench fp8 exact-div"; LLMC_FP8_EXACT_DIV=1 timeout 200 python bench.py --workload fp8 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
done 2>&1 | tee gpurun_out/r04i/fp8_bench_ab2.txt
