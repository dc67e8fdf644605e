#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04i
timeout 600 python -m pytest tests/test_fp8_block_gpu.py tests/test_fp8_fast_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/bench_fp8_block.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r04i/fp8_block_rates.txt
