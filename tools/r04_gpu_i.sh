#!/bin/bash
mkdir -p gpurun_out/r04i
cd /root/repo
timeout 120 python tools/probes/fp8_cast_ab.py 2>&1 | grep -v "amdgpu.ids" | tail -10 | tee gpurun_out/r04i/fp8_cast_ab3.txt
timeout 600 python -m pytest tests/test_fp8_fast_gpu.py tests/test_awq_gpu.py tests/test_fp8_block_gpu.py -m gpu -x -q -k "fp8 or fast" 2>&1 | tail -5
