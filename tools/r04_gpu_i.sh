#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04i
PYTHONPATH=/root/repo timeout 300 python tools/bench_linear.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04i/linear_paths.txt
