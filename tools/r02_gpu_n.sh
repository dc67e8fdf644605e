#!/bin/bash
mkdir -p gpurun_out/n
timeout 600 python -m pytest tests/test_quant_gpu.py tests/test_hessian_gpu.py -x -q -m gpu > gpurun_out/n/tests.log 2>&1
tail -30 gpurun_out/n/tests.log
