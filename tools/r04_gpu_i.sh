#!/bin/bash
cd /root/repo
timeout 200 python -m pytest tests/test_fp8_block_gpu.py tests/test_fp8_fast_gpu.py tests/test_awq_gpu.py -m gpu -x -q -k "fp8 or fast" 2>&1 | tail -3
