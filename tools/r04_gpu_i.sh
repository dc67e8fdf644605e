#!/bin/bash
cd /root/repo
timeout 300 python -m pytest tests/test_awq_gpu.py tests/test_fp8_block_gpu.py tests/test_hf_models_gpu.py -m gpu -x -q -k "fused_route or fp8 or float" 2>&1 | tail -3
