#!/bin/bash
O=gpurun_out/r04k
mkdir -p $O
cd /root/repo
export LLMC_TEST_ACTUALS=$O/actuals.jsonl
timeout 900 python -m pytest tests/test_gptq_gpu.py tests/test_config3_shapes_gpu.py -m gpu -x -q -k "chol or factor or split_bf16 or K28672 or 28672" 2>&1 | tail -8 | tee $O/tests.log
echo "--- gemm6 far update (default)"; timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/stage_down_g6far.txt
echo "--- k_gemm3 far update"; LLMC_K3_NO_GEMM6_FAR=1 timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/stage_down_gemm3far.txt
echo "--- 70B, gemm6 far"; timeout 300 python tools/bench_stages.py --70b 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/stage_70b_g6far.txt
