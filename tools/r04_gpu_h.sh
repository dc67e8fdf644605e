#!/bin/bash
mkdir -p gpurun_out/r04h
cd /root/repo
export LLMC_TEST_ACTUALS=gpurun_out/r04h/actuals_fp8rtn.jsonl
timeout 1500 python -m pytest tests/test_ref_pipeline_gpu.py -m gpu -x -q -k fp8_rtn 2>&1 | tail -30 > gpurun_out/r04h/tests_fp8rtn.log
cat gpurun_out/r04h/tests_fp8rtn.log
