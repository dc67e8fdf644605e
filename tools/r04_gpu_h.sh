#!/bin/bash
mkdir -p gpurun_out/r04h
cd /root/repo
export LLMC_TEST_ACTUALS=gpurun_out/r04h/actuals_more.jsonl
timeout 1500 python -m pytest tests/test_ref_pipeline_gpu.py -m gpu -x -q -k more_shipped 2>&1 | tail -40 > gpurun_out/r04h/tests_more.log
cat gpurun_out/r04h/tests_more.log
