#!/bin/bash
mkdir -p gpurun_out/r04h
cd /root/repo
export LLMC_TEST_ACTUALS=gpurun_out/r04h/actuals_export2.jsonl
timeout 1500 python -m pytest tests/test_ref_pipeline_gpu.py -m gpu -x -q -k export_step 2>&1 | tail -30 > gpurun_out/r04h/tests_export2.log
cat gpurun_out/r04h/tests_export2.log
