#!/bin/bash
mkdir -p gpurun_out/r04h
cd /root/repo
export LLMC_TEST_ACTUALS=gpurun_out/r04h/actuals_spqr.jsonl
timeout 1500 python -m pytest tests/test_ref_pipeline_gpu.py -m gpu -x -q -k spqr 2>&1 | tail -40 > gpurun_out/r04h/tests_spqr.log
cat gpurun_out/r04h/tests_spqr.log
