#!/bin/bash
# round 4, call g: W-A AWQ search + wide AutoClipper (new kernel) against the goldens / oracle
mkdir -p gpurun_out/r04g
cd /root/repo
LLMC_TEST_ACTUALS=gpurun_out/r04g/actuals.jsonl timeout 900 python -m pytest tests/test_clip_wide_gpu.py tests/test_clip_v2.py "tests/test_awq_gpu.py" -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r04g/tests.log
cat gpurun_out/r04g/tests.log
ls gpurun_out | head
