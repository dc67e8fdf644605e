#!/bin/bash
# call K: fp8 MFMA probe, FP8 block-wise tests (vectorised casts, K = 64 GEMM), rates
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O; rm -f $O/actuals.jsonl
timeout 120 tools/probes/fp8_mfma_probe > $O/fp8_mfma_probe.txt 2>&1; cat $O/fp8_mfma_probe.txt
LLMC_TEST_ACTUALS=$PWD/$O/actuals.jsonl timeout 600 python -m pytest tests/test_fp8_block_gpu.py tests/test_clip_v2.py tests/test_awq_gpu.py -q -m gpu -x > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python tools/bench_fp8_block.py > $O/fp8_block_rates.txt 2>&1; tail -3 $O/fp8_block_rates.txt
