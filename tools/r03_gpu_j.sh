#!/bin/bash
# call J: AWQ tolerance / flat-minimum tests with measured values, clip v2, FP8 block-wise rates
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O; rm -f $O/actuals.jsonl
LLMC_TEST_ACTUALS=$PWD/$O/actuals.jsonl timeout 600 python -m pytest tests/test_awq_gpu.py tests/test_clip_v2.py -q -m gpu > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python tools/bench_fp8_block.py > $O/fp8_block_rates.txt 2>&1; cat $O/fp8_block_rates.txt | tail -5
