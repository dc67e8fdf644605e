#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03t}; mkdir -p $O; rm -f $O/actuals.jsonl
LLMC_TEST_ACTUALS=$PWD/$O/actuals.jsonl timeout 600 python -m pytest tests/test_fp8_block_gpu.py -q -m gpu -x > $O/tests.log 2>&1; tail -2 $O/tests.log; cat $O/actuals.jsonl
timeout 300 python tools/bench_fp8_block.py --more > $O/fp8_block_rates.txt 2>&1; tail -3 $O/fp8_block_rates.txt
