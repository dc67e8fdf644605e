#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 600 python -m pytest tests/test_fp8_block_gpu.py -q -m gpu -x > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python tools/bench_fp8_block.py --more > $O/fp8_block_rates.txt 2>&1; tail -3 $O/fp8_block_rates.txt
