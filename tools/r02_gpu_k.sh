#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k
mkdir -p $O
timeout 300 python bench.py --workload awq --steps 2 --warmup 1 > $O/bench_awq.json 2> $O/bench_awq.err; tail -2 $O/bench_awq.err; cat $O/bench_awq.json
timeout 300 python -m pytest tests/test_fp8_block_gpu.py tests/test_awq_gpu.py -m gpu -q 2>&1 | tail -3
