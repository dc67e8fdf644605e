#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c
mkdir -p $O
timeout 300 tools/probes/syrk_lab > $O/lab.txt 2>&1
cat $O/lab.txt
timeout 300 python -m pytest tests/test_hessian_gpu.py tests/test_bench_shapes_gpu.py -m gpu -x -q -k "hessian" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
