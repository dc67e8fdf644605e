#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j
mkdir -p $O
timeout 600 python -m pytest tests/test_fp8_block_gpu.py -m gpu -q > $O/pytest.log 2>&1
tail -30 $O/pytest.log
