#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i
mkdir -p $O
timeout 600 python -m pytest tests/test_export_gpu.py tests/test_awq_gpu.py tests/test_quant_gpu.py tests/test_e2e_gpu.py -m gpu -q > $O/pytest.log 2>&1
tail -25 $O/pytest.log
