#!/bin/bash
# round 3, call B: the parity envelope again (host arms at 16 / 32 threads: 256 threads never finish the column loop),
# A/B of the Hessian kernel against the round-2 build on the same box, clip agreement, new GPU tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 300 python -m pytest tests/test_awq_gpu.py tests/test_fp8_block_gpu.py tests/test_e2e_gpu.py -q -m gpu > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 200 python tools/probes/ab_syrk_libs.py tools/probes/libllmc_hip_r02.so llmc_amd/csrc/libllmc_hip.so > $O/ab_syrk.txt 2>&1; cat $O/ab_syrk.txt
timeout 1100 python tools/parity_envelope.py --out $O/envelope > $O/envelope.log 2>&1; tail -70 $O/envelope.log
