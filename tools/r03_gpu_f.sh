#!/bin/bash
# round 3, call F: fp16 roundings through the materialised fp32 value (no v_fma_mixlo_f16 fusion): the whole GPU suite,
# the clip error table against the oracle, clip / AWQ / FP8 / RTN rates
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout 100 python tools/probes/clip_agree.py > $O/clip_agree.txt 2>&1; cat $O/clip_agree.txt
timeout 300 python tools/bench_awq.py > $O/awq_fp8_rtn_rates.txt 2>&1; cat $O/awq_fp8_rtn_rates.txt
