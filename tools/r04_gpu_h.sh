#!/bin/bash
# round 4, call h: the W-A configurations through the reference's own main()
mkdir -p gpurun_out/r04h
cd /root/repo
export LLMC_TEST_ACTUALS=gpurun_out/r04h/actuals.jsonl
timeout 1500 python -m pytest tests/test_ref_pipeline_gpu.py -m gpu -x -q -k activation_quantization 2>&1 | tail -60 > gpurun_out/r04h/tests_pipe.log
cat gpurun_out/r04h/tests_pipe.log
timeout 300 python -m pytest tests/test_quant_gpu.py -m gpu -x -q -k "register_act" 2>&1 | tail -5
