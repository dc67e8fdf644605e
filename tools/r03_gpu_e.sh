#!/bin/bash
# round 3, call E: where the clip kernel's fp16 candidates differ from the oracle's; K3 variants (deep inverse levels on
# gemm6 from h = 2048, in-block update as a split-bf16 product); HF AWQ test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout 200 python tools/probes/clip_qw_probe.py > $O/clip_qw.txt 2>&1; cat $O/clip_qw.txt
timeout 200 python -m pytest tests/test_hf_models_gpu.py tests/test_gptq_gpu.py -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in base "LLMC_K3_G6_MIN_H=2048" "LLMC_K3_INBLOCK_X3=1" "LLMC_K3_G6_MIN_H=2048 LLMC_K3_INBLOCK_X3=1"; do
  echo "== $v" >> $O/k3_variants.txt
  if [ "$v" = base ]; then timeout 100 python tools/bench_stages.py 2>&1 | tail -4 >> $O/k3_variants.txt; else env $v timeout 100 python tools/bench_stages.py 2>&1 | tail -4 >> $O/k3_variants.txt; fi
done
cat $O/k3_variants.txt
LLMC_K3_G6_MIN_H=2048 LLMC_K3_INBLOCK_X3=1 timeout 200 python -m pytest tests/test_gptq_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -k "chol or factor or gptq" > $O/tests_k3.log 2>&1; tail -3 $O/tests_k3.log
