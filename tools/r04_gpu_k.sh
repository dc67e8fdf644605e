#!/bin/bash
O=gpurun_out/r04k
mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests/test_gptq_gpu.py tests/test_sgemm_gpu.py tests/test_spqr_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests_gk32.log
for i in 1 2; do
echo "--- K-step 32"; timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "--- K-step 16"; LLMC_PROBE_LIB=tools/probes/libllmc_gk16.so timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tail -2
done | tee $O/stage_gk.txt
echo "--- 70B K-step 32"; timeout 300 python tools/bench_stages.py --70b 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/stage_70b_gk32.txt
echo "--- 70B K-step 16"; LLMC_PROBE_LIB=tools/probes/libllmc_gk16.so timeout 300 python tools/bench_stages.py --70b 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/stage_70b_gk16.txt
