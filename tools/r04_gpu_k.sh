#!/bin/bash
O=gpurun_out/r04k
mkdir -p $O
cd /root/repo
for i in 1 2; do
echo "--- phased sgemm at 3 workgroups per CU"; LLMC_PROBE_LIB=tools/probes/libllmc_occ3.so timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "--- shipped (2 per CU)"; timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tail -2
done | tee $O/stage_occ3.txt
