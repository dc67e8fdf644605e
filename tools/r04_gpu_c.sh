#!/bin/bash
# round 4, call C: why is the pipelined schedule slower un-profiled than profiled? CU-mask probe, host issue time vs total
# time, number of hardware queues.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 60 tools/probes/cu_mask_probe 2>&1 | tee $O/cu_mask_probe.txt
echo "== default (GPU_MAX_HW_QUEUES unset)"; timeout 200 python tools/probes/k3_time.py 14336 4096 2>&1 | grep -v amdgpu.ids | tee $O/k3_time_default.txt
echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 200 python tools/probes/k3_time.py 14336 4096 2>&1 | grep -v amdgpu.ids | tee $O/k3_time_q8.txt
echo "== LLMC_SIDE_CU_MASK=0 GPU_MAX_HW_QUEUES=8"; LLMC_SIDE_CU_MASK=0 GPU_MAX_HW_QUEUES=8 timeout 200 python tools/probes/k3_time.py 14336 4096 2>&1 | grep -v amdgpu.ids | tee $O/k3_time_q8_nomask.txt
echo "== K=4096, GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 200 python tools/probes/k3_time.py 4096 6144 2>&1 | grep -v amdgpu.ids | tee $O/k3_time_4096_q8.txt
