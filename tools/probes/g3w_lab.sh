#!/bin/bash
# lab builds of libllmc_hip.so that differ only in gemm3_wide.hip's G3W_DBG (1: all tiles read tile (0,0)'s panels, 2: no operand DMA after
# the prologue, 3: no C traffic) -> tools/probes/g3wv/ ; run on the GPU box:  for v in 0 1 2 3: LLMC_PROBE_LIB=... python tools/probes/g3w_lab.py
cd "$(dirname "$0")/../.."
python -c "import llmc_amd.build as b; b.build()"
mkdir -p tools/probes/g3wv
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result -Illmc_amd/csrc -Iinclude"
OBJS=$(ls llmc_amd/csrc/build/*.o | grep -v gemm3_wide.o)
for v in 1 2 3; do
  /opt/rocm/bin/hipcc $F -DG3W_DBG=$v -c llmc_amd/csrc/gemm3_wide.hip -o /tmp/g3w_$v.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/g3wv/libllmc_dbg$v.so $OBJS /tmp/g3w_$v.o && echo built $v
done
