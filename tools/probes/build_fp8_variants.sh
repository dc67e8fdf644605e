#!/bin/bash
# lab builds of libllmc_hip.so that differ only in k_fp8_cast's unroll / layout / grid cap (tools/probes/fp8_cast_ab.py)
cd "$(dirname "$0")/../.."
python -c "import llmc_amd.build as b; b.build()"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
OBJS=$(ls llmc_amd/csrc/build/*.o | grep -v fp8_pack.o)
for v in "1 1 8192" "2 1 8192" "4 1 8192" "4 0 8192" "8 1 8192" "4 1 2048" "8 1 1024" "2 0 8192"; do
  set -- $v
  n=u$1_p$2_g$3
  /opt/rocm/bin/hipcc $F -DFP8_U=$1 -DFP8_PATTERN=$2 -DFP8_GRID_CAP=$3 -c llmc_amd/csrc/fp8_pack.hip -o /tmp/fp8_$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/fp8v/libllmc_$n.so $OBJS /tmp/fp8_$n.o && echo built $n
done
