#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for i in 1 2; do
  echo "sync:"; python tools/bench_syrk.py 262144x4096 65536x14336 2>&1 | grep "T="
  echo "nosync:"; LLMC_SYRK_NOSYNC=1 python tools/bench_syrk.py 262144x4096 65536x14336 2>&1 | grep "T="
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/ab/sync -o p -- python tools/bench_syrk.py 262144x4096 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/ab/sync "k_syrk<"
