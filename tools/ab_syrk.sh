#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in zeros ones randn; do
  echo "fill=$f v2:"; FILL=$f python tools/bench_syrk.py 262144x4096 2>&1 | grep "T="
  echo "fill=$f v1:"; FILL=$f LLMC_SYRK_V1=1 python tools/bench_syrk.py 262144x4096 2>&1 | grep "T="
done
