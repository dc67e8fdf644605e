#!/bin/bash
# round 4, call D: left-looking far panel in K3, FloatQuantizer qtorch semantics + e5m2, reference pipeline test (new bounds),
# K4 group size A/B; full GPU suite, stage times, bench.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
export LLMC_TEST_ACTUALS=$GRAFT_REPO_ROOT/$O/actuals.jsonl
rm -f $LLMC_TEST_ACTUALS
( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1
tail -40 $O/tests.log
echo "== stage times (default)"; timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tee $O/stages.txt
echo "== stage times, LLMC_K4_GRP=8"; LLMC_K4_GRP=8 timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tee $O/stages_grp8.txt
echo "== stage times, helpers off"; LLMC_NO_SIDE_STREAM=1 timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tee $O/stages_serial.txt
for v in default grp8; do
  if [ $v = grp8 ]; then export LLMC_K4_GRP=8; else unset LLMC_K4_GRP; fi
  timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
    print('$v: %.2f layers/s  %.2f ms/step  k_syrk4 %.3f of peak' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
except Exception as e:
    print('$v: failed', e, open('$O/bench_$v.err').read()[-600:])
PY
done
unset LLMC_K4_GRP
