#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -25 $O/pytest.log
for bs in 1 16 128; do timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --calib-bs $bs > $O/bench_bs$bs.json 2>$O/bench_bs$bs.err; python -c "
import json; j=json.loads(open('$O/bench_bs$bs.json').read()); print('calib_bs $bs', round(j['value'],2), 'layers/s', round(j['ms_per_step'],2), 'ms', j['roofline']['frac'], j['roofline']['launches'], j['roofline']['avg_fixup_ms'])"; done
