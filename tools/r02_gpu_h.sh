#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 400 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python -c "
import json; j=json.loads(open('$O/bench.json').read()); print(round(j['value'],2), round(j['ms_per_step'],2), j['roofline']['frac'], j['roofline']['achieved_incl_fixup']); print(j.get('cpu_baseline'))"
