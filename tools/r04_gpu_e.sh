#!/bin/bash
# round 4, call E: quick validation after restoring the round-3 K3 schedule (new potrf kept)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gptq_gpu.py tests/test_e2e_gpu.py tests/test_spqr_gpu.py tests/test_envelope_gpu.py -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1
tail -6 $O/tests.log
echo "== stage times (default)"; timeout 200 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tee $O/stages.txt
echo "== K3/K4 times"; timeout 200 python tools/probes/k3_time.py 14336 4096 2>&1 | grep -v amdgpu.ids | tee $O/k3_time.txt
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('bench: %.2f layers/s  %.2f ms/step  k_syrk4 %.3f of peak' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
PY
