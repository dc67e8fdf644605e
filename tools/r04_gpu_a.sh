#!/bin/bash
# round 4, call A: new parity tests (70B shapes, AWQ configs[2], the reference's own pipeline), K1 diagnostics
# (row stride / sustained clocks / PMC at both widths), potrf phase stamps, bench with more hardware queues.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
export LLMC_TEST_ACTUALS=$GRAFT_REPO_ROOT/$O/actuals.jsonl
rm -f $LLMC_TEST_ACTUALS
( time timeout 1500 python -m pytest tests/test_config3_shapes_gpu.py tests/test_ref_pipeline_gpu.py tests/test_hessian_gpu.py tests/test_clip_v2.py -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1
tail -40 $O/tests.log
timeout 300 python tools/probes/k1_diag.py > $O/k1_diag.txt 2>&1
cat $O/k1_diag.txt
timeout 120 tools/probes/probe_potrf > $O/potrf_stamps.txt 2>&1
head -20 $O/potrf_stamps.txt
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --output-format csv -d $O/pmc1 -o p -- python tools/probes/k1_diag.py --pmc > $O/pmc1.log 2>&1
python tools/probes/pmc_table.py $O/pmc1 k_syrk4 > $O/k1_pmc.txt 2>&1
cat $O/k1_pmc.txt
rm -rf $O/pmc1 $O/pmc2
for q in 4 16; do
  for wh in 0 1; do
    GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --wide-helper $wh > $O/bench_q${q}_wh${wh}.json 2> $O/bench_q${q}_wh${wh}.err
    python - <<PY
import json
try:
    d = json.loads(open('$O/bench_q${q}_wh${wh}.json').read().strip().splitlines()[-1])
    print('GPU_MAX_HW_QUEUES=$q wide_helper=$wh: %.2f layers/s  %.2f ms/step  k_syrk4 %.3f of peak' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
except Exception as e:
    print('GPU_MAX_HW_QUEUES=$q wide_helper=$wh: failed', e)
PY
  done
done
