#!/bin/bash
# round 5, call Q: the far updates on planes (k_split3_planes + k_gemm3s) — full GPU suite, stage times, bench A/B, probe, PMC
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/tests.log 2>&1
tail -6 $O/tests.log
python tools/probes/gemm3s_probe.py > $O/gemm3s_probe.txt 2>&1; head -8 $O/gemm3s_probe.txt
LLMC_K3_NO_PLANES=1 timeout 300 python tools/bench_stages.py > $O/stage_times_noplanes.txt 2>&1; cat $O/stage_times_noplanes.txt
timeout 300 python tools/bench_stages.py > $O/stage_times_planes.txt 2>&1; cat $O/stage_times_planes.txt
run_bench() {  # name, extra args
  n=$1; shift
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1])
    print('bench $n: %.2f layers/s  %.2f ms/step (median %.2f)  k_syrk4 %.3f of peak' % (d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac']))
except Exception as e:
    print('bench $n failed', e); print(open('$O/bench_$n.err').read()[-800:])
PY
}
LLMC_K3_NO_PLANES=1 run_bench noplanes
run_bench planes
LLMC_K3_NO_PLANES=1 run_bench noplanes_again
run_bench planes_again
run_bench planes_chain --order chain
LLMC_K3_NO_PLANES=1 run_bench noplanes_chain --order chain
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/p1 -o p -- python tools/probes/k3_time.py 14336 4096 > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $O/p2 -o p -- python tools/probes/k3_time.py 14336 4096 > $O/p2.log 2>&1
python tools/probes/pmc_kernels.py $O/p1 $O/p2 -- k_gemm3 k_split3 > $O/gemm3_pmc_planes.txt 2>&1
rm -rf $O/p1 $O/p2
