#!/bin/bash
# round 5, call B: split-K small-product GEMM (tests + rates), the corrected shadow order (the WIDEST Hessian is the one
# that leaves CUs free), its timeline, the vendor NT GEMM under PMC.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
( time timeout 900 python -m pytest tests/test_awq_gpu.py tests/test_hf_models_gpu.py -m gpu -q -p no:cacheprovider -x ) > $O/tests.log 2>&1
tail -15 $O/tests.log
timeout 300 python tools/bench_linear.py > $O/linear_paths.txt 2>&1; cat $O/linear_paths.txt
run_bench() {  # name, extra args
  n=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1])
    print('bench $n: %.2f layers/s  %.2f ms/step (median %.2f, min/max %s)  k_syrk4 %.3f of peak  barrier timeouts %s' % (
        d['value'], d['ms_per_step'], d['ms_per_step_median'], ['%.1f' % v for v in d['ms_per_step_min_max']], d['roofline']['frac'], d['roofline']['round_barrier_timeouts_last_step']))
except Exception as e:
    print('bench $n failed', e); print(open('$O/bench_$n.err').read()[-800:])
PY
}
run_bench default
run_bench shadow16 --order shadow --reserve 16
run_bench shadow32 --order shadow --reserve 32
run_bench shadow64 --order shadow --reserve 64
run_bench shadow96 --order shadow --reserve 96
run_bench shadow64_wide --order shadow --reserve 64 --helpers wide
run_bench shadow32_wide --order shadow --reserve 32 --helpers wide
GPU_MAX_HW_QUEUES=8 run_bench shadow64_q8 --order shadow --reserve 64
GPU_MAX_HW_QUEUES=8 run_bench shadow64_q8_wide --order shadow --reserve 64 --helpers wide
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --order shadow --reserve 64 > $O/kt.log 2>&1
F=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/step_timeline.py $F --full --syrk-per-step 4 > $O/step_timeline_shadow64.txt 2>&1; head -24 $O/step_timeline_shadow64.txt
rm -rf $O/kt
# the vendor GEMM: timings, then counter passes
timeout 300 python tools/probes/vendor_gemm_pmc.py > $O/vendor_times.txt 2>&1; cat $O/vendor_times.txt
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --output-format csv -d $O/vp1 -o p -- python tools/probes/vendor_gemm_pmc.py --pmc > $O/vp1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/vp2 -o p -- python tools/probes/vendor_gemm_pmc.py --pmc > $O/vp2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/vp3 -o p -- python tools/probes/vendor_gemm_pmc.py --pmc > $O/vp3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCP_TCC_READ_REQ_sum --output-format csv -d $O/vp4 -o p -- python tools/probes/vendor_gemm_pmc.py --pmc > $O/vp4.log 2>&1
python tools/probes/pmc_kernels.py $O/vp1 $O/vp2 $O/vp3 $O/vp4 -- Cijk k_linear_eval4 gemm Gemm > $O/vendor_pmc.txt 2>&1; cat $O/vendor_pmc.txt
tail -3 $O/vp2.log
rm -rf $O/vp1 $O/vp2 $O/vp3 $O/vp4
