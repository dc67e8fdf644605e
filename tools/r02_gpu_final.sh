#!/bin/bash
# round-2 evidence run: bench lines (GPTQ headline, AWQ), kernel stats of both, stage times, variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final
mkdir -p $O
timeout 400 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
python tools/kernel_stats_csv.py $O/kt/kt_kernel_trace.csv 32 > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
timeout 300 python bench.py --workload awq --steps 3 --warmup 1 > $O/bench_awq.json 2> $O/bench_awq.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kta -o kta -- python bench.py --workload awq --steps 2 --warmup 1 --no-cpu-baseline > $O/kta.log 2>&1
python tools/kernel_stats_csv.py $O/kta/kta_kernel_trace.csv 24 > $O/awq_kernel_stats.txt 2>&1
rm -rf $O/kta
timeout 200 python tools/bench_stages.py > $O/stage_times.txt 2>&1
for bs in 1 16; do timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --calib-bs $bs > $O/bench_calib_bs$bs.json 2>/dev/null; done
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant vllm > $O/bench_vllm_variant.json 2>/dev/null
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --model llama3-70b > $O/bench_llama3_70b_shapes.json 2>/dev/null
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --overlap 0 > $O/bench_no_overlap.json 2>/dev/null
for o in chain shadow; do timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --order $o > $O/bench_order_$o.json 2>/dev/null; done
LLMC_LIN_ABL=1 timeout 200 python bench.py --workload awq --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_awq_lab_mainloop.json 2>/dev/null
LLMC_AWQ_KT=0 timeout 200 python bench.py --workload awq --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_awq_rowmajor_8wave.json 2>/dev/null
head -c 300 $O/bench.json; echo; head -c 300 $O/bench_awq.json; echo; head -8 $O/kernel_stats.txt; head -8 $O/awq_kernel_stats.txt; tail -4 $O/stage_times.txt
