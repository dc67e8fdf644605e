#!/bin/bash
# round-2 evidence run: bench line, kernel stats, stage times, PMC traffic, calib-bs sweep, 70B shapes, vllm variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final
mkdir -p $O
timeout 400 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
python tools/kernel_stats_csv.py $O/kt/kt_kernel_trace.csv 32 > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
timeout 200 python tools/bench_stages.py > $O/stage_times.txt 2>&1
bash tools/pmc_bench.sh $O/pmc > $O/pmc.log 2>&1
rm -rf $O/pmc/f $O/pmc/w
for bs in 1 16; do timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --calib-bs $bs > $O/bench_calib_bs$bs.json 2>/dev/null; done
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant vllm > $O/bench_vllm_variant.json 2>/dev/null
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --model llama3-70b > $O/bench_llama3_70b_shapes.json 2>/dev/null
timeout 300 python bench.py --workload awq --steps 2 --warmup 1 > $O/bench_awq.json 2>/dev/null
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --overlap 0 > $O/bench_no_overlap.json 2>/dev/null
head -c 400 $O/bench.json; echo; cat $O/kernel_stats.txt | head -12; cat $O/stage_times.txt | tail -4; tail -30 $O/pmc.log | head -40
