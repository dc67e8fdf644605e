#!/bin/bash
# round 4, call B: new k_potrf_inv (VALU panels, unserialised loads), pipelined K3 / K4 schedules: full GPU suite,
# stage times with the pipeline on / off, bench with helper streams none / wide / all, K3 timeline.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
export LLMC_TEST_ACTUALS=$GRAFT_REPO_ROOT/$O/actuals.jsonl
rm -f $LLMC_TEST_ACTUALS
timeout 120 tools/probes/probe_potrf > $O/potrf_stamps.txt 2>&1
head -16 $O/potrf_stamps.txt; tail -4 $O/potrf_stamps.txt
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_config3_shapes_gpu.py ) > $O/tests.log 2>&1
tail -25 $O/tests.log
echo "== stage times, pipelined (default)"; timeout 200 python tools/bench_stages.py 2>&1 | tee $O/stages_pipe.txt
echo "== stage times, single stream (helpers off)"; LLMC_NO_SIDE_STREAM=1 timeout 200 python tools/bench_stages.py 2>&1 | tee $O/stages_serial.txt
echo "== stage times, pipelined without CU masks"; LLMC_SIDE_CU_MASK=0 timeout 200 python tools/bench_stages.py 2>&1 | tee $O/stages_nomask.txt
for h in none wide all; do
  timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --helpers $h > $O/bench_$h.json 2> $O/bench_$h.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$h.json').read().strip().splitlines()[-1])
    print('helpers=$h: %.2f layers/s  %.2f ms/step  k_syrk4 %.3f of peak' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
except Exception as e:
    print('helpers=$h: failed', e, open('$O/bench_$h.err').read()[-600:])
PY
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/probes/k3k4_trace.py 14336x4096 > $O/trace.log 2>&1
F=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/probes/trace_window.py $F k_potrf_inv 112 90 > $O/k3_window_early.txt 2>&1
python tools/probes/trace_window.py $F k_potrf_inv 190 90 > $O/k3_window_late.txt 2>&1
python tools/probes/trace_window.py $F k_gptq_block 140 60 > $O/k4_window.txt 2>&1
python - "$F" > $O/k3k4_span.txt 2>&1 <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# second iteration: from the 113th k_potrf_inv to the last kernel before the 3rd antitranspose... report spans per kernel family
pot = [i for i, r in enumerate(rows) if 'k_potrf_inv' in r['Kernel_Name']]
anti = [i for i, r in enumerate(rows) if 'k_antitranspose' in r['Kernel_Name']]
print('antitranspose launches', len(anti), 'potrf launches', len(pot))
if len(anti) >= 4:
    a0, a1 = anti[2], anti[3]
    t0, t1 = int(rows[a0]['Start_Timestamp']), int(rows[a1]['End_Timestamp'])
    print('K3 second run: %.2f ms from first antitranspose start to last antitranspose end' % ((t1 - t0) / 1e6))
    last_pot = [i for i in pot if a0 < i < a1][-1]
    print('   chain (first potrf .. last potrf end): %.2f ms' % ((int(rows[last_pot]['End_Timestamp']) - int(rows[[i for i in pot if i > a0][0]]['Start_Timestamp'])) / 1e6))
    print('   tail after last potrf: %.2f ms' % ((t1 - int(rows[last_pot]['End_Timestamp'])) / 1e6))
    blk = [i for i, r in enumerate(rows) if 'k_gptq_block' in r['Kernel_Name'] and i > a1]
    if blk:
        nb = len(blk) // 1
        print('K4 after it: %.2f ms (first k_gptq_block start .. last kernel end before next antitranspose)' % ((int(rows[blk[len(blk)-1]]['End_Timestamp']) - int(rows[blk[0]]['Start_Timestamp'])) / 1e6))
PY
cat $O/k3k4_span.txt
rm -rf $O/kt
head -60 $O/k3_window_late.txt
