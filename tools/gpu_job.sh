#!/bin/bash
# One parameterised script for every GPU-box job of a round (replaces the per-round tools/rNN_gpu_*.sh of rounds 2-5):
#   gpurun --timeout T -- 'bash tools/gpu_job.sh <out-tag> <job> [<job> ...]'
# Results go to gpurun_out/<out-tag>/ (merged back by gpurun); copy what is to be judged into profiles/.
# Jobs:
#   smoke                 __graft_entry__.smoke()
#   tests[:<pytest args>] python -m pytest tests -m gpu -q <args>  (default: everything; e.g. tests:tests/test_hessian_gpu.py)
#   bench[:<flags>]       python bench.py <flags>  (default "--steps 20 --warmup 5": the driver's command) -> bench*.json + a summary line
#   stats[:<flags>]       rocprofv3 --kernel-trace --stats over bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras <flags>
#                         -> kernel_stats*.txt (tools/kernel_stats_csv.py)
#   timeline[:<flags>]    the same trace -> step_timeline*.txt (tools/step_timeline.py: per-stream busy time of one step)
#   pmc                   HBM-side traffic of k_syrk4 (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) -> pmc_traffic.json
#   py:<script and args>  python <script and args> > py_<n>.txt   (tools/bench_*.py, tools/parity_envelope.py, tools/probes/*.py)
# LLMC_OPTIONS="key=value,..." in the environment reaches llmc_amd._ffi (A/B switches of the library).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
n=0
summ() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
    r = j.get('roofline') or {}
    print('bench value %.2f  ms/step %.2f (median %s)  frac %.3f  launches/step %.1f  packed %s  cpu %s  parity_live %s' % (
        j['value'], j['ms_per_step'], j.get('ms_per_step_median'), r.get('frac', 0), r.get('launches', 0) / max(1, j['steps']),
        j.get('value_packed'), (j.get('cpu_baseline') or {}).get('value'), json.dumps(j.get('parity_live'))[:200]))
    for k, v in (j.get('extra') or {}).items():
        print('  ', k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
except Exception as e:
    print('bench line unreadable:', e)
PY
}
for job in "$@"; do
  n=$((n + 1)); kind=${job%%:*}; arg=""; [ "$job" != "$kind" ] && arg=${job#*:}
  case $kind in
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
    tests) timeout 1500 python -m pytest ${arg:-tests} -m gpu -q 2>&1 | tail -40 > $O/tests_$n.txt; tail -6 $O/tests_$n.txt ;;
    bench) timeout 900 python bench.py ${arg:---steps 20 --warmup 5} > $O/bench_$n.json 2> $O/bench_$n.err || tail -5 $O/bench_$n.err; echo "[$arg]"; summ $O/bench_$n.json ;;
    stats|timeline)
      timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras $arg > $O/ks.log 2>&1
      F=$(ls $O/ks/*/*kernel_trace.csv $O/ks/*kernel_trace.csv 2>/dev/null | head -1)
      if [ "$kind" = stats ]; then python tools/kernel_stats_csv.py $F > $O/kernel_stats_$n.txt 2>&1; head -14 $O/kernel_stats_$n.txt
      else python tools/step_timeline.py $F --step -2 > $O/step_timeline_$n.txt 2>&1; head -30 $O/step_timeline_$n.txt; fi
      rm -rf $O/ks ;;
    pmc) bash tools/pmc_bench.sh $O/pmc > $O/pmc.log 2>&1; cp $O/pmc/pmc_traffic.json $O/ 2>/dev/null; tail -30 $O/pmc.log; rm -rf $O/pmc/f $O/pmc/w ;;
    py) timeout 1500 python $arg > $O/py_$n.txt 2>&1; echo "py rc=$? [$arg]"; tail -25 $O/py_$n.txt ;;
    *) echo "unknown job $job" ;;
  esac
done
