# K3 + K4 of down_proj alone under rocprofv3, once per LLMC_OPTIONS arm given as arguments (default: the far-update kernels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG:-r06s}; mkdir -p $O
[ $# -eq 0 ] && set -- sgemm_no_wide=0 sgemm_no_wide=4 sgemm_no_wide=1
for v in "$@"; do
  LLMC_OPTIONS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$v -o ks -- python tools/probes/k3_alone.py > $O/k3_alone_$v.log 2>&1
  F=$(ls $O/ks_$v/*/*kernel_trace.csv $O/ks_$v/*kernel_trace.csv 2>/dev/null | head -1)
  python tools/kernel_stats_csv.py $F > $O/chain_alone_stats_$v.txt 2>&1
  echo "== $v"; grep "column loop" $O/k3_alone_$v.log | tail -2
  grep -i "sgemm\|gptq_block" $O/chain_alone_stats_$v.txt | cut -c1-150 | head -4
  rm -rf $O/ks_$v
done
