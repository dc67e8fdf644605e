#!/bin/bash
O=gpurun_out/r04j
mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc -o p -- python tools/probes/chain_diag.py > $O/pmc.log 2>&1
for k in k_sgemm k_gemm3 k_gptq_block k_potrf_inv k_linear_eval4; do
  python tools/probes/pmc_table.py $O/pmc $k > $O/pmc_$k.txt 2>&1
  python - "$O/pmc_$k.txt" $k <<'PY'
import sys,re
rows=[l for l in open(sys.argv[1]) if 'clock_GHz' in l]
import statistics as st
def f(l,key):
    m=re.search(key+r'=([0-9.]+)',l); return float(m.group(1)) if m else float('nan')
d=[float(re.search(r'dur_us\s+([0-9.]+)',l).group(1)) for l in rows]
if rows:
    tot=sum(d)
    clk=sum(f(l,'clock_GHz')*x for l,x in zip(rows,d))/tot
    busy=sum(f(l,'mfma_busy')*x for l,x in zip(rows,d))/tot
    print(f'{sys.argv[2]:16s} dispatches {len(rows):5d} total {tot/1e3:8.2f} ms  time-weighted clock {clk:.3f} GHz  mfma_busy {busy:.3f}')
    big=sorted(zip(d,rows))[-3:]
    for x,l in big: print('   longest:', l.strip()[:60], 'dur', x, 'clk', f(l,'clock_GHz'), 'busy', f(l,'mfma_busy'))
PY
done | tee $O/chain_pmc_summary.txt
rm -rf $O/pmc
