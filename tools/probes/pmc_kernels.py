"""Per-kernel summary of one or more rocprofv3 --kernel-trace --pmc output directories:
    python tools/probes/pmc_kernels.py DIR [DIR ...] -- SUBSTR [SUBSTR ...]
For every kernel whose name contains one of the substrings: the full name once, its resources from the kernel trace (LDS,
VGPR, accumulator VGPR, SGPR, workgroup, grid), and per counter the MEAN over its dispatches of the last third of the run
(sustained clocks), with derived clock (GRBM_GUI_ACTIVE / 8 / duration) and MFMA-busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8))."""
import csv
import glob
import sys
from collections import defaultdict

args = sys.argv[1:]
dirs, subs = args[:args.index('--')], args[args.index('--') + 1:]
res, durs, cnts = {}, defaultdict(list), defaultdict(lambda: defaultdict(list))
for d in dirs:
    did = {}
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if not any(s in n for s in subs):
                continue
            wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
            gr = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
            key = (n, gr)
            res[key] = dict(lds=r['LDS_Block_Size'], scratch=r['Scratch_Size'], vgpr=r['VGPR_Count'], agpr=r['Accum_VGPR_Count'],
                            sgpr=r['SGPR_Count'], wg=wg, grid_wg=gr // max(1, wg))
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            durs[key].append((int(r['Dispatch_Id']), dur))
            did[int(r['Dispatch_Id'])] = (key, dur)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(f)):
            i = int(r['Dispatch_Id'])
            if i in did:
                per[i][r['Counter_Name']] += float(r['Counter_Value'])
        for i, c in per.items():
            key, dur = did[i]
            for k, v in c.items():
                cnts[key][k].append((i, v, dur))
for key in sorted(res, key=lambda k: (k[0], k[1])):
    n, gr = key
    r = res[key]
    ds = sorted(durs[key])
    tail = ds[len(ds) * 2 // 3:] or ds
    print(f'\n{n[:400]}')
    print(f'  grid {r["grid_wg"]} wg x {r["wg"]} threads | LDS {r["lds"]} B | VGPR {r["vgpr"]} AGPR {r["agpr"]} SGPR {r["sgpr"]} scratch {r["scratch"]} | '
          f'{len(ds)} dispatches, sustained mean {sum(d for _, d in tail) / len(tail):.1f} us')
    line, derived = [], {}
    for k in sorted(cnts[key]):
        v = sorted(cnts[key][k])
        t = v[len(v) * 2 // 3:] or v
        m = sum(x for _, x, _ in t) / len(t)
        md = sum(x for _, _, x in t) / len(t)
        derived[k] = (m, md)
        line.append(f'{k}={m:.4g}')
    print('  ' + '  '.join(line))
    if 'GRBM_GUI_ACTIVE' in derived:
        g, md = derived['GRBM_GUI_ACTIVE']
        s = f'  clock_GHz={g / 8 / md / 1e3:.3f}'
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in derived:
            s += f'  mfma_busy={derived["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 1024.0 / (g / 8):.3f}'
        print(s)
