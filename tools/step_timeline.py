"""Whole-step per-stream timeline from a rocprofv3 --kernel-trace CSV of `bench.py --steps N`.

    python tools/step_timeline.py kt_kernel_trace.csv [--step -1] [--full]

A step starts at the first k_syrk4 launch of a block (the four Hessians come first in the default order) and ends where the
next step's first k_syrk4 starts (the last step ends with its last kernel). Printed for the chosen step:
  * per stream (HSA queue): busy time (union of its kernels), kernel count, first start / last end;
  * the phases: K1 (first k_syrk4 start .. last k_syrk_fixup end) and the chain phase (.. step end), with the per-stream busy
    time inside each and the device-level union (time with at least one kernel running) and the mean number of kernels in flight;
  * per kernel name: launches, total and mean duration, streams it ran on;
  * the critical stream's (the one that ends last) gaps > 20 us inside the chain phase: what it waited behind;
  * --full: every dispatch as `stream start_us end_us dur_us kernel grid`.
Stream ids are the trace's Queue_Id, renumbered in order of first use in the step."""
import argparse
import csv
import re
from collections import defaultdict


def short(n):
    n = re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '')).replace('void ', '').replace('llmc::', '')
    return n[:44]


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def clip(iv, a, b):
    return [(max(s, a), min(e, b)) for s, e in iv if e > a and s < b]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--step', type=int, default=-1)
    ap.add_argument('--full', action='store_true')
    ap.add_argument('--gap-us', type=float, default=20.0)
    ap.add_argument('--syrk-per-step', type=int, default=0, help='a step = this many k_syrk4 launches (orders that interleave Hessians and chains)')
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    def grid(r):
        if 'Grid_Size_X' in r:
            wg = max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
            return f"{int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // wg}wg x {wg}"
        return f"{r.get('Grid_Size', '')}/{r.get('Workgroup_Size', '')}"
    # a HIP stream is the trace's Stream_Id where the column exists (several streams can share a hardware queue)
    K = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp']),
          (r.get('Stream_Id') or r['Queue_Id']) + '/q' + str(r['Queue_Id']), grid(r)) for r in rows]
    # step boundaries: a k_syrk4 whose previous k_syrk4 is more than 3 other-kernel dispatches... simpler: groups of
    # consecutive k_syrk4 launches separated by < 60 dispatches belong to one step's K1 phase
    syrk = [i for i, k in enumerate(K) if k[0].startswith('k_syrk4')]
    starts = [syrk[0]] if syrk else []
    for p, q in zip(syrk, syrk[1:]):
        if q - p > 60:
            starts.append(q)
    if a.syrk_per_step:
        starts = syrk[0::a.syrk_per_step]
    if not starts:
        raise SystemExit('no k_syrk4 dispatch in the trace')
    si = a.step if a.step >= 0 else len(starts) + a.step
    lo = starts[si]
    hi = starts[si + 1] if si + 1 < len(starts) else len(K)
    W = K[lo:hi]
    t0 = W[0][1]
    t_end = max(e for _, _, e, _, _ in W)
    if si + 1 < len(starts):
        t_end = min(t_end, K[hi][1]) if K[hi][1] > t0 else t_end
    us = lambda t: (t - t0) / 1e3
    qids = []
    for _, _, _, q, _ in W:
        if q not in qids:
            qids.append(q)
    qn = {q: i for i, q in enumerate(qids)}
    print(f'# step {si} of {len(starts)} in the trace: {len(W)} dispatches, {us(t_end):.1f} us, {len(qids)} streams')
    k1_end = max((e for n, _, e, _, _ in W if n.startswith('k_syrk')), default=t0)
    print(f'# K1 phase (first k_syrk4 start .. last k_syrk_fixup end): 0 .. {us(k1_end):.1f} us; chain phase: {us(k1_end):.1f} .. {us(t_end):.1f} us '
          f'= {us(t_end) - us(k1_end):.1f} us')
    per_q = defaultdict(list)
    for n, s, e, q, g in W:
        per_q[q].append((s, e))
    print('\n== streams ==')
    print(f'{"stream":>6s} {"id/queue":>10s} {"kernels":>8s} {"first_us":>10s} {"last_us":>10s} {"busy_us":>10s} {"busy_K1":>9s} {"busy_chain":>10s}')
    for q in qids:
        iv = per_q[q]
        print(f'{qn[q]:6d} {q:>10s} {len(iv):8d} {us(min(s for s, _ in iv)):10.1f} {us(max(e for _, e in iv)):10.1f} {union(iv) / 1e3:10.1f} '
              f'{union(clip(iv, t0, k1_end)) / 1e3:9.1f} {union(clip(iv, k1_end, t_end)) / 1e3:10.1f}')
    allv = [(s, e) for _, s, e, _, _ in W]
    for name, lo_t, hi_t in (('K1 phase', t0, k1_end), ('chain phase', k1_end, t_end)):
        c = clip(allv, lo_t, hi_t)
        span = max(1, hi_t - lo_t)
        print(f'{name}: device busy (>= 1 kernel running) {union(c) / 1e3:.1f} us of {span / 1e3:.1f} us; '
              f'kernel time {sum(e - s for s, e in c) / 1e3:.1f} us = {sum(e - s for s, e in c) / span:.2f} kernels in flight on average')
    print('\n== kernels ==')
    agg = defaultdict(lambda: [0, 0, set()])
    for n, s, e, q, g in W:
        agg[n][0] += 1
        agg[n][1] += e - s
        agg[n][2].add(qn[q])
    print(f'{"kernel":44s} {"calls":>6s} {"total_us":>10s} {"mean_us":>9s}  streams')
    for n, (c, t, qs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{n:44s} {c:6d} {t / 1e3:10.1f} {t / c / 1e3:9.1f}  {sorted(qs)}')
    # the stream that ends last
    crit = max(qids, key=lambda q: max(e for _, e in per_q[q]))
    print(f'\n== critical stream {qn[crit]} (ends last): gaps > {a.gap_us:.0f} us in the chain phase, and what ran meanwhile ==')
    seq = [(n, s, e, g) for n, s, e, q, g in W if q == crit and e > k1_end]
    tot_gap = 0.0
    shown = 0
    for (n0, s0, e0, _), (n1, s1, e1, _) in zip(seq, seq[1:]):
        gap = (s1 - e0) / 1e3
        if gap > a.gap_us:
            tot_gap += gap
            others = defaultdict(float)
            for n, s, e, q, g in W:
                if q != crit and e > e0 and s < s1:
                    others[n] += (min(e, s1) - max(s, e0)) / 1e3
            top = ', '.join(f'{k} {v:.0f}' for k, v in sorted(others.items(), key=lambda kv: -kv[1])[:3])
            if shown < 40:
                print(f'  at {us(e0):9.1f} us: {gap:7.1f} us between {n0} and {n1}; other streams: {top}')
                shown += 1
    print(f'  total of those gaps: {tot_gap:.1f} us; critical stream busy in the chain phase: '
          f'{union(clip(per_q[crit], k1_end, t_end)) / 1e3:.1f} us of {us(t_end) - us(k1_end):.1f} us')
    if a.full:
        print('\n== dispatches ==')
        print(f'{"stream":>6s} {"start_us":>10s} {"end_us":>10s} {"dur_us":>9s}  kernel  grid')
        for n, s, e, q, g in W:
            print(f'{qn[q]:6d} {us(s):10.1f} {us(e):10.1f} {(e - s) / 1e3:9.1f}  {n}  {g}')


if __name__ == '__main__':
    main()
