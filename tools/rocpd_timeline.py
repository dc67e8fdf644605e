"""Kernel timeline from a rocprofv3 rocpd database: for a window of dispatches print start offset, duration,
gap to the previous kernel's end on the same queue, grid and queue. Usage:
    python tools/rocpd_timeline.py DB [--after NAME_SUBSTR] [--skip N] [--count M]
The window starts at the (skip+1)-th dispatch whose kernel name contains NAME_SUBSTR."""
import argparse
import re
import sqlite3


def short(n):
    n = re.sub(r'\(.*', '', n)
    n = n.replace('void ', '').replace('llmc::', '')
    return n[:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--after', default='')
    ap.add_argument('--skip', type=int, default=0)
    ap.add_argument('--count', type=int, default=80)
    a = ap.parse_args()
    c = sqlite3.connect(a.db).cursor()
    rows = c.execute('select name, start, end, queue_id, grid_x, grid_y, grid_z, workgroup_x from kernels '
                     'order by start').fetchall()
    i0 = 0
    if a.after:
        hits = [i for i, r in enumerate(rows) if a.after in r[0]]
        i0 = hits[a.skip]
    win = rows[i0:i0 + a.count]
    t0 = win[0][1]
    last_end = {}
    print(f'{"kernel":44s} {"q":>3s} {"start_us":>10s} {"dur_us":>9s} {"gap_us":>8s}  grid')
    for n, s, e, q, gx, gy, gz, wx in win:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f'{short(n):44s} {q:3d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap:8.1f}  '
              f'{gx // max(wx, 1)}x{gy}x{gz}')
    print(f'window {(win[-1][2] - t0) / 1e3:.1f} us, kernel time {sum(r[2] - r[1] for r in win) / 1e3:.1f} us')


if __name__ == '__main__':
    main()
