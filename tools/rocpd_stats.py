"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a per-kernel stats table (text)."""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else cols[0]
    rows = c.execute(f'select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), '
                     f'max(end - start) from kernels group by {name_col} order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    print(f'{"kernel":70s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}')
    for n, cnt, tot, avg, mn, mx in rows[:top]:
        print(f'{n[:70]:70s} {cnt:7d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}')
    print(f'total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
