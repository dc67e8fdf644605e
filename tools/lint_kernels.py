"""Compile every csrc/*.hip to gfx950 assembly and list each kernel's VGPRs, spills, scratch and LDS bytes.
Scratch (private segment) > 0 in a hot kernel is almost always an accident: an array indexed with a runtime value
(the AWQ clip search ran 3x slower that way). No GPU needed.   python tools/lint_kernels.py [file.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llmc_amd.build import CSRC, FLAGS, HIPCC  # noqa: E402


def one(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k.s')
        cmd = [HIPCC] + FLAGS + ['-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-S', '--cuda-device-only', '-o',
                                  out, os.path.join(CSRC, src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            return src, None, r.stderr[-400:]
        txt = open(out).read()
    rows = []
    for blk in txt.split('  - .agpr_count:')[1:]:
        def f(key):
            m = re.search(r'\.' + key + r':\s+(\S+)', blk)
            return m.group(1) if m else '?'
        name = f('name')
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        rows.append((re.sub(r'\(.*', '', (dem or name).replace('(anonymous namespace)::', ''))[:64], f('vgpr_count'), f('vgpr_spill_count'), f('sgpr_spill_count'),
                     f('private_segment_fixed_size'), f('group_segment_fixed_size')))
    return src, rows, ''


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, files))
    print(f'{"kernel":64s} {"vgpr":>5s} {"vspill":>6s} {"sspill":>6s} {"scratch":>7s} {"lds":>7s}')
    bad = 0
    for src, rows, err in res:
        print(f'-- {src}')
        if rows is None:
            print('   compile failed:', err)
            bad += 1
            continue
        for n, v, vs, ss, sc, lds in rows:
            flag = '  <-- scratch' if sc not in ('0', '?') else ''
            print(f'{n:64s} {v:>5s} {vs:>6s} {ss:>6s} {sc:>7s} {lds:>7s}{flag}')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
