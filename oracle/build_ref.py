"""Materialise oracle/_ref/ : the REFERENCE's own llmc package made CPU-runnable by the reference's own CI rewrite
(ci_check/change_files.py over ci_check/cpu.txt), so that bench.py's `cpu_baseline` leg can time the reference
itself (`kind: "reference"`) on the GPU box's host cores, where /root/reference does not exist.

  * run here (build container) by __graft_entry__.build(); output only under oracle/_ref/ (git-ignored, travels with
    the gpurun snapshot like a built .so); nothing of it is committed and nothing in the product imports it;
  * the rewrite is executed from the reference's own script, then its two WORK-SHRINKERS are reverted (AWQ n_grid 1 -> 20;
    eval sample caps are irrelevant here) and the same `.cuda()` / device='cuda' substitution is extended to the two
    files its cpu.txt misses (module_utils.py, quant.py: SURVEY.md §8c caveat (i)).
Test infrastructure only."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('LLMC_REFERENCE', '/root/reference')
OUT = os.path.join(HERE, '_ref')


def build(verbose=False):
    if not os.path.isdir(os.path.join(REF, 'llmc')):
        return None                                   # GPU box: use what was built in the container
    stamp = os.path.join(OUT, '.built')
    if os.path.exists(stamp):
        return OUT
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    shutil.copytree(os.path.join(REF, 'llmc'), os.path.join(OUT, 'llmc'),
                    ignore=shutil.ignore_patterns('__pycache__', '*.pyc'))
    ci = os.path.join(OUT, 'ci_check')
    os.makedirs(ci)
    for f in ('change_files.py', 'cpu.txt'):
        shutil.copy(os.path.join(REF, 'ci_check', f), ci)
    r = subprocess.run([sys.executable, 'change_files.py'], cwd=ci, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('reference ci_check/change_files.py failed:\n' + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)
    shutil.rmtree(ci)
    q = os.path.join(OUT, 'llmc', 'compression', 'quantization')
    # revert the CI's work shrinker (the search must run its 20 grid points)
    p = os.path.join(q, 'awq.py')
    s = open(p).read()
    assert 'n_grid_zbl = 1\n' in s
    open(p, 'w').write(s.replace('n_grid_zbl = 1\n', 'n_grid_zbl = 20\n', 1))
    # files the CI list misses (it never runs pack / real quant): same substitutions
    for f in ('module_utils.py', 'quant.py'):
        p = os.path.join(q, f)
        s = open(p).read()
        s = s.replace('.cuda()', ".to('cpu')").replace("device='cuda'", "device='cpu'")
        open(p, 'w').write(s)
    open(stamp, 'w').write('built from ' + REF + '\n')
    return OUT


OUT_GPU = os.path.join(HERE, '_ref_gpu')


def build_gpu():
    """oracle/_ref_gpu/ : a PLAIN copy of the reference's llmc package (no rewrite at all). On the GPU box the unmodified
    reference runs on the MI355X through PyTorch-ROCm (last sentence): the third arm of
    tools/parity_envelope.py and the source of the Triton-kernel goldens (tools/fp8_triton_golden.py). Git-ignored,
    travels with the gpurun snapshot, never imported by the product."""
    if not os.path.isdir(os.path.join(REF, 'llmc')):
        return None
    stamp = os.path.join(OUT_GPU, '.built')
    if os.path.exists(stamp):
        return OUT_GPU
    if os.path.isdir(OUT_GPU):
        shutil.rmtree(OUT_GPU)
    os.makedirs(OUT_GPU)
    shutil.copytree(os.path.join(REF, 'llmc'), os.path.join(OUT_GPU, 'llmc'),
                    ignore=shutil.ignore_patterns('__pycache__', '*.pyc'))
    open(stamp, 'w').write('plain copy of ' + REF + '/llmc\n')
    return OUT_GPU


if __name__ == '__main__':
    print(build(verbose=True))
    print(build_gpu())
