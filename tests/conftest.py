import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(autouse=True)
def _llmc_options_back_to_defaults():
    """A test that flips an A/B switch (llmc_amd._ffi.set_option / option) cannot leak it into the next test."""
    yield
    from llmc_amd import _ffi
    _ffi.reset_options()


class _Merged(dict):
    """Several golden files whose keys carry their case name as a prefix, read as one; `names` is the concatenation."""
    @property
    def files(self):
        return list(self.keys())


def load_golden(name):
    if name == 'gptq+more':          # the round-1 GPTQ cases and the round-3 ones (other bit widths / group sizes)
        out, names = _Merged(), []
        for f in ('gptq', 'gptq_more'):
            g = np.load(os.path.join(GOLDEN, f + '.npz'), allow_pickle=False)
            names += [str(n) for n in g['names']]
            out.update({k: g[k] for k in g.files if k != 'names'})
        out['names'] = np.array(names)
        return out
    if name == 'clip+more':          # same for the AutoClipper cases (meta: sym, gs, clip_sym, n_sample_token[, bit])
        out, names = _Merged(), []
        for f in ('clip', 'clip_more'):
            g = np.load(os.path.join(GOLDEN, f + '.npz'), allow_pickle=False)
            names += [str(n) for n in g['names']]
            out.update({k: g[k] for k in g.files if k != 'names'})
        for n in names:
            m = out[n + '/meta']
            if m.size == 4:
                out[n + '/meta'] = np.concatenate([m, [4]])
        out['names'] = np.array(names)
        return out
    if name == 'awq+more':           # round-1 AWQ cases (4 bit) and the round-3 ones (meta carries the bit width, gs 0 = per channel)
        out, names = _Merged(), []
        for f in ('awq', 'awq_more'):
            g = np.load(os.path.join(GOLDEN, f + '.npz'), allow_pickle=False)
            names += [str(n) for n in g['names']]
            out.update({k: g[k] for k in g.files if k != 'names'})
        for n in names:
            m = out[n + '/meta']
            if m.size == 4:
                out[n + '/meta'] = np.concatenate([m, [4]])
        out['names'] = np.array(names)
        return out
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


def report(test, **vals):
    """Statistical parity checks record the value they measured next to the bound they assert (LLMC_TEST_ACTUALS=<file>):
    the bounds of the end-to-end tests are derived from these records and from profiles/r03_parity_envelope.txt, not guessed."""
    path = os.environ.get('LLMC_TEST_ACTUALS')
    if path:
        import json
        with open(path, 'a') as f:
            f.write(json.dumps({'test': test, **{k: (float(v) if hasattr(v, '__float__') else v) for k, v in vals.items()}}) + '\n')
