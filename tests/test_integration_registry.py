"""INTEGRATION.md §1, executed: bind this repo's classes into the REFERENCE's own registry object
(llmc/utils/registry_factory.py:9-23) and look them up the way llmc/__main__.py:43,62 does.
Runs on CPU; needs the reference tree (present in the build container, absent on the GPU box -> skipped)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

SCRIPT = r'''
import sys
sys.path.insert(0, %(shims)r); sys.path.insert(0, %(ref)r); sys.path.insert(0, %(root)r)
import importlib.util
# the reference's registry module, loaded from its own file (importing the whole llmc.utils package would pull
# eval / dataset dependencies that are not installed here)
spec = importlib.util.spec_from_file_location('llmc_registry_factory', %(ref)r + '/llmc/utils/registry_factory.py')
reg = importlib.util.module_from_spec(spec); spec.loader.exec_module(reg)
R = reg.ALGO_REGISTRY
assert type(R).__name__ == 'Register' and not hasattr(R, 'bind')

# llmc's own classes are registered first, under the same keys (what `import llmc.compression.quantization` does)
@R
class GPTQ: pass
@R
class Awq: pass
@R
class RTN: pass
@R
class SpQR: pass

# ---- INTEGRATION.md section 1, verbatim ----
import llmc_amd
llmc_amd.register_into(R)
# --------------------------------------------
import llmc_amd.compression.quantization as Q
for key in ('GPTQ', 'Awq', 'RTN', 'SpQR'):
    assert R[key] is getattr(Q, key), key          # llmc/__main__.py:62: ALGO_REGISTRY[config.quant.method]
    assert key in R
# and the decorator protocol of the reference's Register accepts our classes too (a fresh registry)
R2 = reg.Register()
for key in ('GPTQ', 'Awq', 'RTN', 'SpQR'):
    assert R2(getattr(Q, key)) is getattr(Q, key)
try:
    R2(Q.GPTQ)
    raise SystemExit('double registration must raise like the reference')
except Exception as e:
    assert 'already exists' in str(e)
print('REGISTRY_OK')
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
def test_register_into_reference_registry():
    code = SCRIPT % {'shims': os.path.join(ROOT, 'oracle', '_shims'), 'ref': REF, 'root': ROOT}
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'REGISTRY_OK' in r.stdout, r.stdout + r.stderr


def test_register_into_any_mapping_in_process():
    import llmc_amd
    import llmc_amd.compression.quantization as Q
    d = {'GPTQ': object}
    bound = llmc_amd.register_into(d)
    assert d['GPTQ'] is Q.GPTQ and d['Awq'] is Q.Awq and d['RTN'] is Q.RTN and d['SpQR'] is Q.SpQR and set(bound) == {'GPTQ', 'Awq', 'RTN', 'SpQR'}
