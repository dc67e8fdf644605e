"""The C-ABI library loads without a GPU and exports every symbol include/llmc_hip.h declares; the ctypes
table binds exactly that set (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'llmc_hip.h')).read()
    assert 'llmc_test_' not in txt                      # test hooks live in include/llmc_hip_test.h, not in the product header
    txt += open(os.path.join(ROOT, 'include', 'llmc_hip_test.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(llmc_[a-z0-9_]+)\s*\(', txt)))


def test_library_builds_and_loads_on_cpu():
    from llmc_amd import build
    so = build.build()
    lib = ctypes.CDLL(so)
    lib.llmc_hip_abi_version.restype = ctypes.c_int
    assert lib.llmc_hip_abi_version() == 1


def test_every_declared_symbol_is_exported_and_bound():
    from llmc_amd import _ffi
    lib = _ffi.lib()
    syms = header_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f'declared in llmc_hip.h but not exported: {missing}'
    unbound = [s for s in syms if s not in _ffi.SIGNATURES]
    assert not unbound, f'declared in llmc_hip.h but not bound in _ffi.SIGNATURES: {unbound}'
    extra = [s for s in _ffi.SIGNATURES if s not in syms]
    assert not extra, f'bound but not declared in the header: {extra}'


def test_workspace_queries_are_pure_host_calls():
    from llmc_amd import _ffi
    lib = _ffi.lib()
    assert lib.llmc_minmax_qparams_ws_bytes(4096, 128) == 0
    assert lib.llmc_minmax_qparams_ws_bytes(1, 1 << 22) > 0
    assert lib.llmc_hessian_accum_ws_bytes(262144, 4096, 4096) > 136 * 256 * 256 * 4          # partial tiles of 136 tiles x S chunks, fp64 diagonal partials, barrier words
    assert lib.llmc_chol_inv_upper_ws_bytes(4096) >= 4096 * 4096 * 4
    assert lib.llmc_gptq_quantize_ws_bytes(4096, 4096) == 3 * 4096 * 512 * 4      # err columns of three groups in flight


def test_invalid_arguments_are_refused_without_touching_the_gpu():
    from llmc_amd import _ffi
    lib = _ffi.lib()
    rc = lib.llmc_minmax_qparams(None, 7, 1, 1, 1, 1, 0.0, 1.0, None, None, None, None)
    assert rc == -22
    assert 'dtype' in _ffi.last_error()
    rc = lib.llmc_pack_lsb(None, 1, 4, 4, 4, None, None)
    assert rc == -22
    rc = lib.llmc_awq_clip_errs_cand(None, None, None, None, 1, 64, 256, 256, 32, 32, 10, None, None)     # null operands
    assert rc == -22 and 'clip_errs_cand' in _ffi.last_error()


def test_product_refuses_cpu_tensors():
    import pytest
    import torch

    from llmc_amd import _ffi
    from llmc_amd.compression.quantization import IntegerQuantizer
    q = IntegerQuantizer(4, True, 'per_group', group_size=128)
    with pytest.raises(_ffi.LlmcHipError):
        q.get_tensor_qparams(torch.zeros(4, 128))


def test_a_stale_library_is_refused(monkeypatch):
    """llmc_hip_build_id() is the hash of csrc/ + include/llmc_hip.h + flags at build time; the loader recomputes it from
    the sources and fails loudly when they differ (a silently failed compile used to leave the old .so in place)."""
    import pytest
    from llmc_amd import _ffi, build
    lib = _ffi.lib()
    assert lib.llmc_hip_build_id().decode() == build.source_digest()
    monkeypatch.setattr(_ffi, '_lib', None)
    monkeypatch.setattr(build, 'source_digest', lambda: '0123456789abcdef')
    with pytest.raises(_ffi.LlmcHipError, match='stale'):
        _ffi.lib()
    monkeypatch.setenv('LLMC_SKIP_BUILD_ID_CHECK', '1')
    assert _ffi.lib() is not None
