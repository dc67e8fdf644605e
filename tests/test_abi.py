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


def test_options_are_explicit_per_thread_switches_not_environment_variables():
    """llmc_hip_set_option / get_option / option_name (round 6): every A/B switch of the library is a named per-thread option, 0 by
    default; unknown keys are refused; the sources read no environment variable outside -DLLMC_LAB builds."""
    import glob
    import threading

    import pytest

    from llmc_amd import _ffi
    keys = _ffi.library_options()
    assert len(keys) >= 14 and len(set(keys)) == len(keys) and 'k3_no_planes' in keys and 'k1_fp32_diag' in keys
    assert all(_ffi.get_option(k) == 0 for k in keys)
    with _ffi.option(k3_no_planes=1, gemm3s_min_tiles=7, awq_kt=0):
        assert _ffi.get_option('k3_no_planes') == 1 and _ffi.get_option('gemm3s_min_tiles') == 7 and _ffi.HOST_OPTIONS['awq_kt'] == 0
        seen = []
        t = threading.Thread(target=lambda: seen.append(_ffi.get_option('k3_no_planes')))      # another thread: its own defaults
        t.start()
        t.join()
        assert seen == [0]
    assert _ffi.get_option('k3_no_planes') == 0 and _ffi.HOST_OPTIONS['awq_kt'] == 1
    with pytest.raises(ValueError):
        _ffi.set_option('no_such_switch', 1)
    assert _ffi.lib().llmc_hip_set_option(b'k3_fp32', -1) == -22
    n = 0
    for f in glob.glob(os.path.join(ROOT, 'llmc_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'llmc_amd', 'csrc', '*.h')):
        n += open(f).read().count('getenv(')
    assert n == 1          # the lab_env() helper of -DLLMC_LAB builds (common.h)


def test_hessian_multi_problem_plan_is_a_pure_host_call():
    """llmc_hessian_accum_multi_ws_bytes: up to llmc_hessian_max_problems() Hessians and llmc_hessian_max_samples() samples per call;
    the workspace holds every problem's partial tiles + fp64 diagonal partials; bad calls return 0 without touching a GPU."""
    import ctypes as C

    from llmc_amd import _ffi
    L = _ffi.lib()
    assert L.llmc_hessian_max_problems() == 4 and L.llmc_hessian_max_samples() == 512

    def problems(P, n, T, K):
        arr = (_ffi.HessianProblem * P)()
        keep = []
        for i in range(P):
            Ts = (C.c_int64 * n)(*([T] * n))
            keep.append(Ts)
            arr[i].T_list_host, arr[i].n, arr[i].K, arr[i].ldx = C.cast(Ts, C.c_void_p), n, K, K
            arr[i].n_before, arr[i].n_after = 0.0, float(n)
        return arr, keep
    one, k1 = problems(1, 128, 2048, 4096)
    three, k3 = problems(3, 128, 2048, 4096)
    b1, b3 = L.llmc_hessian_accum_multi_ws_bytes(one, 1), L.llmc_hessian_accum_multi_ws_bytes(three, 3)
    tile = 256 * 256 * 4
    assert b1 > 9 * 136 * tile                       # 9 token chunks of 136 tiles when the Hessian has a launch to itself (4.8 rounds)
    assert 3 * 5 * 136 * tile < b3 < 3 * 6 * 136 * tile        # 5 chunks each when three share the unit queue: 2040 units = 7.97 rounds
    assert L.llmc_hessian_accum_multi_ws_bytes(three, 5) == 0       # more problems than a launch takes
    many, km = problems(4, 129, 2048, 4096)                          # 516 samples > 512
    assert L.llmc_hessian_accum_multi_ws_bytes(many, 4) == 0
    assert L.llmc_hessian_accum_ptrs_ws_bytes(k1[0], 128, 4096, 4096) == b1
