"""ctypes binding of libllmc_hip.so (include/llmc_hip.h). Thin: raw device pointers + stream in,
status code out. The library is required: every failure to load or run raises (no fallback)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libllmc_hip.so')

F16, BF16, F32 = 0, 1, 2
OUT_FAKE, OUT_I32, OUT_I8, OUT_U8 = 0, 1, 2, 3
SCALAR_QPARAM = 16   # LLMC_SCALAR_QPARAM
LINEAR_YBLOCKED = 4   # LLMC_LINEAR_YBLOCKED
FRACTIONAL_ZP = 32    # LLMC_FRACTIONAL_ZP

_DT = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}

_i64, _i32, _f32, _f64, _vp, _sz = C.c_int64, C.c_int, C.c_float, C.c_double, C.c_void_p, C.c_size_t

# name -> (restype, argtypes). Must list every symbol include/llmc_hip.h declares
# (tests/test_abi.py parses the header and checks both directions).
SIGNATURES = {
    'llmc_hip_abi_version': (_i32, []),
    'llmc_hip_build_id': (C.c_char_p, []),
    'llmc_hip_last_error': (_i32, [C.c_char_p, _sz]),
    'llmc_hip_set_helper_streams': (_i32, [_i32]),
    'llmc_hip_set_cu_reserve': (_i32, [_i32]),
    'llmc_hip_set_option': (_i32, [C.c_char_p, _i32]),
    'llmc_hip_get_option': (_i32, [C.c_char_p]),
    'llmc_hip_option_name': (_i32, [_i32, C.c_char_p, _sz]),
    'llmc_minmax_qparams_ws_bytes': (_sz, [_i64, _i64]),
    'llmc_minmax_qparams': (_i32, [_vp, _i32, _i64, _i64, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp]),
    'llmc_minmax_samples_max': (_i32, []),
    'llmc_minmax_samples_ws_bytes': (_sz, [_vp, _i32]),
    'llmc_minmax_samples': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    'llmc_histc_ws_bytes': (_sz, [_i32]),
    'llmc_histc': (_i32, [_vp, _i32, _i64, _i32, _f32, _f32, _vp, _vp, _vp]),
    'llmc_mse_qparams': (_i32, [_vp, _i32, _i64, _i64, _i32, _i32, _f32, _f32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    'llmc_quant_static': (_i32, [_vp, _i32, _i64, _i64, _vp, _i32, _vp, _i32, _f32, _f32, _i32, _vp, _vp]),
    'llmc_quant_dynamic_ws_bytes': (_sz, [_i64, _i64]),
    'llmc_quant_dynamic': (_i32, [_vp, _i32, _i64, _i64, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'llmc_pack_lsb': (_i32, [_vp, _i32, _i64, _i64, _i32, _vp, _vp]),
    'llmc_hessian_accum_ws_bytes': (_sz, [_i64, _i64, _i64]),
    'llmc_hessian_accum': (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _f64, _f64, _vp, _vp]),
    'llmc_hessian_accum_partials': (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _vp]),
    'llmc_hessian_accum_reduce': (_i32, [_vp, _i64, _i64, _i64, _f64, _f64, _vp, _vp]),
    'llmc_hessian_max_samples': (_i32, []),
    'llmc_hessian_accum_ptrs_ws_bytes': (_sz, [_vp, _i32, _i64, _i64]),
    'llmc_hessian_accum_ptrs': (_i32, [_vp, _vp, _vp, _i32, _i32, _i64, _i64, _f64, _f64, _vp, _vp]),
    'llmc_hessian_accum_ptrs_partials': (_i32, [_vp, _vp, _i32, _i32, _i64, _i64, _vp, _vp]),
    'llmc_hessian_accum_ptrs_reduce': (_i32, [_vp, _vp, _i32, _i64, _i64, _f64, _f64, _vp, _vp]),
    'llmc_hessian_accum_barrier_timeouts': (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _vp]),
    'llmc_hessian_max_problems': (_i32, []),
    'llmc_hessian_accum_multi_ws_bytes': (_sz, [_vp, _i32]),
    'llmc_hessian_accum_multi_partials': (_i32, [_vp, _i32, _i32, _vp, _vp]),
    'llmc_hessian_accum_multi_reduce': (_i32, [_vp, _i32, _vp, _vp]),
    'llmc_hessian_accum_multi_barrier_timeouts': (_i32, [_vp, _i32, _vp, _vp, _vp]),
    'llmc_gather_cols': (_i32, [_vp, _i64, _i64, _vp, _vp, _vp]),
    'llmc_hessian_prep_ws_bytes': (_sz, [_i64]),
    'llmc_hessian_prep': (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _f32, _vp, _vp, _vp, _vp]),
    'llmc_chol_inv_upper_ws_bytes': (_sz, [_i64]),
    'llmc_chol_inv_upper': (_i32, [_vp, _i64, _vp, _vp, _vp]),
    'llmc_hessian_prep_rev': (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _f32, _vp, _vp, _vp, _vp]),
    'llmc_chol_inv_upper_rev': (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    'llmc_gptq_quantize_ws_bytes': (_sz, [_i64, _i64]),
    'llmc_gptq_quantize': (_i32, [_vp, _vp, _i64, _i64, _i32, _f32, _f32, _i64, _i32, _vp, _vp, _vp, _vp, _vp,
                                  _i32, _vp, _vp]),
    'llmc_gptq_quantize_cols': (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _f32, _f32, _i64, _i32, _vp, _vp, _vp, _vp, _vp,
                                       _i32, _vp, _vp]),
    'llmc_spqr_quantize_ws_bytes': (_sz, [_i64, _i64]),
    'llmc_spqr_quantize': (_i32, [_vp, _vp, _i64, _i64, _f32, _f32, _i64, _f32, _i32, _f32, _f32, _f32, _f32, _vp, _vp,
                                  _vp, _vp, _vp, _i32, _vp, _vp]),
    'llmc_awq_act_mean_ws_bytes': (_sz, [_i64, _i64]),
    'llmc_awq_act_mean': (_i32, [_vp, _i32, _i64, _i64, _vp, _vp, _vp]),
    'llmc_awq_weight_mean_ws_bytes': (_sz, [_i64, _i64]),
    'llmc_awq_weight_mean': (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp]),
    'llmc_awq_scales': (_i32, [_vp, _vp, _i32, _i64, _f64, _i32, _vp, _vp]),
    'llmc_awq_scale_fakequant': (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _i32, _f32, _f32, _vp, _vp]),
    'llmc_div_cols': (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _vp]),
    'llmc_div_cols_kt': (_i32, [_vp, _vp, _i32, _i64, _i64, _vp, _vp]),
    'llmc_mul_cols': (_i32, [_vp, _vp, _i32, _i64, _i64, _vp]),
    'llmc_clamp_groups': (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp]),
    'llmc_linear_eval_ws_bytes': (_sz, [_i64, _i64, _i64]),
    'llmc_linear_eval': (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    'llmc_linear_eval_yblocked_bytes': (_sz, [_i64, _i64]),
    'llmc_ktile_pack': (_i32, [_vp, _i32, _i64, _i64, _vp, _vp]),
    'llmc_linear_eval_kt': (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    'llmc_awq_clip_search_ws_bytes': (_sz, [_i64, _i64, _i64, _i64]),
    'llmc_awq_clip_search': (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _f32,
                                    _vp, _vp, _vp, _vp]),
    'llmc_awq_clip_errs': (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp]),
    'llmc_awq_clip_errs_cand': (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _i64, _i32, _vp, _vp]),
    'llmc_fp8_quant_ws_bytes': (_sz, [_i64, _i64]),
    'llmc_fp8_quant': (_i32, [_vp, _i32, _i64, _i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    'llmc_fp8_block_quant': (_i32, [_vp, _i32, _i64, _i64, _i32, _f32, _i32, _vp, _vp, _vp]),
    'llmc_fp8_block_dequant': (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    'llmc_fp8_act_quant': (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    'llmc_fp8_block_gemm': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp]),
    'llmc_pack_awq_gemm': (_i32, [_vp, _i32, _vp, _i32, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    'llmc_test_sgemm': (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                               _i32, _i32, _vp]),
    'llmc_test_sgemm_phased': (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    'llmc_test_gemm3': (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'llmc_test_gemm3_planes': (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'llmc_test_gemm3s_stamps': (_i32, [_vp]),
}



class HessianProblem(C.Structure):
    """llmc_hessian_problem_t (include/llmc_hip.h)"""
    _fields_ = [('H', _vp), ('dstate', _vp), ('X_list_host', _vp), ('T_list_host', _vp), ('n', _i32), ('K', _i64),
                ('ldx', _i64), ('n_before', _f64), ('n_after', _f64)]


_lib = None


class LlmcHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the library. Raises if it is missing — build with `python -m llmc_amd.build`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LlmcHipError(
                f'{LIB_PATH} not found: the HIP extension is required (run `python -m llmc_amd.build`). '
                'llmc_amd has no CPU fallback.')
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        ver = handle.llmc_hip_abi_version()
        if ver != 1:
            raise LlmcHipError(f'libllmc_hip.so ABI version {ver} != 1')
        # a library built from other sources than the ones lying next to it (a failed or forgotten rebuild) fails loudly
        if not os.environ.get('LLMC_SKIP_BUILD_ID_CHECK'):
            from . import build as _build
            have, want = handle.llmc_hip_build_id().decode(), _build.source_digest()
            if have != want:
                raise LlmcHipError(f'{LIB_PATH} is stale: built from sources {have}, csrc/ is {want} '
                                   '(run `python -m llmc_amd.build`; LLMC_SKIP_BUILD_ID_CHECK=1 skips this check)')
        _lib = handle
        _apply_startup_options()
    return _lib


# ---- explicit A/B switches --------------------------------------------------------------------------------------------------
# The library reads no environment variable; its switches are llmc_hip_set_option(key, value) (per calling thread, table in
# include/llmc_hip.h). Host-side switches of the Python layer live in HOST_OPTIONS. `option(k3_no_planes=1)` sets either kind
# for the duration of a `with` block. For command-line A/B runs (tools/) ONE variable is read, once, when the library is
# loaded: LLMC_OPTIONS="key=value,key=value".
HOST_OPTIONS = {
    'awq_kt': 1,                              # 0: AWQ products on row-major operands and the 8-wave kernel (round-1 path)
    'awq_y_bytes': (1 << 32) - (1 << 20),     # chunk bound of the tile-blocked reference output (tests lower it)
    'k3_fused_prep': 1,                       # 0: llmc_hessian_prep + llmc_chol_inv_upper instead of the reversed gather + in-place factor
}

_HOST_DEFAULTS = dict(HOST_OPTIONS)


def reset_options():
    """Every switch back to its default (tests)."""
    HOST_OPTIONS.update(_HOST_DEFAULTS)
    if _lib is not None:
        for k in library_options():
            _lib.llmc_hip_set_option(k.encode(), 0)


def set_option(key, value):
    """Set a switch; returns the previous value. Library keys: llmc_hip_set_option; host keys: HOST_OPTIONS."""
    if key in HOST_OPTIONS:
        prev, HOST_OPTIONS[key] = HOST_OPTIONS[key], int(value)
        return prev
    prev = lib().llmc_hip_set_option(key.encode(), int(value))
    if prev < 0:
        raise ValueError(f'unknown option {key!r} (library keys: {library_options()}; host keys: {sorted(HOST_OPTIONS)})')
    return prev


def get_option(key):
    if key in HOST_OPTIONS:
        return HOST_OPTIONS[key]
    v = lib().llmc_hip_get_option(key.encode())
    if v < 0:
        raise ValueError(f'unknown option {key!r}')
    return v


def library_options():
    out, buf, i = [], C.create_string_buffer(64), 0
    while lib().llmc_hip_option_name(i, buf, 64) > 0:
        out.append(buf.value.decode())
        i += 1
    return out


class option:
    """with _ffi.option(k3_no_planes=1, gemm3s_min_tiles=1): ..."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.prev = {k: set_option(k, v) for k, v in self.kv.items()}
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(k, v)
        return False


def _apply_startup_options():
    spec = os.environ.get('LLMC_OPTIONS', '')
    for item in filter(None, (x.strip() for x in spec.split(','))):
        k, _, v = item.partition('=')
        set_option(k.strip(), int(v or 1))


def last_error():
    buf = C.create_string_buffer(512)
    lib().llmc_hip_last_error(buf, 512)
    return buf.value.decode(errors='replace')


def check(rc, what):
    if rc == 0:
        return
    msg = last_error()
    if rc == -22:
        raise ValueError(f'{what}: invalid argument: {msg}')
    if rc == -95:
        raise NotImplementedError(f'{what}: unsupported: {msg}')
    raise LlmcHipError(f'{what}: HIP error: {msg}')


def dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise ValueError(f'unsupported dtype {t.dtype if isinstance(t, torch.Tensor) else t}')


def require_gpu(*tensors):
    """The product path runs on the GPU only; refuse anything else loudly."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise LlmcHipError(
                'llmc_amd operators run on MI355X only (tensor is on '
                f'{t.device}); there is no CPU fallback. Use oracle/ for CPU checks in tests.')
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            # the C side launches on the CURRENT device's stream and keys helper streams / attributes on it
            raise LlmcHipError(f'tensor lives on {t.device} but the current device is cuda:{cur}: wrap the call in '
                               '`with torch.cuda.device(tensor.device)` (hipSetDevice is the caller\'s job)')


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def workspace(nbytes, device):
    if nbytes <= 0:
        return None
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


class helper_streams:
    """with _ffi.helper_streams(False): ...  — K3/K4 entry points called inside keep all their work on the caller's
    stream (llmc_hip_set_helper_streams): for callers that overlap several subsets on streams of their own."""

    def __init__(self, enable):
        self.enable = int(bool(enable))

    def __enter__(self):
        self.prev = lib().llmc_hip_set_helper_streams(self.enable)

    def __exit__(self, *exc):
        lib().llmc_hip_set_helper_streams(self.prev)
        return False


class cu_reserve:
    """with _ffi.cu_reserve(32): ...  — Hessian kernels launched inside leave 32 CUs to the other streams
    (llmc_hip_set_cu_reserve)."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.prev = lib().llmc_hip_set_cu_reserve(self.n)

    def __exit__(self, *exc):
        lib().llmc_hip_set_cu_reserve(self.prev)
        return False
