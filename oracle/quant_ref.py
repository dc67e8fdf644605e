"""Numpy restatement of llmc's IntegerQuantizer / FloatQuantizer arithmetic and the int packers.

Follows (paths relative to the llmc tree):
  llmc/compression/quantization/quant.py:132-143  get_minmax_range
  llmc/compression/quantization/quant.py:545-559  get_qparams
  llmc/compression/quantization/quant.py:612-658  reshape_tensor / restore_tensor
  llmc/compression/quantization/quant.py:699-717  quant / dequant / quant_dequant
  llmc/compression/quantization/quant.py:785-953  fake/real quant weight static/dynamic
  llmc/compression/quantization/quant.py:1043-1072,1195-1221  FloatQuantizer (e4m3; qtorch pinned to
      torch.float8_e4m3fn's RNE cast — parity otherwise unpinned, qtorch is not vendored)
  llmc/compression/quantization/module_utils.py:836-862   VllmRealQuantLinear.pack
  llmc/compression/quantization/module_utils.py:1004-1065 AutoawqRealQuantLinear.gemm_pack
Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np

F16, BF16, F32 = 'f16', 'bf16', 'f32'


def rnd(a, dt):
    """Round fp32 values to dtype dt (RNE) and return them as fp32."""
    a = np.asarray(a, dtype=np.float32)
    if dt == F32:
        return a
    if dt == F16:
        with np.errstate(over='ignore'):
            return a.astype(np.float16).astype(np.float32)
    if dt == BF16:
        x = a.view(np.uint32).astype(np.uint64)
        nan = (x & 0x7fffffff) > 0x7f800000
        lsb = (x >> 16) & 1
        y = ((x + 0x7fff + lsb) >> 16) << 16
        y = np.where(nan, ((x >> 16) | 0x40) << 16, y)
        return (y & 0xffffffff).astype(np.uint32).view(np.float32).reshape(a.shape)
    raise ValueError(dt)


def promote(a, b):
    return a if a == b else F32


def int_range(bit, sym):
    """quant.py:665-677"""
    if sym:
        return float(-(2 ** (bit - 1))), float(2 ** (bit - 1) - 1)
    return 0.0, float(2 ** bit - 1)


def reshape_rows(w, granularity, group_size=None):
    """quant.py:612-642 (per_group / per_channel / per_tensor). Returns a 2-D [G, g] view."""
    w = np.asarray(w, dtype=np.float32)
    if granularity == 'per_group':
        if w.shape[-1] >= group_size:
            assert w.shape[-1] % group_size == 0
            return w.reshape(-1, group_size)
        return w.reshape(-1, w.shape[-1])
    if granularity == 'per_tensor':
        return w.reshape(1, -1)
    return w.reshape(-1, w.shape[-1])  # per_channel / per_token: reduce over last dim


def qparams_from_minmax(mn, mx, dt, sym, qmin, qmax, round_zp=True):
    """quant.py:545-559. mn/mx: values of dtype dt."""
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        eps = rnd(np.float32(1e-5), dt)
        if sym:
            a = np.maximum(np.abs(mx), np.abs(mn))
            a = np.maximum(a, eps)
            s = rnd(a / rnd(np.float32(qmax), dt), dt)
            z = np.zeros_like(s)
        else:
            d = rnd(mx - mn, dt)
            d = np.maximum(d, eps)
            s = rnd(d / rnd(np.float32(qmax - qmin), dt), dt)
            r = rnd(mn / s, dt)
            if round_zp:
                r = np.rint(r)
                z = rnd(np.float32(qmin) - r, dt)
                z = np.minimum(np.maximum(z, np.float32(qmin)), np.float32(qmax))
            else:
                z = rnd(np.float32(qmin) - r, dt)
    return s.astype(np.float32), z.astype(np.float32)


def minmax_qparams(w2d, dt, sym, qmin, qmax, round_zp=True):
    """get_tensor_qparams on an already reshaped [G, g] array -> (scales [G,1], zeros [G,1])."""
    mx = w2d.max(axis=-1, keepdims=True)
    mn = w2d.min(axis=-1, keepdims=True)
    return qparams_from_minmax(mn, mx, dt, sym, qmin, qmax, round_zp)


def quant_codes(w2d, wdt, s, sdt, z, zdt, qmin, qmax):
    """quant.py:699-701 (round_zp=True): clamp(round(x / s) + z, qmin, qmax); values are integers."""
    p1 = promote(wdt, sdt)
    p2 = promote(p1, zdt) if zdt is not None else p1
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        t = rnd(w2d / s, p1)
        t = np.rint(t)
        t = rnd(t + (z if z is not None else np.float32(0.0)), p2)
        t = np.minimum(np.maximum(t, np.float32(qmin)), np.float32(qmax))
    return t.astype(np.float32), p2


def dequant(q, s, z, p2):
    """quant.py:709-712"""
    with np.errstate(over='ignore', invalid='ignore'):
        t = rnd(q - (z if z is not None else np.float32(0.0)), p2)
        return rnd(t * s, p2)


def fake_quant_static(w2d, wdt, s, sdt, z, zdt, qmin, qmax):
    q, p2 = quant_codes(w2d, wdt, s, sdt, z, zdt, qmin, qmax)
    return rnd(dequant(q, s, z, p2), wdt)


def fake_quant_dynamic(w2d, dt, sym, qmin, qmax):
    s, z = minmax_qparams(w2d, dt, sym, qmin, qmax)
    return fake_quant_static(w2d, dt, s, dt, z, dt, qmin, qmax), s, z


def real_quant_dynamic(w2d, dt, sym, qmin, qmax):
    s, z = minmax_qparams(w2d, dt, sym, qmin, qmax)
    q, _ = quant_codes(w2d, dt, s, dt, z, dt, qmin, qmax)
    return q.astype(np.int32), s, (None if sym else z.astype(np.int32))


def pack_lsb(codes, bits):
    """module_utils.py:836-862: u = uint8(code + 2^(b-1)); word |= u[:, i::pf] << (b*i)."""
    codes = np.asarray(codes)
    off = (2 ** bits) // 2
    u = ((codes.astype(np.int64) + off) & 0xff).astype(np.uint32)
    pf = 32 // bits
    R, K = u.shape
    Kp = -(-K // pf)
    pad = Kp * pf - K
    u = np.pad(u, [(0, 0), (0, pad)], constant_values=0)
    packed = np.zeros((R, Kp), dtype=np.uint32)
    for i in range(pf):
        packed |= u[:, i::pf] << np.uint32(bits * i)
    return packed.view(np.int32)


AWQ_ORDER = [0, 2, 4, 6, 1, 3, 5, 7]


def pack_awq_gemm(w, scales, zeros, group_size, bits=4):
    """module_utils.py:1004-1065. w [R,K] f16 values; scales [R,K/g] (any dt) ; zeros [R,K/g] int.
    Returns qweight [K, R/8] int32, scales [K/g, R] f16 values, qzeros [K/g, R/8] int32."""
    assert bits == 4
    w = np.asarray(w, dtype=np.float32)
    sc = rnd(np.asarray(scales, dtype=np.float32).T.copy(), F16)      # [K/g, R] f16
    zr = np.asarray(zeros).T.astype(np.float32)                        # [K/g, R] (int values)
    # zeros (int32) * scales (f16) -> f16 in torch
    sz = rnd(zr * sc, F16)
    R, K = w.shape
    gidx = np.arange(K) // group_size
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        t = rnd(w + sz[gidx].T, F16)          # weight[:, idx] + scale_zeros[idx // g]
        t = rnd(t / sc[gidx].T, F16)
        iw = np.rint(t).astype(np.int32)      # [R, K]
    iw = iw.T.copy()                          # [K, R]
    pack = 32 // bits
    qw = np.zeros((K, R // pack), dtype=np.int32)
    for i in range(pack):
        qw |= (iw[:, AWQ_ORDER[i]::pack] << (bits * i)).astype(np.int32)
    zi = np.asarray(zeros).T.astype(np.int32)
    qz = np.zeros((zi.shape[0], R // pack), dtype=np.int32)
    for i in range(pack):
        qz |= (zi[:, AWQ_ORDER[i]::pack] << (bits * i)).astype(np.int32)
    return qw, sc, qz


# ---- FP8 e4m3fn (OCP) --------------------------------------------------------------------------
def f32_to_e4m3fn_bits(a):
    """RNE cast fp32 -> float8_e4m3fn bit pattern (torch's .to(torch.float8_e4m3fn): no saturation,
    |x| > 464 -> NaN (0x7f), 448 < |x| <= 464 -> 448)."""
    a = np.asarray(a, dtype=np.float32)
    sign = (a.view(np.uint32) >> 31).astype(np.uint8) << 7
    x = np.abs(a).astype(np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    nan = np.isnan(a)
    # subnormal grid: multiples of 2^-9 below 2^-6
    e = np.floor(np.log2(np.where(x > 0, x, 1.0)))
    e = np.clip(e, -6, 8)
    step = np.exp2(e - 3)
    q = np.rint(x / step) * step              # RNE on the local grid (rint = half-even on the quotient)
    # re-normalise when rounding crossed a binade
    e2 = np.floor(np.log2(np.where(q > 0, q, 1.0)))
    e2 = np.clip(e2, -6, 8)
    mant = q / np.exp2(e2)                    # in [1,2) for normals, [0,1) for subnormals
    is_sub = q < 2.0 ** -6
    exp_field = np.where(is_sub, 0, e2 + 7).astype(np.int64)
    man_field = np.where(is_sub, np.rint(q / 2.0 ** -9), np.rint((mant - 1.0) * 8)).astype(np.int64)
    bits = (exp_field << 3) | man_field
    over = q > 448.0
    bits = np.where(over | nan, 0x7f, bits)
    out = (bits.astype(np.uint8) | sign)
    return out


def e4m3fn_bits_to_f32(b):
    b = np.asarray(b, dtype=np.uint8)
    sign = np.where(b & 0x80, -1.0, 1.0)
    e = ((b >> 3) & 0xf).astype(np.int64)
    m = (b & 7).astype(np.float64)
    v = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * np.exp2(e - 7.0))
    v = np.where((b & 0x7f) == 0x7f, np.nan, v)
    return (sign * v).astype(np.float32)


# ---- qtorch.quant.float_quantize(x, exp, man, rounding='nearest') ---------------------------------------------------------
# THIRD-PARTY ARITHMETIC, ABSENT FROM /root/reference: llmc's FloatQuantizer.quant (quant.py:1061-1072) calls
# `float_quantize(scaled.float(), e_bits, m_bits, rounding='nearest')` of QPyTorch ("qtorch", requirements/runtime.txt:29,
# NO version pin; the last release is 0.3.0). qtorch is not installed here and there is no network, so its published
# algorithm is RESTATED below from QPyTorch 0.3.0's CPU kernel (qtorch/quant/quant_cpu/quant_cpu.cpp:float_quantize_nearest,
# bit_helper.cpp:round_bitwise_nearest / clip_exponent; Python defaults subnormals=True, saturate=True) — parity for this
# one function is anchored on that restatement, not on a run of qtorch itself:
#   target_exp = biased exponent of x - 127;  min_exp = -(2^(exp-1) - 2)
#   x in the target's subnormal range (target_exp < min_exp):   shift = sign(x) * 2^min_exp;  q = rnd(x + shift) - shift
#   otherwise:                                                  q = rnd(x), then clip_exponent
#   rnd(v): (bits(v) + (1 << (22 - man))) & ~((1 << (23 - man)) - 1)     -> round to nearest, TIES AWAY FROM ZERO
#   clip_exponent: max exponent = 2^(exp-1) - 1 (the top exponent code is kept for infinity, IEEE style); a value that
#                  rounds beyond it SATURATES to +-(2 - 2^-man) * 2^max_exp.
# For (exp, man) = (4, 3) this is an IEEE-style e4m3 whose largest value is 240 — NOT the OCP e4m3fn of
# torch.float8_e4m3fn (largest 448, round to nearest EVEN, overflow -> NaN) that llmc takes its qmax = 448 from
# (quant.py:985-1003). The two differ (tests/test_oracle_golden.py::test_qtorch_vs_torch_cast_where_they_differ):
#   * every |x / s| >= 248 (the binades [256, 448] and the rounding interval below them): qtorch returns +-240, the cast
#     returns the nearest of 256, 288, ..., 448;   * exact ties: away from zero vs to even;   * |x / s| > 464: 240 vs NaN.
# For (5, 2) the grids coincide (largest value 57344 either way); only ties and the overflow value differ.
def qtorch_float_quantize(x, exp, man):
    x = np.ascontiguousarray(x, dtype=np.float32)
    bits = x.view(np.uint32)
    sign = bits & np.uint32(0x80000000)
    t_exp = ((bits & np.uint32(0x7fffffff)) >> np.uint32(23)).astype(np.int64) - 127
    min_exp = -((1 << (exp - 1)) - 2)
    max_store = (1 << (exp - 1)) - 1 + 127
    mask = np.uint32((1 << (23 - man)) - 1)
    half = np.uint32(1 << (23 - man - 1))

    def rnd_bits(b):
        return (b + half) & ~mask

    # normal range
    q = rnd_bits(bits)
    qe = ((q & np.uint32(0x7fffffff)) >> np.uint32(23)).astype(np.int64)
    max_man = np.uint32(((1 << man) - 1) << (23 - man))
    sat = sign | np.uint32(max_store << 23) | max_man
    q = np.where((q != 0) & (qe > max_store), sat, q).astype(np.uint32)
    normal = q.view(np.float32)
    # subnormal range of the target format
    shift_bits = (np.uint32((127 + min_exp) << 23) | sign).astype(np.uint32)
    shift = shift_bits.view(np.float32)
    with np.errstate(over='ignore', invalid='ignore'):
        val = (x + shift).astype(np.float32)
        sub = (rnd_bits(val.view(np.uint32)).astype(np.uint32).view(np.float32) - shift).astype(np.float32)
    return np.where(t_exp < min_exp, sub, normal).astype(np.float32)


def f32_to_e5m2_bits(a):
    """torch's .to(torch.float8_e5m2): IEEE-style binary8 (5, 2), round to nearest even, overflow -> inf (0x7c), NaN 0x7f."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = a.view(np.uint32)
    sign = ((b >> np.uint32(24)) & np.uint32(0x80)).astype(np.uint8)
    x = np.abs(a).astype(np.float64)
    nan = np.isnan(a)
    e = np.floor(np.log2(np.where(x > 0, x, 1.0)))
    e = np.clip(e, -14, 15)
    step = np.exp2(e - 2)
    q = np.rint(x / step) * step
    e2 = np.clip(np.floor(np.log2(np.where(q > 0, q, 1.0))), -14, 15)
    is_sub = q < 2.0 ** -14
    exp_field = np.where(is_sub, 0, e2 + 15).astype(np.int64)
    man_field = np.where(is_sub, np.rint(q / 2.0 ** -16), np.rint((q / np.exp2(e2) - 1.0) * 4)).astype(np.int64)
    bits = (exp_field << 2) | man_field
    bits = np.where(q > 57344.0, 0x7c, bits)
    bits = np.where(np.isinf(a), 0x7c, bits)
    bits = np.where(nan, 0x7f, bits)
    return bits.astype(np.uint8) | sign


def e5m2_bits_to_f32(b):
    b = np.asarray(b, dtype=np.uint8)
    return (b.astype(np.uint16) << np.uint16(8)).view(np.float16).astype(np.float32)


FP8_FORMATS = {'e4m3': (4, 3, 448.0), 'e5m2': (5, 2, 57344.0)}


def fp8_encode(t, fmt='e4m3', sem='cast'):
    """(codes uint8, decoded values fp32) of the scaled tensor t. sem 'cast': torch's dtype cast (float8_e4m3fn /
    float8_e5m2, RNE) — make_golden's stand-in of rounds 1-3 and the arithmetic of the reference's Triton kernels;
    sem 'qtorch': qtorch.float_quantize as FloatQuantizer.quant calls it (restated above), followed by the exact cast."""
    e, m, _ = FP8_FORMATS[fmt]
    to_bits, from_bits = (f32_to_e4m3fn_bits, e4m3fn_bits_to_f32) if fmt == 'e4m3' else (f32_to_e5m2_bits, e5m2_bits_to_f32)
    if sem == 'qtorch':
        v = qtorch_float_quantize(t, e, m)
        return to_bits(v), v
    b = to_bits(t)
    return b, from_bits(b)


def fp8_quant(w2d, dt, fmt='e4m3', sem='cast'):
    """FloatQuantizer sym weight path (quant.py:545-553 with qmax = tensor(finfo.max), :1061-1072, :1211).
    Per-tensor (one row): absmax is a 0-dim dt tensor, absmax / tensor(qmax) promotes to fp32 and x / scale keeps
    the tensor dtype with an fp32 scalar. Per-channel: everything stays in dt.
    Returns (bits uint8 [G,g], scales [G,1] fp32 container, scale dtype)."""
    fmax = np.float32(FP8_FORMATS[fmt][2])
    mx = w2d.max(axis=-1, keepdims=True)
    mn = w2d.min(axis=-1, keepdims=True)
    a = np.maximum(np.maximum(np.abs(mx), np.abs(mn)), rnd(np.float32(1e-5), dt))
    sdt = F32 if w2d.shape[0] == 1 else dt
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        s = rnd(a / fmax, sdt)
        s = np.where(s == 0, np.float32(1.0), s).astype(np.float32)      # scales[scales == 0] = 1 (quant.py:1062): an fp16 group scale below 2^-24 is 0
        t = rnd(rnd(w2d / s, dt) + np.float32(0.0), dt)   # tensor / scales + zeros (0.0): -0 becomes +0
    bits, _ = fp8_encode(t, fmt, sem)           # float_quantize(.float(), e, m) then .to(float8 type)
    return bits, s, sdt


def fp8_fake(w2d, dt, fmt='e4m3', sem='cast'):
    """fake_quant_weight_dynamic: (q - 0) * s is an fp32 product, then .to(dt)  (quant.py:1074-1080, 1165-1176)."""
    fmax = np.float32(FP8_FORMATS[fmt][2])
    mx = w2d.max(axis=-1, keepdims=True)
    mn = w2d.min(axis=-1, keepdims=True)
    a = np.maximum(np.maximum(np.abs(mx), np.abs(mn)), rnd(np.float32(1e-5), dt))
    sdt = F32 if w2d.shape[0] == 1 else dt
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        s = rnd(a / fmax, sdt)
        s = np.where(s == 0, np.float32(1.0), s).astype(np.float32)      # scales[scales == 0] = 1 (quant.py:1062): an fp16 group scale below 2^-24 is 0
        t = rnd(rnd(w2d / s, dt) + np.float32(0.0), dt)
    _, v = fp8_encode(t, fmt, sem)
    return rnd(v * s, dt)


def mse_range(x, sym, qmin, qmax, round_zp=True, maxshrink=0.8, grid=100, norm=2.4):
    """quant.py:145-203 (get_mse_range) on the fp32 [G, g] view `x` (the reference casts to float first).
    NOTE the reference's aliasing: `best_min_val` IS `_min_val` (a view of `min_val`), so a row that improves at step i
    continues the search from its already shrunk range: xmin_i = p_i * current_min, not p_i * original_min.
    Returns (min_val, max_val) fp32 [G]. |q - x| ** norm and the row sum are fp32 like torch's; their last bits depend on
    the pow / summation implementation, so the argmin can differ from the reference on near-ties only."""
    x = np.asarray(x, dtype=np.float32)
    cur_min = x.min(axis=1).astype(np.float32)
    cur_max = x.max(axis=1).astype(np.float32)
    best = np.full(x.shape[0], np.inf, dtype=np.float32)
    for i in range(int(maxshrink * grid)):
        p = np.float32(1 - i / grid)           # python float -> fp32 scalar of a fp32 tensor op
        xmin = (p * cur_min).astype(np.float32)
        xmax = (p * cur_max).astype(np.float32)
        s, z = qparams_from_minmax(xmin, xmax, F32, sym, qmin, qmax, round_zp)
        codes, p2 = quant_codes(x, F32, s[:, None], F32, z[:, None], F32, qmin, qmax)
        q = dequant(codes, s[:, None], z[:, None], p2)
        d = np.abs((q - x).astype(np.float32))
        err = np.power(d, np.float32(norm), dtype=np.float32).sum(axis=1, dtype=np.float32)
        better = err < best
        best = np.where(better, err, best)
        cur_min = np.where(better, xmin, cur_min)
        cur_max = np.where(better, xmax, cur_max)
    return cur_min, cur_max


# ---- FP8 block-wise (kernel.py:7-55, 146-242): restated from the Triton source, which cannot run here (no CUDA /
# Triton backend): PARITY UNPINNED for these two functions — they are checked against this restatement only.
def act_quant_ref(x, block=128):
    """kernel.py:7-55. x float array [..., K]; returns (e4m3 bytes as uint8, fp32 scales [..., K / block])."""
    import torch
    xf = np.asarray(x, dtype=np.float32)
    shp = xf.shape
    xb = xf.reshape(-1, block)
    s = (np.abs(xb).max(axis=1) / np.float32(448.0)).astype(np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        y = (xb / s[:, None]).astype(np.float32)
    bits = torch.from_numpy(y).to(torch.float8_e4m3fn).view(torch.uint8).numpy().reshape(shp)
    return bits, s.reshape(*shp[:-1], shp[-1] // block)


def fp8_block_gemm_ref(a_bits, a_s, b_bits, b_s, block=128):
    """kernel.py:146-242: sum over K blocks of (A_kb . B_kb^T) * a_s[m, kb] * b_s[n / block, kb] in fp32 (fp64
    accumulation inside a block: the products are exact, only the summation order is the kernel's own)."""
    import torch
    A = torch.from_numpy(np.ascontiguousarray(a_bits)).view(torch.float8_e4m3fn).float().numpy()
    B = torch.from_numpy(np.ascontiguousarray(b_bits)).view(torch.float8_e4m3fn).float().numpy()
    M, K = A.shape
    N = B.shape[0]
    out = np.zeros((M, N), dtype=np.float32)
    for kb in range((K + block - 1) // block):
        sl = slice(kb * block, min(K, (kb + 1) * block))
        part = (A[:, sl].astype(np.float64) @ B[:, sl].astype(np.float64).T).astype(np.float32)
        t = (part * a_s[:, kb][:, None]).astype(np.float32)
        t = (t * np.repeat(b_s[:, kb], block)[:N][None, :]).astype(np.float32)
        out = (out + t).astype(np.float32)
    return out


# ---- per_tensor asymmetric (quant.py:132-136, 555-556): 0-dim min/max of the tensor dtype against the 0-dim fp32
# (qmax - qmin) promote scales / zeros to fp32; in the ops against the dimensioned tensor the 0-dim qparams keep their
# fp32 VALUE but do not promote the result, which is rounded to the tensor dtype after every op.
def per_tensor_asym_qparams(w, dt, qmin, qmax):
    mn, mx = np.float32(w.min()), np.float32(w.max())
    d = np.maximum(rnd(mx - mn, dt), rnd(np.float32(1e-5), dt))
    s = np.float32(d) / np.float32(qmax - qmin)
    z = np.float32(qmin) - np.rint(np.float32(mn) / s)
    z = np.minimum(np.maximum(z, np.float32(qmin)), np.float32(qmax))
    return np.float32(s), np.float32(z)


def per_tensor_asym_fake_and_codes(w, dt, qmin, qmax):
    s, z = per_tensor_asym_qparams(w, dt, qmin, qmax)
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        t = rnd(w / s, dt)
        t = rnd(np.rint(t), dt)
        t = rnd(t + z, dt)
        codes = np.minimum(np.maximum(t, np.float32(qmin)), np.float32(qmax))
        fake = rnd(rnd(codes - z, dt) * s, dt)
    return fake.astype(np.float32), codes.astype(np.int32), s, z


# ---- FP8 e4m3 per_block (quant.py:132-143, 612-658, 1043-1072): b x b tiles, fp32 scale = max(absmax, 1e-5) / 448,
# q = e4m3(x / scale) in fp32 (the dimensioned fp32 scale promotes the 16-bit tensor), fake = q * scale rounded once.
def fp8_per_block(w, dt, block, sem='cast'):
    w = np.asarray(w, dtype=np.float32)
    M, N = w.shape
    mb, nb = -(-M // block), -(-N // block)
    bits = np.zeros((M, N), dtype=np.uint8)
    fake = np.zeros((M, N), dtype=np.float32)
    scales = np.zeros((mb, nb), dtype=np.float32)
    for i in range(mb):
        for j in range(nb):
            sl = (slice(i * block, min(M, (i + 1) * block)), slice(j * block, min(N, (j + 1) * block)))
            blk = w[sl]
            s = np.float32(max(np.abs(blk).max(), np.float32(1e-5))) / np.float32(448.0)
            scales[i, j] = s
            y = (blk / s + np.float32(0.0)).astype(np.float32)
            b, v = fp8_encode(y, 'e4m3', sem)
            bits[sl] = b
            fake[sl] = rnd((v * s).astype(np.float32), dt)
    return bits, scales, fake
