"""ATen's CPU summation orders, restated (test infrastructure; parity pinned by tests/test_oracle_golden.py against
torch itself on this container's AVX-512 build and against the reference goldens).

aten/src/ATen/native/cpu/SumKernel.cpp (torch 2.x): `cascade_sum` with
  * multi_row_sum: rows added sequentially in chunks of `level_step` (16 for n < 2^20) into acc[0], chunk sums cascaded
    into acc[1..3], everything added in level order at the end;
  * row_sum: four interleaved streams (i % 4) each a multi_row_sum, then ((p0 + p1) + p2) + p3;
  * vectorized_inner_sum for a contiguous reduced dimension of a 16-bit tensor: a Vectorized<BFloat16 / Half> is loaded as
    lo + hi in fp32 and row_sum runs over those vectors, the lane sums are then added sequentially, trailing elements first.
    sum_stub is registered WITHOUT the AVX-512 variant (REGISTER_DISPATCH only), so even on AVX-512 hosts the kernel is
    the AVX2 one: V = 8 fp32 lanes, 16 elements per 16-bit vector (measured: V = 8 reproduces torch on 2e5 fp16 rows,
    V = 16 does not);
  * vectorized_outer_sum / scalar_outer_sum for a strided reduced dimension: columns are walked 4 V = 32 / V = 8 at a time
    (4 at a time when there are fewer than V): the first kind of block uses multi_row_sum, the remainders row_sum.
16-bit `mean` (ReduceOps.cpp): cast to fp32, sum, divide, cast back (one rounding). Valid for the SERIAL iterator (small
inputs); above TensorIterator's grain size the reduced dimension may be split across threads."""
import numpy as np

f32 = np.float32


def _ceil_log2(n):
    n = int(n)
    return 0 if n <= 1 else (n - 1).bit_length()


def multi_row_sum(rows):
    """rows [size, ...] fp32 -> sum over axis 0 in ATen's cascade order."""
    size = rows.shape[0]
    num_levels = 4
    level_power = max(4, _ceil_log2(size) // num_levels)
    level_step = 1 << level_power
    level_mask = level_step - 1
    acc = [np.zeros(rows.shape[1:], f32) for _ in range(num_levels)]
    i = 0
    while i + level_step <= size:
        for _ in range(level_step):
            acc[0] = (acc[0] + rows[i]).astype(f32)
            i += 1
        for j in range(1, num_levels):
            acc[j] = (acc[j] + acc[j - 1]).astype(f32)
            acc[j - 1] = np.zeros_like(acc[j - 1])
            if (i & (level_mask << (j * level_power))) != 0:
                break
    while i < size:
        acc[0] = (acc[0] + rows[i]).astype(f32)
        i += 1
    for j in range(1, num_levels):
        acc[0] = (acc[0] + acc[j]).astype(f32)
    return acc[0]


def row_sum(rows):
    size, ilp = rows.shape[0], 4
    size_ilp = size // ilp
    if size_ilp > 0:
        part = multi_row_sum(rows[:size_ilp * ilp].reshape(size_ilp, ilp, *rows.shape[1:]))
        part = [part[k].copy() for k in range(ilp)]
    else:
        part = [np.zeros(rows.shape[1:], f32) for _ in range(ilp)]
    for i in range(size_ilp * ilp, size):
        part[0] = (part[0] + rows[i]).astype(f32)
    for k in range(1, ilp):
        part[0] = (part[0] + part[k]).astype(f32)
    return part[0]


def inner_sum_16bit(x, V=8):
    """x [..., n]: values of a 16-bit dtype held as fp32; contiguous inner reduction -> fp32 [...] (before the final
    rounding to the tensor dtype). n >= 2 V (one Vectorized<16-bit>)."""
    n = x.shape[-1]
    vb = 2 * V
    if n < vb:
        raise NotImplementedError('inner sums shorter than one vector take ATen\'s scalar path')
    lead = x.shape[:-1]
    xf = np.ascontiguousarray(x, dtype=f32).reshape(-1, n)
    nv = n // vb
    vecs = xf[:, :nv * vb].reshape(-1, nv, 2, V)
    loads = (vecs[:, :, 0, :] + vecs[:, :, 1, :]).astype(f32)          # [rows, nv, V]
    acc = row_sum(np.moveaxis(loads, 1, 0))                            # -> [rows, V]
    fin = np.zeros(xf.shape[0], f32)
    for k in range(nv * vb, n):
        fin = (fin + xf[:, k]).astype(f32)
    for lane in range(V):
        fin = (fin + acc[:, lane]).astype(f32)
    return fin.reshape(lead)


def outer_sum_fp32(t, V=8):
    """t [oc, tok, ng] fp32 -> sum over tok, fp32 [oc, ng], serial iterator order."""
    oc, tok, ng = t.shape
    out = np.empty((oc, ng), f32)
    j = 0
    if ng >= V:
        while j + 4 * V <= ng:
            out[:, j:j + 4 * V] = multi_row_sum(np.moveaxis(t[:, :, j:j + 4 * V], 1, 0))
            j += 4 * V
        while j + V <= ng:
            out[:, j:j + V] = row_sum(np.moveaxis(t[:, :, j:j + V], 1, 0))
            j += V
    else:
        while j + 3 < ng:
            out[:, j:j + 4] = multi_row_sum(np.moveaxis(t[:, :, j:j + 4], 1, 0))
            j += 4
    while j < ng:
        out[:, j] = row_sum(np.moveaxis(t[:, :, j], 1, 0))
        j += 1
    return out
