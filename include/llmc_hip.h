/*
 * llmc_hip.h — C ABI of libllmc_hip.so: the MI355X (gfx950) implementation of llmc's per-Linear
 * weight-quantization hot path (GPTQ / AWQ / RTN quantizer arithmetic, packing).
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - the library never allocates or frees device memory: callers pass outputs and a workspace whose
 *     size comes from the matching `*_ws_bytes` query;
 *   - every entry point takes the hipStream_t to launch on (as void*) and never synchronises the device;
 *   - return value: 0 ok; LLMC_EINVAL bad shape/dtype/alignment; LLMC_ENOTSUP unsupported combination;
 *     LLMC_EIO a HIP runtime error (text via llmc_hip_last_error). No C++ exception crosses the ABI;
 *   - dtype codes: LLMC_F16 / LLMC_BF16 / LLMC_F32. "Tensor dtype" arithmetic follows ATen's rule the
 *     reference relies on: every elementwise op is evaluated in fp32 and rounded (RNE) to the op's
 *     result dtype, no FMA contraction, IEEE division.
 *
 * Each entry cites the reference code (paths relative to the llmc tree) it replaces.
 */
#ifndef LLMC_HIP_H_
#define LLMC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLMC_HIP_ABI_VERSION 1

#define LLMC_OK 0
#define LLMC_EINVAL (-22)
#define LLMC_ENOTSUP (-95)
#define LLMC_EIO (-5)

#define LLMC_F16 0
#define LLMC_BF16 1
#define LLMC_F32 2
/* OR-ed into the dtype code of llmc_quant_static's scales / zeros: the operand is a 0-dim tensor in the reference
 * (per_tensor qparams): it is used at its own precision but ATen leaves it out of type promotion, so every op still
 * rounds to the weight dtype (quant.py:699-717 with the 0-dim scales of quant.py:132-136,555-556). */
#define LLMC_SCALAR_QPARAM 16
/* OR-ed into llmc_quant_static's zdt: round_zp = False (quant.py:557-558, 702-707): the zero point is not an integer and
 * the code is clamp(round(x / max(s, 1e-9) + z)) instead of clamp(round(x / s) + z). */
#define LLMC_FRACTIONAL_ZP 32

/* integer code container for llmc_quant_static / llmc_quant_dynamic */
#define LLMC_OUT_FAKE 0 /* dequantised values, written in the weight dtype              */
#define LLMC_OUT_I32 1  /* int32 codes   (reference: bit != 8  -> torch.int32)          */
#define LLMC_OUT_I8 2   /* int8 codes    (reference: bit == 8, qmin != 0 -> torch.int8) */
#define LLMC_OUT_U8 3   /* uint8 codes   (reference: bit == 8, qmin == 0 -> torch.uint8)*/

typedef void* llmc_stream_t; /* hipStream_t */

int llmc_hip_abi_version(void);
/* hash (16 hex digits) of the sources and compiler flags the library was built from; the Python loader recomputes it from
 * the sources next to the .so and refuses a stale library (llmc_amd/build.py:source_digest, llmc_amd/_ffi.py:lib) */
const char* llmc_hip_build_id(void);
/* copies the last HIP error string of the calling thread into buf (NUL-terminated); returns its length */
int llmc_hip_last_error(char* buf_host, size_t n);

/* Per calling-thread switch: may llmc_chol_inv_upper / llmc_gptq_quantize use an internal low-priority helper stream
 * to overlap their large trailing updates with their latency-bound chains (default 1)? A caller that already runs
 * several of them side by side on its own streams (the subsets of a block) turns it off for all but the longest
 * chain: every entry point still completes, in stream order, on the stream it was given. Returns the previous value.
 * No reference counterpart (the reference runs one default stream). */
int llmc_hip_set_helper_streams(int enable);

/* Per calling-thread: the Hessian kernel (llmc_hessian_accum*) is persistent, one workgroup per compute unit, and a
 * workgroup owns its CU (all VGPRs, 128 KiB LDS). n_cus > 0 makes it leave that many CUs (rounded up to a multiple of
 * 8, one per XCD) to kernels the caller runs on other streams meanwhile; 0 (default) = the whole device. Returns the
 * previous value. No reference counterpart. */
int llmc_hip_set_cu_reserve(int n_cus);

/* Explicit A/B switches, per calling thread, all 0 by default. The library reads NO environment variable: what used to be
 * LLMC_* variables (rounds 2-5) are these keys. Every value produces valid results; "same bits" = bit-identical to the default.
 *   k3_fp32           K3's large products on the fp32 MFMA pipe instead of split-bf16 (other rounding, same accuracy class)
 *   k3_no_gemm6       deep levels of the triangular inverse on k_gemm3 instead of gemm6 (other rounding)
 *   k3_no_planes      K3's far updates split inside every tile (k_gemm3) instead of pre-split planes + k_gemm3s; same bits
 *   k3_split_far      without a helper stream, an outer block's far update as two launches instead of one; same bits
 *   k4_split_far      without helper streams, a column group's far update as three launches instead of one; same bits
 *   k4_err_rowmajor   the column loop's error columns row-major instead of k-major; same bits
 *   gptq_generic      k_gptq_block: every wave on the generic IEEE-division path; same bits
 *   gemm3_nospec      never k_gemm3s; same bits.   gemm3s_min_tiles = n > 0: its tile-count threshold (tests)
 *   no_shortk         short products through the general fp32 GEMM; same bits
 *   linear_nosplit    the k-tiled GEMM never cuts a small product into k-slices (single pass: the row-major kernel's bits)
 *   fp8_exact_div     FP8 casts through the IEEE division + general encoder; same bits
 *   side_cu_mask      helper streams created with a CU mask (read when a caller stream's helper set is first created)
 *   k1_batch_off      llmc_hessian_accum_multi as one launch per problem instead of one tile queue; same bits
 *   k1_fp32_diag      diag(H) as the MFMA kernel's fp32 chain leaves it instead of the fp64-folded one (other diagonal)
 *   fp8_no_packed16   FP8 e4m3 cast of bf16 tensors (qtorch rounding, codes out): the float form of the division-free path; same bits
 *   gemm3s_no_dma     k_gemm3s (planes form): producer waves copy through registers instead of LDS-DMA; same bits
 *   gemm3_no_wide     K3's far updates on k_gemm3s instead of k_gemm3w (128 x 128 tiles, two workgroups per CU); same bits
 *   sgemm_no_wide     K4's far update: 1 = on k_sgemm (the kernel of rounds 2-5), 4 = k_sgemm_wide's 256 x 128 form (one workgroup
 *                     per CU) instead of its 128 x 128 form (two per CU, the default); same bits
 * set: returns the previous value, or LLMC_EINVAL for an unknown key / negative value. get: the value, or LLMC_EINVAL.
 * option_name: the key of index 0, 1, ... (copied into buf), LLMC_EINVAL past the last one. No reference counterpart. */
int llmc_hip_set_option(const char* key, int value);
int llmc_hip_get_option(const char* key);
int llmc_hip_option_name(int index, char* buf_host, size_t n);

/* ------------------------------------------------------------------------------------------------
 * Quantizer arithmetic (llmc/compression/quantization/quant.py)
 * ---------------------------------------------------------------------------------------------- */

/* get_minmax_range + get_qparams (quant.py:132-143, 545-559) on a [G, g] view (reshape_tensor,
 * quant.py:612-642: per_group -> g = group_size, per_channel -> g = in_features, per_tensor -> G = 1).
 * scales / zeros: [G] in the tensor dtype `dt`. zeros may be NULL when sym (reference returns 0).
 * round_zp = 1 is the reference default; 0 gives zeros = qmin - min/scale. */
size_t llmc_minmax_qparams_ws_bytes(int64_t G, int64_t g);
int llmc_minmax_qparams(const void* W, int dt, int64_t G, int64_t g, int sym, int round_zp,
                        float qmin, float qmax, void* scales, void* zeros, void* ws,
                        llmc_stream_t stream);

/* calib_algo 'static_hist', data pass (quant.py:462-512): torch.histc(sample.float(), bins, min, max) -> out fp32 [bins]
 * counts; bin = int((x - min) * bins / (max - min)) in fp32, the right edge belongs to the last bin, values outside the
 * range are dropped, min == max widens the range by one on both sides (ATen). ws: llmc_histc_ws_bytes(bins). The merge
 * of the per-sample histograms and the range search (quant.py:279-460) work on `bins` numbers and stay host code. */
/* Per-sample min / max of n <= llmc_minmax_samples_max() calibration samples in one launch pair: what `sample.min()`,
 * `sample.max()` give for every sample in get_minmax_stats / get_moving_minmax / the first pass of the histogram observer
 * (quant.py:253-263, 524-543, 462-475). X_list_host[i]: device address of sample i (16-B aligned, its own allocation),
 * len_list_host[i] its element count; both arrays live on the HOST (they travel in the kernel arguments). mn / mx: fp32 [n]
 * on the device (the values are elements of the samples, exact in fp32); a NaN in a sample makes both of its results NaN,
 * like torch. ws: llmc_minmax_samples_ws_bytes(len_list_host, n). */
int llmc_minmax_samples_max(void);
size_t llmc_minmax_samples_ws_bytes(const int64_t* len_list_host, int n);
int llmc_minmax_samples(const void* const* X_list_host, const int64_t* len_list_host, int n, int dt, float* mn, float* mx,
                        void* ws, llmc_stream_t stream);

size_t llmc_histc_ws_bytes(int bins);
int llmc_histc(const void* x, int dt, int64_t n, int bins, float min, float max, float* out, void* ws,
               llmc_stream_t stream);

/* calib_algo 'mse': BaseQuantizer.get_mse_range (quant.py:145-203) followed by get_qparams (:545-559) on a [G, g]
 * view. The reference casts the tensor to fp32 first, so ranges and qparams are fp32 whatever `dt` is:
 * scales / zeros / min_out / max_out are fp32 [G] (zeros may be NULL when sym, min_out / max_out may be NULL).
 * nsteps = int(maxshrink * mse_grid) and grid = mse_grid as Python computes them; norm = 2.4 in the reference. */
int llmc_mse_qparams(const void* W, int dt, int64_t G, int64_t g, int sym, int round_zp, float qmin, float qmax,
                     int nsteps, int grid, float norm, float* scales, float* zeros, float* min_out, float* max_out,
                     llmc_stream_t stream);

/* IntegerQuantizer.quant / quant_dequant with given qparams (quant.py:699-717), i.e. the arithmetic of
 * fake_quant_weight_static (quant.py:785-831) and real_quant_weight_static (quant.py:871-914).
 * W: [G, g] dtype wdt. scales [G] dtype sdt. zeros [G] dtype zdt, or NULL (== 0, the symmetric case).
 * out_kind LLMC_OUT_FAKE writes (q - z) * s cast to wdt; the others write integer codes.
 * Promotion follows torch: x/s in promote(wdt,sdt); +z in promote(.,zdt); rounding after each op. */
int llmc_quant_static(const void* W, int wdt, int64_t G, int64_t g, const void* scales, int sdt,
                      const void* zeros, int zdt, float qmin, float qmax, int out_kind, void* out,
                      llmc_stream_t stream);

/* fake_quant_weight_dynamic (quant.py:833-869) / real_quant_weight_dynamic (quant.py:916-953):
 * min/max -> qparams -> quant(-> dequant) in one pass over W. scales_out / zeros_out may be NULL
 * (fake path does not keep them); when given they are [G] in dtype dt. */
size_t llmc_quant_dynamic_ws_bytes(int64_t G, int64_t g);
int llmc_quant_dynamic(const void* W, int dt, int64_t G, int64_t g, int sym, int round_zp, float qmin,
                       float qmax, int out_kind, void* out, void* scales_out, void* zeros_out, void* ws,
                       llmc_stream_t stream);

/* VllmRealQuantLinear.pack (module_utils.py:836-862): codes [R, K] (int32 or int8 container) ->
 * packed int32 [R, ceil(K / (32/bits))]; u = (code + 2^(bits-1)) & 0xff, LSB-first nibbles/bytes,
 * zero padding on the right. code_kind is LLMC_OUT_I32 or LLMC_OUT_I8. */
int llmc_pack_lsb(const void* codes, int code_kind, int64_t R, int64_t K, int bits, int32_t* packed,
                  llmc_stream_t stream);

/* AutoawqRealQuantLinear.gemm_pack (module_utils.py:1004-1065): recompute codes
 * round((w + z*s) / s) in fp16 from weight [R, K] (f16), scales/zeros [R, K/g] as returned by
 * real_quant_weight_* (scales any float dtype -> cast to f16, zeros int32), transpose to [K, R] and
 * pack 8 nibbles per int32 along R with AWQ's order map [0,2,4,6,1,3,5,7].
 * qweight [K, R/8] int32, qzeros [K/g, R/8] int32, scales_out [K/g, R] f16. */
int llmc_pack_awq_gemm(const void* weight, int wdt, const void* scales, int sdt, const int32_t* zeros,
                       int64_t R, int64_t K, int64_t g, int32_t* qweight, int32_t* qzeros,
                       void* scales_out_f16, llmc_stream_t stream);

/* FloatQuantizer weight / activation path, e4m3 and e5m2 (quant.py:963-1003, 1043-1072, 1161-1221):
 * scale = absmax.clamp(1e-5) / finfo(float8 type).max (448 / 57344) per row of the [G, g] view (G = 1 per-tensor, G = R
 * per-channel / per-token), q = float_quantize(x / scale, e_bits, m_bits, 'nearest'). `fake` is a flag word:
 *   bit 0      write dequantised q * scale in dt to `out` instead of 8-bit codes
 *   bits 4-5   format: 0 = e4m3, 1 = e5m2 (quant.py:983-984, 1162-1163)
 *   bit 8      rounding semantics: 1 = qtorch.quant.float_quantize as llmc calls it (third-party, not vendored by the
 *              reference: QPyTorch 0.3.0's published CPU algorithm restated in csrc/fp8_math.h and oracle/quant_ref.py —
 *              IEEE-style formats, ties away from zero, saturation at 240 / 57344); 0 = torch's dtype cast
 *              (.to(torch.float8_e4m3fn / float8_e5m2): round to nearest even, OCP e4m3fn up to 448), which is what
 *              the reference's real-quant path ends in (quant.py:1183, 1211) and what its Triton kernels compute.
 *   bit 9      evaluate every element with the IEEE division and the general encoder. Results are identical with and
 *              without it: 16-bit e4m3 calls otherwise use w * fl(1 / scale) and send only the lanes near a rounding
 *              boundary (or outside the 8-bit format's normal range) through the division (tests/test_fp8_fast_gpu.py).
 * Codes are OCP e4m3fn / IEEE e5m2 bytes in both cases (every qtorch result is representable). scales [G] in dtype sdt:
 * ATen yields fp32 for the 0-dim per-tensor scale and the tensor dtype for per-channel.
 * static_scales != 0: `scales` is INPUT (fake_quant_act_static / real_quant_weight_static, quant.py:1083-1099). */
size_t llmc_fp8_quant_ws_bytes(int64_t G, int64_t g);
int llmc_fp8_quant(const void* W, int dt, int64_t G, int64_t g, int fake, void* out, void* scales, int sdt,
                   int static_scales, void* ws, llmc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GPTQ (llmc/compression/quantization/gptq.py)
 * ---------------------------------------------------------------------------------------------- */

/* GPTQ.add_batch (gptq.py:254-295):  H <- H * (n_before / n_after) + (2 / n_after) * X^T X
 * X: [T, K] tokens x channels (row stride ldx elements), 16-bit dtype dt (F16 / BF16); H: [K, K] fp32,
 * full symmetric on output. n_before / n_after are the reference's `nsamples` before and after the call
 * (counts of sequences, not tokens). Products are exact (16-bit x 16-bit in fp32), accumulation is fp32
 * on the MFMA pipe in a fixed, launch-independent order (deterministic). */
size_t llmc_hessian_accum_ws_bytes(int64_t T, int64_t K, int64_t ldx);
int llmc_hessian_accum(float* H, const void* X, int dt, int64_t T, int64_t K, int64_t ldx,
                       double n_before, double n_after, void* ws, llmc_stream_t stream);
/* The two launches of llmc_hessian_accum, separately (same ws): the MFMA kernel that writes per-unit
 * partial tiles, and the ordered reduction that folds them into H. Used by bench.py to time the MFMA
 * kernel alone with HIP events, and by callers that overlap the reduction with other work. */
int llmc_hessian_accum_partials(const void* X, int dt, int64_t T, int64_t K, int64_t ldx, void* ws,
                                llmc_stream_t stream);
int llmc_hessian_accum_reduce(float* H, int64_t T, int64_t K, int64_t ldx, double n_before, double n_after,
                              const void* ws, llmc_stream_t stream);

/* The same update from a LIST of samples that stay where they are (no staging copy): llmc's hooks call add_batch once
 * per calibration sample (calib.bs = 1, configs/quantization/methods/GPTQ/gptq_w_only.yml:12 -> 128 calls of
 * [1, 2048, K] per layer, gptq.py:254-295); a caller that keeps those tensors resident passes their addresses here and
 * gets ONE launch and ONE reduction for all of them:  X^T X = sum_i X_i^T X_i,  n_after - n_before = the number of
 * sequences in the list. X_list_host / T_list_host: HOST arrays of n device pointers (16-B aligned rows, common row
 * stride ldx and dtype dt) and their token counts (> 0; any lengths: every sample is walked in 128-token groups and
 * the last group of a sample is zero-filled by its buffer descriptor). n <= llmc_hessian_max_samples(). The host arrays
 * are consumed before the call returns (they travel in the kernel arguments: up to 8 KiB of them). The result is bit-identical to
 * llmc_hessian_accum on the concatenation of the samples whenever every T_i is a multiple of 128. ws from
 * llmc_hessian_accum_ptrs_ws_bytes (0 = invalid arguments). */
int llmc_hessian_max_samples(void);
size_t llmc_hessian_accum_ptrs_ws_bytes(const int64_t* T_list_host, int n, int64_t K, int64_t ldx);
int llmc_hessian_accum_ptrs(float* H, const void* const* X_list_host, const int64_t* T_list_host, int n, int dt,
                            int64_t K, int64_t ldx, double n_before, double n_after, void* ws, llmc_stream_t stream);
int llmc_hessian_accum_ptrs_partials(const void* const* X_list_host, const int64_t* T_list_host, int n, int dt,
                                     int64_t K, int64_t ldx, void* ws, llmc_stream_t stream);
int llmc_hessian_accum_ptrs_reduce(float* H, const int64_t* T_list_host, int n, int64_t K, int64_t ldx,
                                   double n_before, double n_after, const void* ws, llmc_stream_t stream);
/* Several Hessians in ONE launch pair (one unit queue): the problems' triangular tile grids fill the persistent grid's
 * rounds together — the three K = 4096 inputs of a Llama block pay one partially filled last round instead of three — and
 * the exact diagonal below. A problem = one llmc_hessian_accum_ptrs call: H [K, K], its sample list (host arrays of n device
 * pointers / token counts, consumed before the call returns), K, the common row stride ldx, the running-mean weights.
 * Up to llmc_hessian_max_problems() problems and llmc_hessian_max_samples() samples (all problems together) per call; one
 * dtype per call. The one-problem entry points above are wrappers of these (dstate = NULL).
 *
 * Exact diagonal (always on; no extra pass over the samples): in a DIAGONAL 256 x 256 tile of the kernel the upper-right
 * quadrant is the transpose of the lower-left one, so the reduction mirrors that one and the quadrant's wave spends its
 * MFMAs on the eight 32 x 32 blocks ON the diagonal, restarting its accumulators every 2048 tokens (every 256 in short units)
 * and folding each block's diagonal into fp64. diag(H) — what actorder sorts (gptq.py:58-83) and the damping averages
 * (gptq.py:169) — then sits within 2e-7 of the exact value (an fp32 rounding is 6e-8), instead of the 2-3e-6 an fp32 chain over a
 * whole token chunk leaves (twice the reference's sgemm; rounds 1-5 offered a second fp64 pass over the samples for this). dstate [K] fp64 (may
 * be NULL) carries the unrounded running diagonal across calls (overwritten when n_before == 0); without it H's own fp32
 * diagonal is the carried value. llmc_hip_set_option("k1_fp32_diag", 1) keeps the MFMA kernel's fp32 diagonal (A/B). */
typedef struct {
    float* H;
    double* dstate;
    const void* const* X_list_host;
    const int64_t* T_list_host;
    int n;
    int64_t K;
    int64_t ldx;
    double n_before;
    double n_after;
} llmc_hessian_problem_t;
int llmc_hessian_max_problems(void);
size_t llmc_hessian_accum_multi_ws_bytes(const llmc_hessian_problem_t* probs_host, int P);
int llmc_hessian_accum_multi_partials(const llmc_hessian_problem_t* probs_host, int P, int dt, void* ws, llmc_stream_t stream);
int llmc_hessian_accum_multi_reduce(const llmc_hessian_problem_t* probs_host, int P, const void* ws, llmc_stream_t stream);
int llmc_hessian_accum_multi_barrier_timeouts(const llmc_hessian_problem_t* probs_host, int P, const void* ws, unsigned* out_host,
                                              llmc_stream_t stream);
/* Diagnostic: how many round barriers of the LAST llmc_hessian_accum*_partials launch on `ws` (same T list, K, ldx) gave
 * up waiting because workgroups of its persistent grid were kept off their CUs by other streams. Synchronises `stream`
 * and writes the count to *out_host. 0 in a healthy run; > 0 leaves the result correct but the launch slower. */
int llmc_hessian_accum_barrier_timeouts(const void* ws, const int64_t* T_list_host, int n, int64_t K, int64_t ldx,
                                        unsigned* out_host, llmc_stream_t stream);

/* GPTQ.process_hessian_and_weights, first half (gptq.py:135-152, 169-171):
 *   dead = diag(H) == 0 -> H[dead,dead] = 1, W[:,dead] = 0; optional symmetric gather by perm
 *   (Hout = H[perm][:,perm], Wout = W[:,perm]); damp = percdamp * mean(diag) ; Hout += damp * I.
 * W [R, K] dtype wdt (model dtype or f32) -> Wout [R, K] fp32. perm may be NULL (identity).
 * H is modified in place only for the dead-column fix; Hout must not alias H.
 * Hout may be NULL (only W is gathered: layers that share an input share the factor), and W/Wout may
 * both be NULL (only H is prepared). */
size_t llmc_hessian_prep_ws_bytes(int64_t K);
int llmc_hessian_prep(float* H, const void* W, int wdt, int64_t R, int64_t K, const int64_t* perm,
                      float percdamp, float* Hout, float* Wout, void* ws, llmc_stream_t stream);

/* `W = tmp[:, invperm]` after the column loop (gptq.py:186-188) and any other fp32 column gather:
 * out[r][j] = in[r][idx[j]], in/out [R, K] fp32 contiguous, K % 4 == 0, K <= 40960 (one row staged in LDS), out must not alias in. */
int llmc_gather_cols(const float* in, int64_t R, int64_t K, const int64_t* idx, float* out, llmc_stream_t stream);

/* gptq.py:172-174: cholesky -> cholesky_inverse -> cholesky(upper). Computes the same upper factor U
 * (H^-1 = U^T U) by one reverse Cholesky H = R R^T (R upper) and one triangular inverse U = R^-1, in
 * fp32. A [K, K] is overwritten by U (strict lower triangle zeroed). info_dev: int32, 0 on success,
 * i+1 if the leading minor i is not positive definite. */
size_t llmc_chol_inv_upper_ws_bytes(int64_t K);
int llmc_chol_inv_upper(float* A, int64_t K, void* ws, int32_t* info_dev, llmc_stream_t stream);
/* The two calls above without the transposing pass between them (K^2 floats read + written; 0.5 ms at K = 14336):
 * llmc_hessian_prep_rev writes Hout index-reversed (Hout[i][j] = Hp[K-1-i][K-1-j]: the same gather with the permutation read
 * backwards; workspace llmc_hessian_prep_ws_bytes), which for a symmetric H is Hp reflected across its anti-diagonal — the
 * matrix llmc_chol_inv_upper builds first. llmc_chol_inv_upper_rev factors it in place (Arev is destroyed) and writes U to
 * Uout (a separate buffer; workspace llmc_chol_inv_upper_ws_bytes). U is bit-identical to the two-call form. */
int llmc_hessian_prep_rev(float* H, const void* W, int wdt, int64_t R, int64_t K, const int64_t* perm,
                          float percdamp, float* Hout, float* Wout, void* ws, llmc_stream_t stream);
int llmc_chol_inv_upper_rev(float* Arev, float* Uout, int64_t K, void* ws, int32_t* info_dev, llmc_stream_t stream);

/* GPTQ.weight_transform (gptq.py:199-244), the blocked column loop, with the quantizer of
 * search_column_qparams (gptq.py:359-366) fused at group starts.
 *   W      [R, K] fp32, (already permuted / dead-fixed), overwritten with the running updated weights
 *   Hinv   [K, K] fp32 upper factor U
 *   Wout   [R, K] fp32 = `tmp`: the error-compensated weight of each column at the time it was visited
 *   losses [R, K] fp32 = `Losses` (may be NULL)
 *   group_size: 0 = per_channel (one qparam pair per row, given in scales/zeros), else group size
 *   static_groups = 0: qparams re-derived from the current W at each group start and written to
 *     scales/zeros [R, K/group_size] (fp32) in processing (permuted) order;
 *   static_groups = 1: scales/zeros [R, K/group_size] are INPUT in original column order and
 *     col_group[i] (int32 [K], = perm[i]/group_size) selects the group of processed column i.
 *   sym: zeros ignored on input when static and sym (zero == 0).
 * blocksize must be 128 (the reference default in every shipped GPTQ config). */
size_t llmc_gptq_quantize_ws_bytes(int64_t R, int64_t K);
int llmc_gptq_quantize(float* W, const float* Hinv, int64_t R, int64_t K, int sym, float qmin,
                       float qmax, int64_t group_size, int static_groups, const int32_t* col_group,
                       float* scales, float* zeros, float* Wout, float* losses, int blocksize,
                       void* ws, llmc_stream_t stream);
/* OWQ form of the same loop (gptq.py:44-56, 66-83, 199-244 with n_nonout < columns): only the first n_quant columns are
 * visited; the trailing K - n_quant columns (kept in floating point) still receive every block's error feedback.
 * Dynamic groups are clipped at n_quant like `min(i + group_size, columns - n_out)` (gptq.py:216-221). */
int llmc_gptq_quantize_cols(float* W, const float* Hinv, int64_t R, int64_t K, int64_t n_quant, int sym, float qmin,
                       float qmax, int64_t group_size, int static_groups, const int32_t* col_group,
                       float* scales, float* zeros, float* Wout, float* losses, int blocksize,
                       void* ws, llmc_stream_t stream);

/* SpQR.weight_transform (spqr.py:185-254) for asymmetric per-group weights (a symmetric weight quantizer crashes in the
 * reference's get_group_qparams): the blocked column loop with
 *   - leave-one-out outlier detection per group (spqr.py:186-203, 214-229) unless simplified_outliers or threshold = inf,
 *   - the round_zp = False quantizer (quant.py:555-559, 702-707),
 *   - the second-level scale / zero quantizers as the reference executes them on its [R, 1] tensors (spqr.py:323-345):
 *     stored scale = fl(fl(s / ss) * ss), ss = 1e-5 / (scale_qmax - scale_qmin), zero point likewise,
 *   - outlier mask err^2 > threshold, masked weights keep their value (spqr.py:239-244),
 *   - W[:, i+1:i2] -= err x Hinv[i, i+1:i2] in the block, W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:] after it.
 * W [R, K] fp32 running weights (in/out), Hinv [K, K] fp32 upper factor, threshold = relative_threshold * outlier_scale
 * (caller computes spqr.py:205-206; INFINITY = no outliers), group_size in {16, 32, 64, 128}, blocksize 128.
 * Outputs: Wout (tmp) / losses [R, K] fp32, mask [R, K] uint8, scales / zeros [R, K/group_size] fp32 in processing
 * order. Bit-exact against oracle/csrc/spqr_canon.c, which is pinned bit-exactly to the reference (tests/golden/spqr.npz). */
size_t llmc_spqr_quantize_ws_bytes(int64_t R, int64_t K);
int llmc_spqr_quantize(float* W, const float* Hinv, int64_t R, int64_t K, float qmin, float qmax,
                       int64_t group_size, float threshold, int simplified_outliers, float scale_qmin,
                       float scale_qmax, float zero_qmin, float zero_qmax, float* scales, float* zeros,
                       float* Wout, float* losses, uint8_t* mask, int blocksize, void* ws,
                       llmc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * AWQ (llmc/compression/quantization/awq.py, auto_clip.py)
 * ---------------------------------------------------------------------------------------------- */

/* Awq.get_act_scale (awq.py:74-85): mean over tokens of |x| per channel. X [N, K] dt -> out [K] dt
 * (fp32 accumulation, one rounding to dt at the end, as ATen's mean does). */
size_t llmc_awq_act_mean_ws_bytes(int64_t N, int64_t K);
int llmc_awq_act_mean(const void* X, int dt, int64_t N, int64_t K, void* out, void* ws,
                      llmc_stream_t stream);

/* Awq.get_weight_scale (awq.py:48-72) for one layer: mean over rows of |W| / rowgroup-max(|W|).
 * W [R, K] dt, groups of g along K (g = K for per-channel) -> out [K] dt: the layer's `layer_scale.mean(0)`;
 * the caller adds the layers' vectors and divides by their count (awq.py:66-72, [K]-sized torch ops). */
size_t llmc_awq_weight_mean_ws_bytes(int64_t R, int64_t K);
int llmc_awq_weight_mean(const void* W, int dt, int64_t R, int64_t K, int64_t g, void* out_dt, void* ws,
                         llmc_stream_t stream);

/* Awq.get_scales (awq.py:88-108): v2: s = x_mean^ratio clamp(1e-4); v1: x^r / w^(1-r) clamp(1e-4);
 * then s /= sqrt(max(s) * min(s)). x_mean / w_mean / out: [K] dt. w_mean may be NULL for v2. The exponents
 * ratio and 1-ratio are rounded to dt first, as ATen's CPU Tensor.pow(python_float) does. */
int llmc_awq_scales(const void* x_mean, const void* w_mean, int dt, int64_t K, double ratio, int version,
                    void* out, llmc_stream_t stream);

/* fake_quantize_weight + scaling_weight (awq.py:40-46,147-164): out = fakequant_dyn(W * s[None,:]) in dt. */
int llmc_awq_scale_fakequant(const void* W, const void* s, int dt, int64_t R, int64_t K, int64_t g,
                             int sym, float qmin, float qmax, void* out, llmc_stream_t stream);

/* scaling_input (base_blockwise_quantization.py:877-889): out = X / s[None,:] in dt. */
int llmc_div_cols(const void* X, const void* s, int dt, int64_t N, int64_t K, void* out,
                  llmc_stream_t stream);
/* The same quotient written in the k-tiled layout llmc_linear_eval_kt reads (see below): out[K/32][N][32], f16/bf16,
 * K % 32 == 0, out of place. Same bits as llmc_div_cols followed by llmc_ktile_pack. */
int llmc_div_cols_kt(const void* X, const void* s, int dt, int64_t N, int64_t K, void* out,
                     llmc_stream_t stream);
/* apply_scale helpers (base_blockwise_quantization.py:597-611,750-778): W *= s[None,:] ; v /= s. */
int llmc_mul_cols(void* W, const void* s, int dt, int64_t R, int64_t K, llmc_stream_t stream);

/* The fake-quant W4A16 matmul of Awq.search_scale_subset (awq.py:110-145, 229-236):
 *   Y = X [N, K] . Wq [R, K]^T  (16-bit in, fp32 accumulate on MFMA, rounded to dt like F.linear)
 * mode 0: store Y to Yout [N, R] (dt); Y0, when not NULL, is a bias [R] (dt) added to the fp32 sum before the single
 *         rounding, like F.linear / addmm    -- get_original_out, FakeQuantLinear.forward (module_utils.py:619-644)
 * mode 1: loss += sum((Y0 - Y)^2), diff formed in dt like the reference, squared and summed in fp32;
 *         *loss_sum (device fp32, caller zeroes) ; mean = loss_sum / (N*R) is taken by the caller. */
size_t llmc_linear_eval_ws_bytes(int64_t N, int64_t K, int64_t R);
int llmc_linear_eval(const void* X, const void* Wq, int dt, int64_t N, int64_t K, int64_t R, int mode,
                     void* Yout, const void* Y0, float* loss_sum, void* ws, llmc_stream_t stream);
/* The same product from K-TILED operands, layout T[K/32][rows][32] (the 32-k slice of every row contiguous): the
 * layout the one-wave-per-SIMD GEMM streams at whole cache lines. llmc_ktile_pack converts a row-major [rows, K]
 * 16-bit matrix; llmc_linear_eval_kt is llmc_linear_eval (same modes, outputs, Y0 and workspace, same bits) with
 * both operands k-tiled; K % 128 == 0. Used by the AWQ grid step, where x / s and fakequant(W * s) are produced
 * per evaluation anyway (awq.py:229-236).
 * mode | LLMC_LINEAR_YBLOCKED: Yout (mode 0) / Y0 (mode 1) is an opaque TILE-BLOCKED image of the [N, R] matrix,
 * llmc_linear_eval_yblocked_bytes(N, R) bytes: per 256 x 256 tile one contiguous 128 KiB in the kernel's own
 * accumulator order (tile, wave, 32 KiB-pieces, lane x 16 B), so that the reference output of the search
 * (get_original_out) is written with 16-B stores and re-read 20 times as one contiguous run per tile.
 * Small products (mode 0, row-major Yout, R % 8 == 0): when the output has fewer 256 x 256 tiles than 3/4 of the CUs and the
 * caller passes a workspace of llmc_linear_eval_ws_bytes(N, K, R) bytes, the k range is cut into 2..8 slices so that
 * tiles x slices fills the chip; the slices' fp32 partial tiles are summed in slice order and rounded once by a second
 * kernel. A reordering of the fp32 sum: within one rounding of the single-pass result. ws == NULL keeps the single pass. */
#define LLMC_LINEAR_YBLOCKED 4
size_t llmc_linear_eval_yblocked_bytes(int64_t N, int64_t R);
int llmc_ktile_pack(const void* src, int dt, int64_t rows, int64_t K, void* dst, llmc_stream_t stream);
int llmc_linear_eval_kt(const void* Xt, const void* Wt, int dt, int64_t N, int64_t K, int64_t R, int mode,
                        void* Yout, const void* Y0, float* loss_sum, void* ws, llmc_stream_t stream);

/* AutoClipper.auto_clip_layer (auto_clip.py:84-191), clip_version v1, w_only:
 *   W [R, K] dt; X [n_tok, K] dt (already token-subsampled); groups of g along K.
 *   for i_s in 0 .. n_shrink-1: max = org_max * (1 - i_s / n_grid), min = -max (clip_sym) or
 *   org_min * (1 - i_s/n_grid); q_w = fakequant_dyn(clamp(w, min, max));
 *   err = mean_tok((sum_g x*q_w - sum_g x*w)^2); keep argmin.  Outputs best_max / best_min [R, K/g] dt. */
size_t llmc_awq_clip_search_ws_bytes(int64_t R, int64_t K, int64_t g, int64_t n_tok);
int llmc_awq_clip_search(const void* W, const void* X, int dt, int64_t R, int64_t K, int64_t g,
                         int64_t n_tok, int n_grid, int n_shrink, int clip_sym, int sym, float qmin,
                         float qmax, void* best_max, void* best_min, void* ws, llmc_stream_t stream);
/* The same evaluation, returning the error table instead of its argmin: errs [n_shrink, R, K/g] (dt) = the
 * token-mean squared output error of every shrink level (auto_clip.py:150-180, `err`). Callers with SEVERAL calibration
 * batches (auto_clip_layer's `inputs` list) form err_mean = sum_i err_i / len(inputs) and the strict-< argmin
 * themselves, in the tensor dtype like the reference (auto_clip.py:176-184). */
int llmc_awq_clip_errs(const void* W, const void* X, int dt, int64_t R, int64_t K, int64_t g, int64_t n_tok,
                       int n_grid, int n_shrink, int clip_sym, int sym, float qmin, float qmax, void* errs,
                       llmc_stream_t stream);
/* The error table from GIVEN candidates — every group width (per_channel / per_tensor: g = K), every quantizer (integer,
 * FP8, learnable ranges of clip_version v2), optionally quantized activations (auto_clip.py:150-170 with
 * fake_quantize_weight :258-274 and fake_quantize_input :276-281 evaluated by the caller):
 *   Q [n_shrink, R, K] dt = the fake-quantized weights of every shrink level; XT [K, ldt] dt = the sampled tokens
 *   TRANSPOSED (token t of column k at XT[k * ldt + t], ldt % 8 == 0, 16-byte aligned, zero padded); XQT = the same
 *   layout of their fake-quantized form, or XT itself for weight-only;
 *   errs[s, r, j] = mean_tok((sum_g xq * Q[s] - sum_g x * W)^2) in dt, the sums in ATen's CPU orders
 *   (vectorized_inner_sum with its level cascade over g <= 262144 elements; the token mean as llmc_awq_clip_errs). */
int llmc_awq_clip_errs_cand(const void* W, const void* Q, const void* XT, const void* XQT, int dt, int64_t R,
                            int64_t K, int64_t g, int64_t n_tok, int64_t ldt, int n_shrink, void* errs,
                            llmc_stream_t stream);

/* apply_clip v1 (auto_clip.py:194-212): W = clamp(W, min, max) per (row, group). */
int llmc_clamp_groups(void* W, int dt, int64_t R, int64_t K, int64_t g, const void* min_val,
                      const void* max_val, llmc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * FP8 block-wise (DeepSeek-V3 style): llmc/compression/quantization/kernel.py (Triton) and FloatQuantizer `per_block`
 * ------------------------------------------------------------------------------------------------ */
/* weight_cast_to_fp8 (kernel.py:58-86; quant.py:33-43) and FloatQuantizer per_block fake / real quant
 * (quant.py:132-143, 612-658, 1043-1072, 1195-1221). W [M, N] dt; one fp32 scale per block x block tile, row-major
 * [ceil(M/block), ceil(N/block)]: scale = max(absmax, clamp_min) / 448; q = e4m3fn(W / scale) (fp32 division, RNE).
 * clamp_min = 1e-5: FloatQuantizer (`.clamp(min=1e-5)`, zero scales replaced by 1); clamp_min = 0: the Triton kernel
 * (an all-zero block yields scale 0 and NaN codes, like 0 / 0 there). fake bit 0: out is dt = q * scale, else e4m3
 * bytes; fake bit 1: `scales` is an INPUT (the *_static forms, quant.py:1074-1160); fake bit 8: qtorch.float_quantize
 * rounding (FloatQuantizer) instead of the e4m3fn cast (the Triton kernels), see llmc_fp8_quant. */
int llmc_fp8_block_quant(const void* W, int dt, int64_t M, int64_t N, int block, float clamp_min, int fake, void* out,
                         float* scales, llmc_stream_t stream);
/* weight_cast_to_bf16 (kernel.py:89-143; quant.py:18-30): out[m, n] = float(W8[m, n]) * scales[m/block, n/block] in out_dt. */
int llmc_fp8_block_dequant(const void* W8, const float* scales, int64_t M, int64_t N, int block, int out_dt, void* out,
                           llmc_stream_t stream);
/* act_quant (kernel.py:7-55): X contiguous, n_elem elements, every `block` consecutive ones share scale = absmax / 448
 * (fp32, no clamp); out8 e4m3 bytes, scales [n_elem / block]. */
int llmc_fp8_act_quant(const void* X, int dt, int64_t n_elem, int block, void* out8, float* scales, llmc_stream_t stream);
/* fp8_gemm + block_wise_fp8_forward_func (kernel.py:146-242; module_utils.py:40-45): A8 [M, K] e4m3 with a_s [M, K/128],
 * B8 [N, K] e4m3 (a weight) with b_s [N/128, K/128]; C [M, N] out_dt = sum over 128-deep K blocks of
 * (A_kb . B_kb^T) * a_s * b_s, fp32 accumulation on the fp8 MFMA; bias [N] (out_dt) is added after the rounding.
 * The K-block update is the reference Triton kernel's, acc = fma(part * a_s, b_s, acc): bit-identical outputs, two VALU ops per
 * element and K block, which bound the kernel at 0.31-0.37 of the fp8 MFMA peak. out_dt | LLMC_FP8_GEMM_FUSED_SCALE (opt-in, for
 * callers that do not need bit-identity with the Triton kernel): acc = fma(part, fl(a_s * b_s), acc), one op — the same value up
 * to one fp32 rounding of the scale product per row and K block (the 256 x 256-tile kernel only; the small-shape kernel ignores it). */
#define LLMC_FP8_GEMM_FUSED_SCALE 0x100
int llmc_fp8_block_gemm(const void* A8, const float* a_s, const void* B8, const float* b_s, int64_t M, int64_t N,
                        int64_t K, int out_dt, const void* bias, void* C, llmc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LLMC_HIP_H_ */
