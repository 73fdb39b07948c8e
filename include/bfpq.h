/*
 * bfpq.h -- C ABI of libbfpq.so, the MI355X (gfx950) BFP quantize + sparsify engine.
 *
 * This is the drop-in boundary for the reference's hot path
 *     src/transformers/bfp/bfp_ops.py:16-149   (get_exponent, _convert_blocked_float_to_bfp,
 *     _no_sparsity_float_to_bfp, _structured_N_M_sparsity, _unstructured_sparsity, _sparsify,
 *     _quantize, float_to_bfp_blocked)
 * The reference has no native code on this path (it is ~25 ATen passes per call); these entry
 * points are what a ctypes / pybind binding placed in that file would call instead.
 *
 * Conventions
 *   - every pointer named *_dev is a DEVICE pointer owned by the caller; *_host is host memory
 *   - tensors are dense row-major [rows, cols]; blocks and N:M groups run along cols only and
 *     never cross a row (bfp_ops.py:50-59, :79-91); leading dims are flattened into rows
 *   - dtype: BFPQ_F32 / BFPQ_F16 / BFPQ_BF16 (input and dequantised output share the dtype)
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*); nothing here
 *     synchronises, allocates or frees device memory, so every call is hipGraph-capturable
 *   - return value: 0 on success, a negative BFPQ_E_* for argument errors, a positive
 *     hipError_t if a launch failed
 *   - rounding is round-half-to-even ('determ', bfp_ops.py:24-25) unless a stochastic seed is given
 */
#ifndef BFPQ_H
#define BFPQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFPQ_VERSION 4

enum { BFPQ_F32 = 0, BFPQ_F16 = 1, BFPQ_BF16 = 2 };

enum {
    BFPQ_E_ARG = -1,        /* bad shape / pointer / enum                                        */
    BFPQ_E_UNSUPPORTED = -2,/* valid in the reference but not handled by this entry point        */
    BFPQ_E_ALIGN = -3       /* a fast-path entry point was given a misaligned pointer            */
};

/* sizes of the small constant tables the kernels read (built on the host, uploaded by the caller
 * once per device; keeping them caller-owned keeps the library stateless) */
#define BFPQ_EXP_WIN_ENTRIES 320   /* uint8 per unbiased exponent k in [-160, 160)              */
#define BFPQ_NM4_LUT_ENTRIES 729   /* uint8 keep-mask per 3^6 pairwise-comparison signature     */
#define BFPQ_NM8_LUT_ENTRIES (1 << 24) /* uint8 prune-mask per (number of smaller keys)^8 vector, 3 bits each */

int bfpq_version(void);
/* process-wide tuning knobs (measurement aid; defaults are the measured optimum on MI355X) */
#define BFPQ_TUNE_MAX_GRID 0       /* cap on workgroups of the streaming kernels (default 1024) */
#define BFPQ_TUNE_GEMM_ROW_TILES 1 /* 16-row tiles per wave in bfpq_hbfp_linear_decode_tiled: 0 = choose (default), 1, 2, 4 */
#define BFPQ_TUNE_LIST_OWN_MB 3    /* list calls: tensors from this many MB on get launches of their own (default 24; 0 = all, a huge value = none) */
#define BFPQ_TUNE_MX8_VARIANT 2    /* tile shape of bfpq_hbfp_linear_mx8: -1 = choose (default), 0..6 force (A/B measurements) */
int bfpq_tune(int key, int value);
const char* bfpq_error_string(int code);

/* ---- host-side table builders ---------------------------------------------------------------
 * bfpq_exp_window_host: replaces the dtype-dependent rounding inside get_exponent
 * (bfp_ops.py:29-33): for s = 2^k (1+f) the reference's ceil(log2(s)) evaluated in `dtype` is k
 * when the dtype mantissa field of s is <= table[k + 160], else k + 1.
 * bfpq_nm4_lut_host: keep-mask (bit i set = element i kept) of one group of 4 for "keep N of 4",
 * indexed by sum_{p} c_p 3^p over pairs p = (0,1),(0,2),(0,3),(1,2),(1,3),(2,3) with
 * c = 0/1/2 for |a_i| <,==,> |a_j|; reproduces ATen CPU topk's std::nth_element tie order
 * used by _structured_N_M_sparsity (bfp_ops.py:85). */
int bfpq_exp_window_host(int dtype, uint8_t* table_host /* [BFPQ_EXP_WIN_ENTRIES] */);
int bfpq_nm4_lut_host(int N, uint8_t* lut_host /* [BFPQ_NM4_LUT_ENTRIES] */);
/* bfpq_nm8_lut_host: PRUNE mask (bit i set = element i zeroed) of one group of 8 for "keep N of 8", indexed by
 * sum_i less_i << (3 i) with less_i = number of keys strictly smaller than key i (this vector identifies the weak
 * ordering of the group, which is all std::nth_element's behaviour depends on).  16 MiB, optional: pass it as nm4_lut_dev
 * when M == 8; with NULL the kernel replays nth_element for the groups whose ties straddle the cut (slower, same result). */
int bfpq_nm8_lut_host(int N, uint8_t* lut_host /* [BFPQ_NM8_LUT_ENTRIES] */);
/* host restatement of the N:M selection for one group (any 1 <= N <= M <= 64); keys are
 * non-negative magnitudes compared as unsigned integers. Returns the 64-bit PRUNE mask. Used by the
 * host-logic tests and to build the LUT above. */
uint64_t bfpq_nm_prune_mask_host(const uint32_t* keys, int N, int M);

/* ---- fused quantize (+ N:M) -----------------------------------------------------------------
 * Replaces float_to_bfp_blocked (bfp_ops.py:124-149) for sparsity_mode 'structured' or no
 * sparsity, sparsity_num_format 'bfp' or 'fp32':
 *     sparsify_first != 0:  out = Q(S(in))   (first == 's')
 *     sparsify_first == 0:  out = S(Q(in))
 * with Q = _no_sparsity_float_to_bfp (block_size, mant_bits; block_size == 0 -> identity, the
 * 'fp32' format) and S = _structured_N_M_sparsity (N, M; M == 0 -> identity).
 * Outputs (each nullable, at least one required):
 *   out_deq_dev   [rows, cols] dtype          the reference's fake-quantised tensor (drop-in mode)
 *   out_codes_dev sign-magnitude mantissas as two's-complement integers, code_bits wide
 *                 (4: two per byte, low nibble first, [rows, ceil(cols/2)] bytes; 8: int8; 16: int16;
 *                 32: not codes but the fp32 image of the dequantised tensor, [rows, cols] float -- what the
 *                 reference returns for 'stoc' rounding of a half tensor; single-pass kernel only)
 *   out_exp_dev   int8 [rows, ceil(cols/block)] shared exponent e (value = code * 2^(e-mant_bits));
 *                 saturated to [-127, 127]; -128 marks a block the reference turns into NaN
 * Any shape, block size and 1 <= N <= M <= 64 is accepted; the single-pass fused kernel is used
 * when cols % block == 0, block is 16 bytes x a power of two <= 64 lanes, and M in {0, 2, 4} (or 8 on a
 * 16-bit dtype) divides the block; otherwise the work is split into a sparsify and a quantize launch through
 * scratch_dev (rows*cols elements of dtype, may be NULL when the fused kernel applies).
 * stoch_seed == 0 -> round-half-even; otherwise stochastic rounding (uniform dither in
 * [-0.5, 0.5) from a counter-based generator keyed by (seed, element index), bfp_ops.py:21-23). */
int bfpq_quantize_nm(const void* in_dev, void* out_deq_dev, void* out_codes_dev, int8_t* out_exp_dev,
                     int64_t rows, int64_t cols, int dtype,
                     int block_size, int mant_bits, double epsilon,
                     int N, int M, int sparsify_first,
                     int code_bits, uint64_t stoch_seed,
                     const uint8_t* exp_win_dev, const uint8_t* nm4_lut_dev,
                     void* scratch_dev, void* stream);

/* The drop-in call with its constants bound once (what a module's forward issues per tensor): the same work as
 * bfpq_quantize_nm(in, out, NULL, NULL, rows, cols, plan->...) in round-half-even mode, six arguments instead of nineteen.
 * Shapes that need the two-launch general path work in place through out_dev (no scratch). */
typedef struct bfpq_plan {
    int dtype, block_size, mant_bits, N, M, sparsify_first;
    double epsilon;
    const uint8_t* exp_win_dev;   /* BFPQ_EXP_WIN_ENTRIES, for dtype (NULL when block_size == 0) */
    const uint8_t* nm_lut_dev;    /* bfpq_nm4_lut_host(N) for M == 4, bfpq_nm8_lut_host(N) or NULL for M == 8, else NULL */
} bfpq_plan;
int bfpq_fake_quantize(const bfpq_plan* plan_host, const void* in_dev, void* out_dev, int64_t rows, int64_t cols, void* stream);

/* A LIST of tensors in as few launches as possible (up to 64 tensors per launch, their descriptors travel as kernel
 * arguments: no device memory, graph-capturable).  All tensors share the plan (dtype, block, mantissa width, N:M); apply_nm
 * says per tensor whether the plan's N:M pruning applies (a Linear's weight) or not (its activation).  What a model pass
 * issues for "all Linear weights", and what one BFPLinear forward issues for its activation + weight.  Tensors whose
 * shape the single-pass kernel does not take (ragged rows, M = 8, ...) are handled one by one inside the call. */
typedef struct bfpq_tensor_desc { const void* in_dev; void* out_dev; int64_t rows, cols; int apply_nm; int reserved; } bfpq_tensor_desc;
int bfpq_fake_quantize_batched(const bfpq_plan* plan_host, const bfpq_tensor_desc* descs_host, int n, void* stream);
/* The same list over several LANES (ABI 4): the caller's stream and up to 7 aux streams.  Inside either call a tensor of 24 MB or more
 * gets a launch of its own (the list kernel only pays where a launch of its own would be mostly ramp and tail); with aux streams those
 * launches are spread over the lanes -- each to the lane that has been given the fewest bytes so far -- so that the tail of one
 * tensor's launch runs beside the ramp of another's (64 x [4096,11008] bf16 2:4 -> HBFP4, us per tensor: one aux stream 29.5-30.2,
 * none 31.9-33.4, the list kernel 35.7-37.0; LLaMA-7B's 224 weights 4.07 ms against 5.31; more than one aux stream buys nothing here).
 * The aux lanes wait for what `stream` held at the call and `stream` waits for them before the call returns (one fork, one join, no
 * event between the lanes: hipGraph-capturable from `stream`).  A list with fewer than two such tensors stays on `stream` (a Linear's
 * weight next to its activation: the fork and join would cost more than they gain).  The tensors of one list must not overlap one
 * another (they are processed side by side).  bfpq_fake_quantize_batched(...) is bfpq_fake_quantize_list(..., stream, NULL, 0). */
int bfpq_fake_quantize_list(const bfpq_plan* plan_host, const bfpq_tensor_desc* descs_host, int n, void* stream,
                            void* const* aux_streams_host, int n_aux);

/* returns 1 if bfpq_quantize_nm would take the single-pass fused kernel for this problem */
int bfpq_is_fused(int64_t rows, int64_t cols, int dtype, int block_size, int N, int M);

/* ---- N:M only (replaces _structured_N_M_sparsity, bfp_ops.py:73-91) --------------------------- */
int bfpq_nm_sparsify(const void* in_dev, void* out_dev, int64_t rows, int64_t cols, int dtype,
                     int N, int M, const uint8_t* nm4_lut_dev, void* stream);

/* ---- unstructured magnitude pruning (replaces _unstructured_sparsity, bfp_ops.py:61-71) -------
 * Exact global k-th smallest |v| by radix select on the magnitude bit pattern, then "zero everything below the
 * threshold tau and the first `need` (flat index order, lower ranks first) of the elements equal to it".
 *
 * Single device, TWO launches for a 16-bit dtype:
 *   1. bfpq_select(...)            one launch: per-segment histograms, and the LAST workgroup to finish (an atomic ticket, no
 *                                  workgroup ever waits for another) resolves threshold + tie bookkeeping into ws_dev.
 *                                  fp32: two such launches (the high 15 bits of the key, then the low 16 bits of the keys that
 *                                  share them): the tensor is read three times in all, the launch-pair form reads it four times.
 *   2. bfpq_threshold_apply(...) or bfpq_quantize_threshold(...)   prune (and quantize) in one pass
 * ws_dev must be ZERO before its first use; every call leaves it ready for the next one.
 *
 * Multi-GPU (row slabs, one global threshold): for pass p in [0, bfpq_select_passes(dtype)):
 *     bfpq_select_hist(...)      histogram of the current digit of this device's elements into a caller-owned hist_dev
 *                                (BFPQ_SELECT_HIST_COPIES x BFPQ_SELECT_HIST_ENTRIES uint32, zero on entry); on the last pass also
 *                                per-segment windows of it -> ws_dev
 *     ALL-GATHER the per-rank buffers into hist_all_dev [n_ranks][BFPQ_SELECT_HIST_COPIES][BFPQ_SELECT_HIST_ENTRIES] -- the one
 *                                exchange of the path: the per-rank counts of the threshold bin also tell every rank how many
 *                                ties lower ranks hold
 *     bfpq_select_resolve(...)   picks the digit (pass 0 starts the selection with k); on the last pass leaves tau, need, ties
 *                                and the tie bookkeeping of this device's slab in ws_dev; clears the local histogram for its
 *                                next use (zero_hist_dev = hist_dev)
 *   then the apply launch as above.  (hist_dev = NULL / hist_all_dev = NULL run the same launches on histogram buffers inside
 *   ws_dev; fp32 takes three pairs, 11 + 11 + 9 bits.)
 * ws_dev: BFPQ_SELECT_WS_BYTES bytes, zeroed once, 16-byte aligned; it begins with a bfpq_select_state.
 * k is the GLOBAL prune count int(numel_global * frac) (bfp_ops.py:66); numel_global < 2^32.
 * in_dev and out_dev of the apply launch must not alias (the tie ranks are counted from the input while other tiles are written).
 * Tie positions: the reference's are those of a sequential introselect over the whole tensor and
 * are not reproduced; threshold, count and every element outside the tie class are.
 * How ties are ranked without a rank per tile: csrc/bfpq_device.h, block comment at "Unstructured pruning". */
#define BFPQ_SELECT_STATE_BYTES 80
#define BFPQ_SELECT_MAX_SEGMENTS 256
#define BFPQ_SELECT_WINDOW_BINS 2048
#define BFPQ_SELECT_HIST_ENTRIES (32768 + 256)   /* fine bins, then (16-bit dtypes) 256 coarse bins of 128 */
#define BFPQ_SELECT_HIST_COPIES 8                /* a device accumulates into 8 copies (cuts the contention of the flush) */
#define BFPQ_SELECT_WS_BYTES (BFPQ_SELECT_STATE_BYTES + 4 * (12 + BFPQ_SELECT_HIST_COPIES * 512 + 3 * BFPQ_SELECT_HIST_COPIES * BFPQ_SELECT_HIST_ENTRIES + \
                              3 * BFPQ_SELECT_MAX_SEGMENTS + 2 * BFPQ_SELECT_MAX_SEGMENTS * BFPQ_SELECT_WINDOW_BINS))

int bfpq_select_passes(int dtype);
int64_t bfpq_select_ws_bytes(void);
/* single device: every launch of the selection (16-bit dtypes: ONE, fp32: TWO) */
int bfpq_select(const void* in_dev, int64_t numel, int dtype, int64_t k, void* ws_dev, void* stream);
int bfpq_select_hist(const void* in_dev, int64_t numel, int dtype, int pass, int64_t k, int64_t numel_global,
                     void* ws_dev, uint32_t* hist_dev, void* stream);
int bfpq_select_resolve(const void* in_dev, int64_t numel, int dtype, int pass, int64_t k,
                        const uint32_t* hist_all_dev /* [n_ranks][COPIES][ENTRIES] */, int n_ranks, int rank,
                        void* ws_dev, uint32_t* zero_hist_dev, void* stream);
/* The same resolve launch over a gathered block of another shape (ABI 4): per rank `copies` histograms (1 ... BFPQ_SELECT_HIST_COPIES) of
 * BFPQ_SELECT_HIST_ENTRIES words, rank r's set at hist_all_dev + r * rank_stride_words.  What a LIST of row-sharded tensors uses to get by with ONE
 * exchange per radix pass for the whole list instead of one per tensor: every tensor's local histogram is launched first, the copies of each are
 * summed (a tensor then sends 132 KB instead of 1 MB), the [tensors, ENTRIES] block is all-gathered in one collective, and tensor i resolves
 * from hist_all_dev = gathered + i * ENTRIES with copies = 1, rank_stride_words = tensors * ENTRIES.  zero_hist_dev as above: the tensor's
 * local 8-copy buffer, cleared for its next use.  bfpq_select_resolve(...) is the call with copies = 8, rank_stride_words = 8 * ENTRIES. */
int bfpq_select_resolve_ex(const void* in_dev, int64_t numel, int dtype, int pass, int64_t k,
                           const uint32_t* hist_all_dev, int n_ranks, int rank, int copies, int64_t rank_stride_words,
                           void* ws_dev, uint32_t* zero_hist_dev, void* stream);
int bfpq_select_reset(void* ws_dev, void* stream);
int bfpq_threshold_apply(const void* in_dev, void* out_dev, int64_t numel, int dtype, void* ws_dev, void* stream);
/* step 3 fused with the quantizer: out = Q(S_threshold(in)) in one pass over the tensor (first == 's',
 * bfp_ops.py:141-144 with sparsity_mode 'unstructured'); same outputs / tables / fallbacks as
 * bfpq_quantize_nm (scratch_dev only when the shape needs the two-launch path and out_deq_dev is NULL). */
int bfpq_quantize_threshold(const void* in_dev, void* out_deq_dev, void* out_codes_dev, int8_t* out_exp_dev,
                            int64_t rows, int64_t cols, int dtype, int block_size, int mant_bits, double epsilon,
                            int code_bits, uint64_t stoch_seed, const uint8_t* exp_win_dev,
                            void* ws_dev, void* scratch_dev, void* stream);

/* The whole s-first unstructured drop-in op, out = Q(S_unstructured(in)) with k = int(numel * frac) elements pruned
 * (float_to_bfp_blocked with sparsity_mode 'unstructured', first 's': bfp_ops.py:61-71, :141-144), single device, round-half-even:
 * bfpq_select + bfpq_quantize_threshold behind one call (two launches for a 16-bit dtype; the tensor is read by both). */
int bfpq_prune_quantize(const void* in_dev, void* out_dev, int64_t rows, int64_t cols, int dtype, int block_size, int mant_bits,
                        double epsilon, int64_t k, const uint8_t* exp_win_dev, void* ws_dev, void* stream);
/* The same op for a LIST of tensors -- every Linear weight of a model (BASELINE config 4) -- over independent LANES (ABI 4): the
 * caller's stream and up to 7 aux streams, a workspace per lane (ws_devs: n_ws distinct caller-owned workspaces, BFPQ_SELECT_WS_BYTES
 * each, zeroed once; lanes used = min(1 + n_aux, n_ws)).  A tensor's two launches go back to back to the lane that has been given the
 * fewest bytes so far; nothing orders the lanes against one another, so a lane's serial tail (publishing, the ticket, the resolve
 * step, the kernel boundary: ~13 us in which that lane moves no data) is covered by the other lanes' streaming.  LLaMA-13B, all 280
 * weights, bf16: one lane 16.8 ms, two 14.2, three 12.7, four 12.5.  The aux lanes wait for what `stream` held at the call and
 * `stream` waits for them before the call returns (one fork, one join: hipGraph-capturable from `stream`).  The tensors of one list
 * must not overlap one another.  Creates and destroys a handful of hipEvents per call; launches only, no synchronisation.
 * bfpq_prune_quantize_batched (ABI 3) is the same call with at most one aux stream. */
typedef struct bfpq_prune_desc { const void* in_dev; void* out_dev; int64_t rows, cols; int64_t k; } bfpq_prune_desc;
int bfpq_prune_quantize_list(const bfpq_prune_desc* descs_host, int n, int dtype, int block_size, int mant_bits, double epsilon,
                             const uint8_t* exp_win_dev, void* const* ws_devs_host, int n_ws, void* stream,
                             void* const* aux_streams_host, int n_aux);
int bfpq_prune_quantize_batched(const bfpq_prune_desc* descs_host, int n, int dtype, int block_size, int mant_bits, double epsilon,
                                const uint8_t* exp_win_dev, void* const* ws_devs_host, int n_ws, void* stream, void* aux_stream);

/* ---- 'int' per-channel format (replaces _quantize's 'int' branch, bfp_ops.py:111-120, i.e.
 * int_ops.Quantizer.configure/find_params/quantize with the defaults perchannel=True, sym=True) -------
 * The tensor is viewed as [outer, C, inner] with the channel in the middle (int_ops.py:38-50):
 * weight: outer = 1, C = shape[0], inner = rest; 2-D/3-D activation: outer = rows, C = last dim,
 * inner = 1; 4-D activation: outer = N, C = shape[1], inner = H*W.  maxq = 2^bits - 1.
 * out_dev is fp32 for every input dtype (as in the reference).  ws_dev: bfpq_int_workspace_elems(C)
 * uint32, needed (any content) unless outer == 1. */
int64_t bfpq_int_workspace_elems(int64_t C);
int bfpq_int_quantize(const void* in_dev, float* out_dev, int64_t outer, int64_t C, int64_t inner, int dtype, int bits,
                      uint32_t* ws_dev, void* stream);

/* ---- packed HBFP -> tensor (no counterpart in the reference, which never stores quantized tensors;
 * SURVEY §8f next #3): out = code * 2^(exp - mant_bits) in dtype, from the codes / exponents that
 * bfpq_quantize_nm / bfpq_quantize_threshold write.  Equals their out_deq except that a negative zero
 * comes back as +0 (a two's-complement code has no -0) and saturated exponents (|e| > 127) are lost. */
int bfpq_dequantize(const void* codes_dev, const int8_t* exp_dev, void* out_dev, int64_t rows, int64_t cols, int dtype,
                    int block_size, int mant_bits, int code_bits, void* stream);

/* ---- 2:4 compaction of 4-bit codes (optional part of the packed format) --------------------
 * codes_dev: n_bytes bytes of 4-bit codes as bfpq_quantize_nm writes them (n_bytes % 4 == 0; 8 elements per 4 bytes),
 * every aligned group of 4 codes holding at most 2 non-zeros (N=2, M=4 in either order guarantees it).
 * vals_dev: n_bytes / 2 bytes (per group one byte: the two kept codes, lower position in the low nibble);
 * idx_dev: n_bytes / 4 bytes (per group 4 bits: the two positions).  0.375 B per element instead of 0.5.
 * status_dev[0] (int32, zeroed by the caller) is set to 1 if some group had more than 2 non-zero codes.
 * bfpq_expand24 is the exact inverse. */
int bfpq_compact24(const void* codes_dev, void* vals_dev, void* idx_dev, int64_t n_bytes, int32_t* status_dev, void* stream);
int bfpq_expand24(const void* vals_dev, const void* idx_dev, void* codes_dev, int64_t n_bytes, void* stream);

/* ---- packed-format consumer for decode (no reference counterpart: it runs F.linear on fake-quantised tensors,
 * bfp_ops.py:187-190): out[t][n] = sum_k x[t][k] W[n][k], T <= 16 tokens, W packed HBFP
 * (4-bit codes, mant_bits <= 3) and x packed HBFP (int8 codes, mant_bits <= 7), both block 64, exact integer
 * block sums on the int8 matrix cores, fp32 across blocks.  N % 16 == 0, K % 256 == 0.
 *   xcodes_dev [16, K] int8 and xexp_dev [16, K/64] (rows >= T: any content, their results are discarded),
 *   slabs_dev [bfpq_hbfp_linear_slices(N, K), 16, N] fp32 scratch, out_dev [T, N] of out_dtype. */
int bfpq_hbfp_linear_slices(int64_t N, int64_t K);
int bfpq_hbfp_linear_decode(const void* wcodes_dev, const int8_t* wexp_dev, const int8_t* xcodes_dev, const int8_t* xexp_dev,
                            void* out_dev, float* slabs_dev, int64_t T, int64_t N, int64_t K,
                            int out_dtype, int w_mant_bits, int x_mant_bits, void* stream);

/* the same product from the MFMA-tiled weight layout (a one-time repack of the packed weight; K % 128 == 0, K >= 256):
 *   wtiles_dev [N/16][K/128][64][16] bytes: lane l = r + 16 q of the pair p holds the 16 codes k = 16q..16q+15 of row
 *               16*rt + r for block 2p (bytes 0-7) and block 2p+1 (bytes 8-15)
 *   wexpt_dev  [N/16][K/128][16][2] int8: exponents of the pair's two blocks per row
 * Up to 64 tokens per call here (groups of 16 tokens walk the weight one after the other inside the one launch; xcodes_dev /
 * xexp_dev then hold T rows).
 * One launch, no workspace: each workgroup owns one or two 16-row tiles, its waves split K and are summed in slice order
 * through LDS, so the result is reproducible. */
int bfpq_hbfp_linear_tiled_ok(int64_t N, int64_t K);   /* 1 when the tiled layout applies to [N, K] */
int bfpq_hbfp_linear_decode_tiled(const void* wtiles_dev, const void* wexpt_dev, const int8_t* xcodes_dev, const int8_t* xexp_dev,
                                  void* out_dev, int64_t T, int64_t N, int64_t K,
                                  int out_dtype, int w_mant_bits, int x_mant_bits, void* stream);

/* ---- packed-format consumer for prefill (replaces F.linear on the two fake-quantised operands,
 * bfp_ops.py:187-190): out[T, N] = x[T, K] W[N, K]^T (+ bias[N]) for any T, both operands HBFP with block 64 and mantissas that
 * fit e4m3 exactly (mant_bits <= 4), on CDNA4's block-scaled matrix instruction (one v_mfma_scale_f32_32x32x64_f8f6f4 = the
 * exact integer dot product of one HBFP block times the two power-of-two block scales, fp32 across blocks).
 * Operand images ("mx8"): e4m3 bytes [rows, K] + E8M0 scale bytes [rows, K/64], made by bfpq_mx8_from_hbfp from the codes
 * (code_bits 4: two's-complement nibbles, or 8: int8) and int8 exponents that bfpq_quantize_nm writes (exponent -128, the NaN
 * block marker, becomes the E8M0 NaN).  K % 256 == 0; x8 / w8 16-byte aligned, the scale arrays 4-byte aligned;
 * bias (nullable) and out of out_dtype. */
int bfpq_mx8_from_hbfp(const void* codes_dev, const int8_t* exp_dev, void* out8_dev, void* out_scale_dev, int64_t rows, int64_t cols,
                       int code_bits, int mant_bits, void* stream);
/* a tensor straight to its mx8 image (dense HBFP quantize, block 64, round-half-even: the activation operand of a
 * prefill; one pass, sizeof(dtype) B read + 1.016 B written per element).  cols % 64 != 0 / unaligned pointers: BFPQ_E_UNSUPPORTED
 * (take bfpq_quantize_nm with int8 codes + bfpq_mx8_from_hbfp). */
int bfpq_quantize_mx8(const void* in_dev, void* out8_dev, void* out_scale_dev, int64_t rows, int64_t cols, int dtype, int mant_bits,
                      double epsilon, const uint8_t* exp_win_dev, void* stream);
int bfpq_hbfp_linear_mx8_ok(int64_t T, int64_t N, int64_t K);   /* 1 when the kernel applies */
int bfpq_hbfp_linear_mx8(const void* x8_dev, const void* xscale_dev, const void* w8_dev, const void* wscale_dev, const void* bias_dev,
                         void* out_dev, int64_t T, int64_t N, int64_t K, int out_dtype, void* stream);

/* short token counts (<= 256) leave too few output tiles for the chip (128 tokens x [4096 x 11008]: 32 tiles): K is then split over
 * `parts` workgroups per tile, each writing an fp32 slab, added in part order (deterministic) by a second small launch.
 * bfpq_hbfp_linear_mx8_parts: the number of parts the plan wants for this shape (1: use bfpq_hbfp_linear_mx8);
 * slabs_dev: parts * T * N floats, 16-byte aligned. */
int bfpq_hbfp_linear_mx8_parts(int64_t T, int64_t N, int64_t K);
int bfpq_hbfp_linear_mx8_splitk(const void* x8_dev, const void* xscale_dev, const void* w8_dev, const void* wscale_dev, const void* bias_dev,
                                void* out_dev, float* slabs_dev, int parts, int64_t T, int64_t N, int64_t K, int out_dtype, void* stream);

/* the first BFPQ_SELECT_STATE_BYTES of ws_dev, as read back by a host that wants tau / counts (little-endian) */
typedef struct bfpq_select_state {
    uint32_t prefix;      /* magnitude bits decided so far (high digits)                         */
    uint32_t prefix_mask; /* which bits of prefix are decided                                    */
    int64_t k_rem;        /* how many of the elements matching prefix are still to be pruned     */
    uint32_t tau;         /* after the last pass: magnitude bit pattern of the threshold         */
    uint32_t done;        /* 1 after the last pass                                               */
    int64_t need;         /* how many elements equal to tau get pruned (over all ranks)          */
    int64_t ties;         /* how many elements equal tau in total (over all ranks)               */
    int64_t k;            /* the k given to the selection                                        */
    int64_t tie_base;     /* elements equal to tau held by lower ranks                           */
    uint32_t flags;       /* bit 0: cut_* below are valid (else the apply launch derives them from the per-segment tie counts);
                             bit 1: the fine histograms inside ws_dev were used (the apply launch clears them) */
    uint32_t cut_lo;      /* this device's elements equal to tau go in lane items (16 B) [0, cut_lo), stay in [cut_hi, end), */
    uint32_t cut_hi;      /* and inside [cut_lo, cut_hi) -- one segment of the histogram launch, or empty -- the first        */
    uint32_t cut_within;  /* cut_within of them (flat order) go                                                              */
    uint32_t cut_total;   /* elements equal to tau inside [cut_lo, cut_hi)                                                   */
    uint32_t reserved;
} bfpq_select_state;

#ifdef __cplusplus
}
#endif
#endif /* BFPQ_H */
