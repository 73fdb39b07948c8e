/*
 * oracle/bfp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU-only restatement of the reference's BFP quantize + sparsify hot path
 * (reference: src/transformers/bfp/bfp_ops.py:16-149).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's library; the product (the HIP path
 * under quantization-sparsity-interplay_amd/) never does.
 *
 * Parity pin: this restatement is checked against outputs of the reference itself, run in the
 * build container by tests/golden/make_golden.py and frozen under tests/golden/ (*.npz).
 *
 * The reference computes everything with ATen ops *in the tensor's dtype*.  ATen's CPU kernels
 * for fp16/bf16 evaluate each elementwise op in fp32 and round the result back to the dtype, so
 * every step below is "fp32 op, then round to dtype" (rnd()).
 *
 * Third-party arithmetic on the path (not in /root/reference, restated from its published
 * algorithm): ATen CPU topk (torch==2.1.0 pinned by the reference, requirements_pip.txt:58;
 * 2.10.0 in the container) selects the k smallest |v| with libstdc++ std::nth_element over
 * (value,index) pairs whenever k*64 > n, which holds for every N:M group and for 50 %
 * unstructured pruning.  nth_element / introselect / heap_select / insertion sort below follow
 * libstdc++'s <bits/stl_algo.h>, <bits/stl_heap.h> (GCC 11) step by step, because the tie
 * behaviour (which of several equal |v| gets pruned) is defined by that exact sequence.
 *
 * dtype codes: 0 = fp32, 1 = fp16, 2 = bf16.  Tensors travel as raw bit patterns.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * dtype helpers: bits <-> fp32, round-to-nearest-even fp32 -> dtype
 * ---------------------------------------------------------------------------------------- */
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }

static inline uint16_t f32_to_bf16(float f)
{
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline float f16_to_f32(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return u2f(sign);
        /* subnormal: value = man * 2^-24 */
        float v = (float)man * 5.9604644775390625e-08f;
        return (sign ? -v : v);
    }
    if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
    return u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

static inline uint16_t f32_to_f16(float f)
{
    uint32_t u = f2u(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);            /* NaN */
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);           /* >= 65520 -> inf */
    if (a < 0x33000001u) return (uint16_t)sign;                        /* <= 2^-25 -> 0 (ties to even) */
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;                          /* 24-bit significand */
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                     /* bits dropped */
    uint32_t kept = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (kept & 1u))) kept++;
    uint32_t out;
    if (e < -14) out = kept;                                           /* subnormal (may carry into normal) */
    else out = ((uint32_t)(e + 15) << 10) + (kept - 0x400u);           /* carry propagates into exponent */
    return (uint16_t)(sign | out);
}

static inline float ld(const void* p, int64_t i, int dtype)
{
    if (dtype == 0) return ((const float*)p)[i];
    if (dtype == 1) return f16_to_f32(((const uint16_t*)p)[i]);
    return bf16_to_f32(((const uint16_t*)p)[i]);
}

static inline void st(void* p, int64_t i, int dtype, float v)
{
    if (dtype == 0) ((float*)p)[i] = v;
    else if (dtype == 1) ((uint16_t*)p)[i] = f32_to_f16(v);
    else ((uint16_t*)p)[i] = f32_to_bf16(v);
}

/* round an fp32 intermediate to the tensor dtype and come back to fp32 */
static inline float rnd(float v, int dtype)
{
    if (dtype == 0) return v;
    if (dtype == 1) return f16_to_f32(f32_to_f16(v));
    return bf16_to_f32(f32_to_bf16(v));
}

static inline size_t esize(int dtype) { return dtype == 0 ? 4 : 2; }

/* torch.max / torch.min (elementwise "maximum"/"minimum"): NaN propagates */
static inline float t_max(float a, float b) { if (a != a) return a; if (b != b) return b; return a > b ? a : b; }
static inline float t_min(float a, float b) { if (a != a) return a; if (b != b) return b; return a < b ? a : b; }

/* ------------------------------------------------------------------------------------------
 * a2: get_exponent  (bfp_ops.py:29-33)   (max|t| + eps).log2().ceil(), all in t.dtype
 * a3: _convert_blocked_float_to_bfp (bfp_ops.py:35-44), rounding_mode == 'determ'
 * ---------------------------------------------------------------------------------------- */
static float block_exponent(const float* blk, int n, float eps, int dtype)
{
    /* t.abs().max(dim=1): NaN propagates through max */
    float m = 0.0f;
    int has_nan = 0;
    for (int i = 0; i < n; i++) {
        float a = fabsf(blk[i]);
        if (a != a) has_nan = 1;
        else if (a > m) m = a;
    }
    if (has_nan) m = NAN;
    /* bfp_ops.py:33  max_v + epsilon: ATen casts the Python scalar to the tensor dtype first
     * (bf16(1e-8) = 0x322C, fp16(1e-8) = 0), adds in fp32 and rounds the sum to the dtype --
     * pinned by fixture G1 (patterns 0x31BE, 0x32B5, 0x3360 tell the two orders apart). */
    float s = rnd(m + rnd(eps, dtype), dtype);
    float l = rnd((float)log2((double)s), dtype);  /*                 .log2()             */
    /* log2 of the fp32 value: correctly rounded via double, then to the dtype */
    return ceilf(l);                               /*                 .ceil()             */
}

/* quantize one block in place (values are fp32 images of dtype values) */
static void quantize_block(float* blk, int n, int mant_bits, float eps, int dtype, float* exp_out)
{
    float e = block_exponent(blk, n, eps, dtype);
    if (exp_out) *exp_out = e;
    float em = rnd(e - (float)mant_bits, dtype);                   /* exp - mant_bits            :38 */
    float interval = rnd((float)pow(2.0, (double)em), dtype);      /* torch.pow(2.0, exp-mant)   :38 */
    float p2e = rnd((float)pow(2.0, (double)e), dtype);            /* torch.pow(2.0, exp)        :39 */
    float max_v = rnd(p2e - interval, dtype);                      /*            - interval      :39 */
    for (int i = 0; i < n; i++) {
        float q = rnd(blk[i] / interval, dtype);                   /* t / interval               :40 */
        float r = rnd(rintf(q), dtype);                            /* t.round() (half to even)   :25 */
        float y = rnd(r * interval, dtype);                        /* rounded *= interval        :42 */
        blk[i] = t_min(t_max(y, -max_v), max_v);                   /* min(max(.., -max_v), max_v):44 */
    }
}

/* a4: _no_sparsity_float_to_bfp (bfp_ops.py:46-59): pad last dim with zeros to a multiple of
 * block_size, quantize each block, cut the pad off.  exps (optional) receives one float per block,
 * rows * ceil(cols / block) entries. */
int oracle_bfp_quantize(const void* in, void* out, int64_t rows, int64_t cols, int dtype,
                        int block_size, int mant_bits, double epsilon, float* exps)
{
    if (block_size <= 0 || rows < 0 || cols < 0 || dtype < 0 || dtype > 2) return -1;
    int64_t nblk = (cols + block_size - 1) / block_size;
    float eps = (float)epsilon;
    #pragma omp parallel
    {
        float* blk = (float*)malloc(sizeof(float) * (size_t)block_size);
        #pragma omp for schedule(static)
        for (int64_t r = 0; r < rows; r++) {
            for (int64_t b = 0; b < nblk; b++) {
                int64_t c0 = b * block_size;
                int64_t n = cols - c0 < block_size ? cols - c0 : block_size;
                for (int64_t i = 0; i < n; i++) blk[i] = ld(in, r * cols + c0 + i, dtype);
                for (int64_t i = n; i < block_size; i++) blk[i] = 0.0f;       /* F.pad :52 */
                float e;
                quantize_block(blk, block_size, mant_bits, eps, dtype, &e);
                if (exps) exps[r * nblk + b] = e;
                for (int64_t i = 0; i < n; i++) st(out, r * cols + c0 + i, dtype, blk[i]);
            }
        }
        free(blk);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * libstdc++ std::nth_element over (|v|, index) pairs, comparator as ATen's topk(largest=False):
 *     comp(x, y) = (!isnan(x) && isnan(y)) || (x < y)
 * ---------------------------------------------------------------------------------------- */
typedef struct { float key; int64_t idx; } kv_t;

static inline int kv_lt(const kv_t* x, const kv_t* y)
{
    int xn = x->key != x->key, yn = y->key != y->key;
    return (!xn && yn) || (x->key < y->key);
}
static inline void kv_swap(kv_t* a, kv_t* b) { kv_t t = *a; *a = *b; *b = t; }

static void push_heap_(kv_t* first, int64_t hole, int64_t top, kv_t value)
{
    int64_t parent = (hole - 1) / 2;
    while (hole > top && kv_lt(&first[parent], &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void adjust_heap_(kv_t* first, int64_t hole, int64_t len, kv_t value)
{
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (kv_lt(&first[child], &first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value);
}

static void make_heap_(kv_t* first, int64_t len)
{
    if (len < 2) return;
    int64_t parent = (len - 2) / 2;
    for (;;) {
        kv_t value = first[parent];
        adjust_heap_(first, parent, len, value);
        if (parent == 0) return;
        parent--;
    }
}

/* __heap_select(first, middle, last) on a[first..last) */
static void heap_select_(kv_t* a, int64_t first, int64_t middle, int64_t last)
{
    make_heap_(a + first, middle - first);
    for (int64_t i = middle; i < last; i++) {
        if (kv_lt(&a[i], &a[first])) {
            /* __pop_heap(first, middle, i) */
            kv_t value = a[i];
            a[i] = a[first];
            adjust_heap_(a + first, 0, middle - first, value);
        }
    }
}

static void move_median_to_first_(kv_t* a, int64_t result, int64_t x, int64_t y, int64_t z)
{
    if (kv_lt(&a[x], &a[y])) {
        if (kv_lt(&a[y], &a[z])) kv_swap(&a[result], &a[y]);
        else if (kv_lt(&a[x], &a[z])) kv_swap(&a[result], &a[z]);
        else kv_swap(&a[result], &a[x]);
    } else if (kv_lt(&a[x], &a[z])) kv_swap(&a[result], &a[x]);
    else if (kv_lt(&a[y], &a[z])) kv_swap(&a[result], &a[z]);
    else kv_swap(&a[result], &a[y]);
}

static int64_t unguarded_partition_(kv_t* a, int64_t first, int64_t last, int64_t pivot)
{
    for (;;) {
        while (kv_lt(&a[first], &a[pivot])) first++;
        last--;
        while (kv_lt(&a[pivot], &a[last])) last--;
        if (!(first < last)) return first;
        kv_swap(&a[first], &a[last]);
        first++;
    }
}

static void insertion_sort_(kv_t* a, int64_t first, int64_t last)
{
    if (first == last) return;
    for (int64_t i = first + 1; i != last; i++) {
        if (kv_lt(&a[i], &a[first])) {
            kv_t val = a[i];
            memmove(&a[first + 1], &a[first], sizeof(kv_t) * (size_t)(i - first));
            a[first] = val;
        } else {
            kv_t val = a[i];
            int64_t hole = i, next = i - 1;
            while (kv_lt(&val, &a[next])) { a[hole] = a[next]; hole = next; next--; }
            a[hole] = val;
        }
    }
}

static void nth_element_(kv_t* a, int64_t n, int64_t nth)
{
    if (n == 0 || nth == n) return;
    int64_t first = 0, last = n;
    int64_t depth = 0;
    for (int64_t t = n; t > 1; t >>= 1) depth++;   /* __lg(n) */
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {
            heap_select_(a, first, nth + 1, last);
            kv_swap(&a[first], &a[nth]);
            return;
        }
        depth--;
        int64_t mid = first + (last - first) / 2;
        move_median_to_first_(a, first, first + 1, mid, last - 1);
        int64_t cut = unguarded_partition_(a, first + 1, last, first);
        if (cut <= nth) first = cut; else last = cut;
    }
    insertion_sort_(a, first, last);
}

/* test hook: prune mask of one slice of n values; mask[i] = 1 if element i is among the k pruned */
int oracle_topk_smallest_mask(const float* absvals, int64_t n, int64_t k, uint8_t* mask)
{
    memset(mask, 0, (size_t)n);
    if (k <= 0) return 0;
    if (k > n) return -1;
    kv_t* q = (kv_t*)malloc(sizeof(kv_t) * (size_t)n);
    for (int64_t i = 0; i < n; i++) { q[i].key = absvals[i]; q[i].idx = i; }
    nth_element_(q, n, k - 1);
    for (int64_t i = 0; i < k; i++) mask[q[i].idx] = 1;
    free(q);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a5: _structured_N_M_sparsity (bfp_ops.py:73-91): pad last dim to a multiple of M, per group of
 * M zero the (M-N) smallest |v| (ATen topk -> nth_element), cut the pad off.
 * Kept values are bit-unchanged, pruned ones become +0.
 * ---------------------------------------------------------------------------------------- */
int oracle_nm_sparsify(const void* in, void* out, int64_t rows, int64_t cols, int dtype, int N, int M)
{
    if (!(N > 0 && M > 0 && N <= M)) return -1;                       /* assert :74 */
    int64_t ngrp = (cols + M - 1) / M;
    int k = M - N;
    size_t es = esize(dtype);
    #pragma omp parallel
    {
        kv_t* q = (kv_t*)malloc(sizeof(kv_t) * (size_t)M);
        #pragma omp for schedule(static)
        for (int64_t r = 0; r < rows; r++) {
            for (int64_t g = 0; g < ngrp; g++) {
                int64_t c0 = g * M;
                int64_t n = cols - c0 < M ? cols - c0 : M;
                memcpy((char*)out + (size_t)(r * cols + c0) * es, (const char*)in + (size_t)(r * cols + c0) * es, (size_t)n * es);
                if (k == 0) continue;
                for (int64_t i = 0; i < M; i++) {
                    q[i].key = i < n ? fabsf(ld(in, r * cols + c0 + i, dtype)) : 0.0f;   /* F.pad :81 */
                    q[i].idx = i;
                }
                nth_element_(q, M, k - 1);
                for (int i = 0; i < k; i++)
                    if (q[i].idx < n) memset((char*)out + (size_t)(r * cols + c0 + q[i].idx) * es, 0, es);
            }
        }
        free(q);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a6: _unstructured_sparsity (bfp_ops.py:61-71): one global topk over the flattened tensor,
 * k = int(numel * frac).  Sequential by nature (single introselect over numel pairs).
 * tau_out (optional): fp32 value of the k-th smallest |v| (the threshold).
 * ---------------------------------------------------------------------------------------- */
int oracle_unstructured_sparsify(const void* in, void* out, int64_t numel, int dtype, double frac,
                                 float* tau_out, int64_t* k_out)
{
    if (!(frac > 0)) return -1;                                        /* assert :62 */
    int64_t k = (int64_t)((double)numel * frac);                       /* int(temp.shape[1]*frac) :66 */
    if (k > numel) return -2;                                          /* topk would raise */
    size_t es = esize(dtype);
    memcpy(out, in, (size_t)numel * es);
    if (k_out) *k_out = k;
    if (k == 0) { if (tau_out) *tau_out = -1.0f; return 0; }
    kv_t* q = (kv_t*)malloc(sizeof(kv_t) * (size_t)numel);
    if (!q) return -3;
    for (int64_t i = 0; i < numel; i++) { q[i].key = fabsf(ld(in, i, dtype)); q[i].idx = i; }
    nth_element_(q, numel, k - 1);
    if (tau_out) *tau_out = q[k - 1].key;
    for (int64_t i = 0; i < k; i++) memset((char*)out + (size_t)q[i].idx * es, 0, es);
    free(q);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a7-a9: float_to_bfp_blocked (bfp_ops.py:124-149) for rounding_mode='determ'.
 *   sparsity_mode: 0 = none, 1 = structured N:M, 2 = unstructured
 *   quantize:      0 = 'fp32' (identity), 1 = 'bfp'
 *   sparsify_first: 1 -> Q(S(t))  (first == 's'),  0 -> S(Q(t))
 * The tensor is [rows, cols] (any leading dims flattened into rows; blocks and groups run along
 * the last dim only, Appendix A.1/A.4 of SURVEY.md).
 * ---------------------------------------------------------------------------------------- */
int oracle_float_to_bfp_blocked(const void* in, void* out, int64_t rows, int64_t cols, int dtype,
                                int quantize, int block_size, int mant_bits, double epsilon,
                                int sparsity_mode, int N, int M, double frac, int sparsify_first)
{
    size_t bytes = (size_t)(rows * cols) * esize(dtype);
    void* tmp = malloc(bytes ? bytes : 1);
    if (!tmp) return -3;
    int rc = 0;
    const void* cur = in;
    for (int stage = 0; stage < 2 && rc == 0; stage++) {
        int do_sparsify = (stage == 0) == (sparsify_first != 0);
        void* dst = (stage == 0) ? tmp : out;
        if (do_sparsify) {
            if (sparsity_mode == 1) rc = oracle_nm_sparsify(cur, dst, rows, cols, dtype, N, M);
            else if (sparsity_mode == 2) rc = oracle_unstructured_sparsify(cur, dst, rows * cols, dtype, frac, 0, 0);
            else memcpy(dst, cur, bytes);
        } else {
            if (quantize == 1) rc = oracle_bfp_quantize(cur, dst, rows, cols, dtype, block_size, mant_bits, epsilon, 0);
            else memcpy(dst, cur, bytes);
        }
        cur = dst;
    }
    free(tmp);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * 'int' format: _quantize's branch bfp_ops.py:111-120 -> int_ops.Quantizer with configure() defaults
 * (perchannel=True, sym=True, mse=False, grouprows=1; int_ops.py:18-31), find_params (:33-115) and
 * quantize (:6-8, :117-120).  The tensor is viewed as [outer, C, inner] with the channel in the middle:
 *   weight (identifier 'w'):  x.flatten(1)            -> outer = 1, C = shape[0], inner = rest   (:39-42)
 *   2-D / 3-D activation:     reshape(-1, C).t()      -> outer = rows, C = last dim, inner = 1   (:47-50)
 *   4-D activation:           permute(1,0,2,3)        -> outer = N, C = shape[1], inner = H*W    (:44-46)
 * min/max are taken against an fp32 zero tensor (:54-56), which promotes everything after it to fp32:
 * the result is an fp32 tensor whatever the input dtype.
 * ---------------------------------------------------------------------------------------- */
int oracle_int_quantize(const void* in, float* out, int64_t outer, int64_t C, int64_t inner, int dtype, int bits)
{
    if (outer < 0 || C < 0 || inner < 0 || dtype < 0 || dtype > 2 || bits < 0 || bits > 30) return -1;
    const float maxq = (float)((1u << bits) - 1u);                       /* 2**bits - 1            :24 */
    const float zero = (maxq + 1.0f) / 2.0f;                             /* (maxq + 1) / 2         :69 */
    #pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < C; c++) {
        float xmin = 0.0f, xmax = 0.0f;                                  /* minimum / maximum with zeros :55-56 */
        int nan = 0;
        for (int64_t o = 0; o < outer; o++)
            for (int64_t i = 0; i < inner; i++) {
                const float v = ld(in, (o * C + c) * inner + i, dtype);
                if (v != v) nan = 1;
                else { if (v < xmin) xmin = v; if (v > xmax) xmax = v; }
            }
        if (nan) { xmin = NAN; xmax = NAN; }                             /* torch min/max propagate NaN */
        xmax = t_max(fabsf(xmin), xmax);                                 /* sym :59 */
        if (xmin < 0) xmin = -xmax;                                      /* :60-62 */
        if (xmin == 0 && xmax == 0) { xmin = -1.0f; xmax = 1.0f; }       /* :63-65 */
        const float scale = (xmax - xmin) / maxq;                        /* :67 */
        for (int64_t o = 0; o < outer; o++)
            for (int64_t i = 0; i < inner; i++) {
                const int64_t idx = (o * C + c) * inner + i;
                const float x = ld(in, idx, dtype);
                float q = rintf(x / scale) + zero;                       /* round(x / scale) + zero :7 */
                q = t_min(t_max(q, 0.0f), maxq);                         /* clamp(., 0, maxq)       :7 */
                out[idx] = scale * (q - zero);                           /* scale * (q - zero)      :8 */
            }
    }
    return 0;
}

int oracle_version(void) { return 1; }
