// bfpq_common.h -- dtype traits, bit-pattern helpers and the per-block HBFP scale derivation.
//
// Everything the reference does in "the tensor's dtype" (bfp_ops.py:29-44) is done here in fp32
// registers with an explicit round-to-dtype (rnd<DT>) wherever ATen's CPU kernels round, so the
// device result is bit-identical to the reference's CPU result for fp32, fp16 and bf16 inputs.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "bfpq.h"

namespace bfpq {

template <int DT> struct Traits;
template <> struct Traits<BFPQ_F32> {
    using raw_t = uint32_t;
    static constexpr int VEC = 4;              // elements per 16-byte lane access
    static constexpr int MBITS = 23;           // mantissa field width
    static constexpr int SIG = 24;             // significand bits (integers < 2^SIG are exact)
    static constexpr uint32_t ABS = 0x7fffffffu;
    static constexpr uint32_t INF = 0x7f800000u;
};
template <> struct Traits<BFPQ_F16> {
    using raw_t = uint16_t;
    static constexpr int VEC = 8;
    static constexpr int MBITS = 10;
    static constexpr int SIG = 11;
    static constexpr uint32_t ABS = 0x7fffu;
    static constexpr uint32_t INF = 0x7c00u;
};
template <> struct Traits<BFPQ_BF16> {
    using raw_t = uint16_t;
    static constexpr int VEC = 8;
    static constexpr int MBITS = 7;
    static constexpr int SIG = 8;
    static constexpr uint32_t ABS = 0x7fffu;
    static constexpr uint32_t INF = 0x7f80u;
};

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

template <int DT> __device__ __forceinline__ float raw_to_f32(uint32_t raw)
{
    if constexpr (DT == BFPQ_F32) return u2f(raw);
    else if constexpr (DT == BFPQ_BF16) return u2f(raw << 16);
    else return __half2float(__ushort_as_half((unsigned short)raw));
}

// fp32 -> dtype bits, round-to-nearest-even (what ATen does when it stores an fp32 intermediate)
template <int DT> __device__ __forceinline__ uint32_t f32_to_raw(float f)
{
    if constexpr (DT == BFPQ_F32) return f2u(f);
    else if constexpr (DT == BFPQ_BF16) {
        uint32_t u = f2u(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    } else {
        // Keep the conversion a plain v_cvt_f16_f32: hipcc (ROCm 7.2) otherwise folds a preceding
        // multiply into v_fma_mixlo_f16 a, b, +0, and (-0)*b + (+0) = +0 loses the sign of a zero
        // product that the reference keeps (fixture G7, fp16 row "near fp16 max", element 7).
        asm volatile("" : "+v"(f));
        return (uint32_t)__half_as_ushort(__float2half_rn(f));
    }
}

template <int DT> __device__ __forceinline__ float rnd(float f)
{
    if constexpr (DT == BFPQ_F32) return f;
    else return raw_to_f32<DT>(f32_to_raw<DT>(f));
}

// magnitude key: integer order of keys == the reference comparator's order of |v|
// (ATen topk treats every NaN as one largest value: NaNs collapse to INF+1)
template <int DT> __device__ __forceinline__ uint32_t mag_key(uint32_t raw)
{
    uint32_t k = raw & Traits<DT>::ABS;
    return k > Traits<DT>::INF ? Traits<DT>::INF + 1u : k;
}

// elementwise torch.maximum / torch.minimum: NaN propagates
__device__ __forceinline__ float t_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
__device__ __forceinline__ float t_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }

// ---------------------------------------------------------------------------------------------
// Per-block scale.  mode: 0 = exact power-of-two fast path, 1 = step-by-step emulation,
// 2 = the reference turns the whole block into NaN (non-finite max, or log2(0) = -inf).
// ---------------------------------------------------------------------------------------------
struct BlockScale {
    float inv;        // 2^-(e-m)            (mode 0)
    float interval;   // rnd(2^(e-m))
    float max_v;      // rnd(rnd(2^e) - interval)
    float qmax;       // 2^m - 1
    int e;            // shared exponent
    int mode;
};

// get_exponent (bfp_ops.py:29-33) for a block whose max |v| has magnitude bits max_key (un-collapsed:
// a NaN max shows as > INF), then the scale constants of _convert_blocked_float_to_bfp (:38-39).
// exp_win: BFPQ_EXP_WIN_ENTRIES bytes (LDS or global), see bfpq_exp_window_host.
template <int DT>
__device__ __forceinline__ BlockScale block_scale(uint32_t max_key, int mant_bits, float eps_dt, const uint8_t* exp_win)
{
    using T = Traits<DT>;
    BlockScale b;
    b.qmax = (float)((1u << mant_bits) - 1u);
    b.inv = 0.f; b.interval = 0.f; b.max_v = 0.f; b.e = -128; b.mode = 2;
    if (max_key > T::INF) return b;                                  // NaN in the block
    const float s = rnd<DT>(raw_to_f32<DT>(max_key) + eps_dt);        // max + epsilon, in dtype
    uint32_t sb = f2u(s);
    if (sb == 0u || sb >= 0x7f800000u) return b;                     // log2(0) = -inf, or inf: block -> NaN
    int k = (int)(sb >> 23) - 127;
    uint32_t mant = sb & 0x7fffffu;
    if ((sb >> 23) == 0u) {                                          // fp32 subnormal (only with epsilon == 0)
        const int lz = __clz((int)mant) - 9;                         // leading zeros inside the 23-bit field (bits 22..0)
        mant = (mant << (lz + 1)) & 0x7fffffu;
        k = -127 - lz;
    }
    const uint32_t win = exp_win[k + 160];
    const int e = k + ((mant >> (23 - T::MBITS)) > win ? 1 : 0);
    const int em = e - mant_bits;
    const float p2em = ldexpf(1.0f, em);
    b.e = e;
    b.interval = rnd<DT>(p2em);
    // interval underflows to 0 in dtype (fp16: block max below 2^(m - 24)): the reference divides by it -- 0 / 0 and
    // (x / 0) * 0 are NaN for every element -- so this is a NaN block as well, and the packed exponent must say so
    if (b.interval == 0.f) { b.e = -128; return b; }
    const float p2e = rnd<DT>(ldexpf(1.0f, e));
    b.max_v = rnd<DT>(p2e - b.interval);
    const bool fast = (mant_bits <= T::SIG) && (b.interval == p2em) && (em >= -126) && (em <= 126) &&
                      (f2u(p2e) < 0x7f800000u);
    b.mode = fast ? 0 : 1;
    b.inv = fast ? ldexpf(1.0f, -em) : 0.f;
    return b;
}

// counter-based uniform in [0,1) with 24 bits, keyed by (seed, element index): one round of a
// 32-bit integer mixer (multiply / xor-shift) over the index -- no state.  rng_item_key() does the 64-bit
// part once per lane item, uniform24k() the 7-op per-element part.
__device__ __forceinline__ uint32_t rng_item_key(uint64_t seed, uint64_t idx0)
{
    return (uint32_t)idx0 * 0x9E3779B9u + (uint32_t)seed + (uint32_t)(idx0 >> 32) * 0x85EBCA6Bu + (uint32_t)(seed >> 32);
}
__device__ __forceinline__ float uniform24k(uint32_t key, uint32_t j)
{
    uint32_t x = key + j * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return (float)(x >> 8) * 5.9604644775390625e-08f;
}
__device__ __forceinline__ float uniform24(uint64_t seed, uint64_t idx) { return uniform24k(rng_item_key(seed, idx), 0u); }

// One element through _convert_blocked_float_to_bfp (:40-44).  Returns the dequantised value
// (exactly representable in dtype) and the integer mantissa in *code.
// dither is only added when stoch (adding +0.0 would turn a -0.0 quotient into +0.0).
template <int DT>
__device__ __forceinline__ float quant_elem(float x, const BlockScale& b, bool stoch, float dither, float* code)
{
    if (b.mode == 0) {
        float t = x * b.inv;
        if (stoch) t += dither;
        float q = rintf(t);
        q = fminf(fmaxf(q, -b.qmax), b.qmax);
        *code = q;
        return q * b.interval;
    }
    if (b.mode == 2) { *code = 0.f; return u2f(0x7fc00000u); }
    float q = rnd<DT>(x / b.interval);
    if (stoch) q += dither;
    const float r = rnd<DT>(rintf(q));
    const float y = rnd<DT>(r * b.interval);
    const float c = fminf(fmaxf(r, -b.qmax), b.qmax);
    *code = (c == c) ? c : 0.f;
    return t_min(t_max(y, -b.max_v), b.max_v);
}

__device__ __forceinline__ int8_t sat_exp(const BlockScale& b)
{
    if (b.mode == 2) return (int8_t)-128;
    return (int8_t)(b.e < -127 ? -127 : (b.e > 127 ? 127 : b.e));
}

}  // namespace bfpq
