// bfpq_quant_math.h -- scale derivation and packed-VALU helpers shared by the quantizer kernels (the exact power-of-two
// fast path and the lean 16-bit drop-in path; the step-by-step replay is in bfpq_common.h).
#pragma once
#include "bfpq_device.h"

namespace bfpq_dev {


__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t bfi_b32(uint32_t mask, uint32_t a, uint32_t b) { uint32_t d; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b)); return d; }
template <bool HI> __device__ __forceinline__ float fma_mix_f16(uint32_t a, float c)
{
    float d;                                                 // src0: the low / high half of a as fp16; src1 = 1.0, src2 = c in fp32
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
    return d;
}

// scale of a block on the branch-free path; ok == false -> the caller emulates step by step instead
#ifndef BFPQ_USE_BUF
#define BFPQ_USE_BUF 0              // A/B knob: drop-in instantiations address the streams through buffer descriptors.  14 VALU
#endif                              // instructions fewer per item (95 vs 109) and SLOWER: 32.7 vs 31.65 us on one box, interleaved
#ifndef BFPQ_BUF_DUMMY
#define BFPQ_BUF_DUMMY 1
#endif
struct FastScale { float inv, interval, qmax; int e; bool ok; };

template <int DT>
__device__ __forceinline__ FastScale fast_scale(uint32_t max_key, int mant_bits, float eps_dt, const uint8_t* s_win)
{
    using T = Traits<DT>;
    FastScale f;
    // max + epsilon rounded to dtype.  bf16: branch-free round-half-even on the bits; a NaN / inf sum
    // keeps an all-ones exponent (or carries into the sign bit) and fails the range test below
    uint32_t sb;
    if constexpr (DT == BFPQ_BF16) {
        const uint32_t u = f2u(raw_to_f32<DT>(max_key) + eps_dt);
        sb = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    } else sb = f2u(rnd<DT>(raw_to_f32<DT>(max_key) + eps_dt));
    const uint32_t kb = (sb >> 23) & 0x1ffu;                              // biased exponent (9 bits incl. sign: 0 here)
    const uint32_t mant = (sb >> (23 - T::MBITS)) & ((1u << T::MBITS) - 1u);
    const uint32_t win = s_win[(kb + 33u) & 511u];                        // table index k + 160, k = kb - 127
    const int eb = (int)kb + (mant > win ? 1 : 0);                        // biased shared exponent
    const int emb = eb - mant_bits;                                       // biased exponent of the interval
    bool ok = (kb >= 1u) && (kb <= 254u) && (emb >= 1) && (emb <= 253) && (eb <= 254);
    if constexpr (DT == BFPQ_F16) ok = ok && (emb >= 103) && (eb <= 142); // 2^-24 <= interval, 2^e finite in fp16
    f.ok = ok;
    f.interval = u2f((uint32_t)emb << 23);
    f.inv = u2f((uint32_t)(254 - emb) << 23);
    f.qmax = (float)((1u << mant_bits) - 1u);
    f.e = eb - 127;
    return f;
}

// The lean form of the same scale for 16-bit dtypes in drop-in mode ("hot16"), valid when the block max lies in the
// range [kb_lo, kb_lo + kb_span] of dtype exponent fields that the host derived (FusedArgs): there max + epsilon rounds
// back to max (epsilon below half an ulp), the interval and the magic constant below are normal numbers and nothing
// overflows, so the shared exponent is just "exponent field of the max, plus one if its mantissa field is above the window".
// Rounding then needs no division and no integer code at all:
//     out = sign(x) * ((min(|x|, max_v) + C) - C),   C = 1.5 * 2^23 * interval
// The sum is a multiple of ulp(C) = interval, rounded half-to-even by the adder -- the reference's round(x / interval)
// * interval -- and clamping the magnitude first equals clamping the rounded value (max_v is on the grid).  The clamp is a
// packed 16-bit integer min on the magnitude bits, the sign comes back with one bit-field insert per two elements.
struct Hot16 { uint32_t maxv2; float C; int e; bool ok; };

template <int DT>
__device__ __forceinline__ Hot16 hot16_scale(uint32_t max_key, const FusedArgs& a, const uint8_t* s_win)
{
    using T = Traits<DT>;
    constexpr uint32_t EOFF = DT == BFPQ_F16 ? 112u : 0u;                 // dtype exponent field -> fp32 exponent field
    Hot16 h;
    const uint32_t kb = max_key >> T::MBITS, mant = max_key & ((1u << T::MBITS) - 1u);
    const uint32_t win = s_win[kb + 33u + EOFF];                          // (kb <= 255: inside the 512-byte LDS copy)
    const uint32_t eb = kb + EOFF + (mant > win ? 1u : 0u);               // fp32-biased shared exponent
    h.ok = (kb - (uint32_t)a.kb_lo) <= (uint32_t)a.kb_span;
    h.e = (int)eb - 127;
    h.C = u2f(((eb - (uint32_t)a.mant_bits) << 23) + 0x0BC00000u);       // 1.5 * 2^(23 + e - m)
    const uint32_t maxv = (eb << T::MBITS) + a.maxv_c;                    // (2^m - 1) * 2^(e - m) in dtype bits
    h.maxv2 = maxv | (maxv << 16);
    return h;
}

}  // namespace bfpq_dev
