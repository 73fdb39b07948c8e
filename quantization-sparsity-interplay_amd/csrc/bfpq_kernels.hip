// bfpq_kernels.hip -- gfx950 kernels + C-ABI launchers of libbfpq.so (see include/bfpq.h).
//
// Reference path replaced: src/transformers/bfp/bfp_ops.py:16-149.
//
// Kernels
//   k_fused_flat   the hot kernel: [N:M mask] + shared exponent + mantissa rounding in ONE pass.
//                  16 B per lane per access, a block of `block_size` elements lives in `lpb`
//                  adjacent lanes of one wavefront (block 64 bf16 = 8 lanes), block max by
//                  DPP, N:M keep-mask from a 729-entry LDS table, no atomics, no LDS staging of data,
//                  no second read.  HBM-bound: 2 x sizeof(dtype) B/elem.  NM == -1: global magnitude
//                  threshold (the apply launch of the unstructured path, bfpq_unstructured.hip).
//   k_fused_batched  the same item pipeline over a list of tensors (descriptors in the kernel arguments)
//   k_nm_rows      general N:M (any M <= 64, ragged rows): one thread per group, the group's
//                  (key,index) pairs in an LDS column, libstdc++ introselect replayed (nm_select.h)
//   k_quant_rows_tiled / _vec / k_quant_rows   HBFP quantizers for ragged rows and odd block sizes
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <type_traits>
#include "bfpq.h"
#include "bfpq_common.h"
#include "nm_select.h"

using namespace bfpq;

extern "C" __attribute__((visibility("hidden"))) int bfpq_g_gemm_rt;   // bfpq_gemm.hip

#include "bfpq_device.h"

extern "C" { __attribute__((visibility("hidden"))) int bfpq_g_max_grid = BFPQ_MAXGRID; }

using namespace bfpq_dev;

namespace {

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
struct Hot16 { uint32_t maxv2; float C; bool ok; };

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
    h.C = u2f(((eb - (uint32_t)a.mant_bits) << 23) + 0x0BC00000u);       // 1.5 * 2^(23 + e - m)
    const uint32_t maxv = (eb << T::MBITS) + a.maxv_c;                    // (2^m - 1) * 2^(e - m) in dtype bits
    h.maxv2 = maxv | (maxv << 16);
    return h;
}

// ---------------------------------------------------------------------------------------------
// k_fused_flat: the tensor is a flat array of 16-byte lane items; rows do not matter because
// cols % block == 0 (and cols % M == 0).  NM in {0,2,4}.  One HBM read, one HBM write per output.
//   hot path per item: [packed 3-way comparisons -> 729-entry LDS table -> AND masks], packed abs/max,
//   DPP group max, exponent from a 320-byte LDS table, mul / rndne / med3 / mul per element, pack.
//   Anything unusual in a block (non-finite or zero max, scale outside the normal range, mantissa
//   wider than the dtype) makes the whole wavefront replay that item through the step-by-step
//   emulation (quant_elem); the branch is wave-uniform and never taken on ordinary weights.
// ---------------------------------------------------------------------------------------------
// A list of tensors in one launch (bfpq_fake_quantize_batched): up to kMaxBatch descriptors travel in the kernel arguments
// (no device memory, graph-capturable).  The tensors share dtype / block / mantissa width; each is a flat array of lane
// items cut into chunks of 256 (the last chunk of a tensor is ragged), the chunks of all tensors form one index space
// that the workgroups stride over.  flags bit 0: apply the N:M mask to this tensor (a Linear's weight) or not (its activation).
constexpr int kMaxBatch = 64;
struct BatchDesc { const void* in; void* out; int64_t n_items; uint32_t chunk0; uint32_t flags; };
struct BatchArgs { int n; uint32_t total_chunks; BatchDesc d[kMaxBatch]; };

template <int DT, int NM, bool SFIRST, bool STOCH, int LPBT, bool DEQ_ONLY, bool BATCHED>
__device__ __forceinline__ void fused_flat_body(const FusedArgs& a, [[maybe_unused]] const BatchArgs* b)
{
    using T = Traits<DT>;
    constexpr int VEC = T::VEC;
    __shared__ __attribute__((aligned(16))) uint8_t s_win[512];
    __shared__ uint2 s_mask[NM == 4 && VEC == 8 ? 736 : 1];      // 16-bit dtypes: AND masks for the two dwords of a group
    __shared__ uint8_t s_keep[NM == 4 && VEC == 4 ? 736 : 1];    // fp32: 4-bit keep mask
    __shared__ uint64_t s_kv[NM == 8 ? 8 * kThreads : 1];        // N:8: column per thread for the nth_element replay (rare)
    // (the tables are filled further down, behind the first tile's load: one memory round trip for everything)

    const bool do_quant = a.lpb > 0;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    const int64_t n_round = (a.n_items + kThreads - 1) / kThreads * kThreads;   // uniform trip count per block
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(a.in);

    // N:M mask on the 4 dwords of an item (16-bit dtypes: 2 groups of 4; fp32: 1 group)
    ThrCtx thr;
    bool item_valid = true;
    int64_t item_index = 0;
    auto nm_mask = [&](uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3) __attribute__((always_inline)) {
        if constexpr (NM == -1) {                       // global magnitude threshold (unstructured, bfp_ops.py:61-71)
            const int64_t tile0 = uniform64(item_index - (threadIdx.x & 63));      // the wave's 64 lanes hold one aligned tile
            const bool ranked = tile0 >= thr.rs && tile0 < thr.re;
            if (__builtin_expect(!ranked, 1)) {
                // every tile but (normally) one: the ties of this tile all go (in front of the cut) or all stay, i.e. one
                // comparison against tau + 1 or tau
                const uint32_t teff = thr.tau + (tile0 < thr.rs ? 1u : 0u);     // (tau == 0 when nothing is pruned at all)
                if constexpr (VEC == 8) {
                    // packed: keys <= 0x7f81 and teff <= 0x7f82, so teff - 1 - key fits 16 signed bits; its sign says keep
                    const uint32_t absm = T::ABS | (T::ABS << 16), nanc = (T::INF + 1u) | ((T::INF + 1u) << 16);
                    const uint32_t tm1 = (teff - 1u) & 0xffffu, t2 = tm1 | (tm1 << 16);
                    auto keep = [&](uint32_t d) { return pk_ashr_i16_s(pk_sub_i16(t2, pk_min_i16_s(d & absm, nanc)), 0x000f000fu); };
                    d0 &= keep(d0); d1 &= keep(d1); d2 &= keep(d2); d3 &= keep(d3);
                } else {
                    d0 = mag_key<DT>(d0) < teff ? 0u : d0; d1 = mag_key<DT>(d1) < teff ? 0u : d1;
                    d2 = mag_key<DT>(d2) < teff ? 0u : d2; d3 = mag_key<DT>(d3) < teff ? 0u : d3;
                }
            } else {
                uint32_t raw[VEC];
                if constexpr (VEC == 4) { raw[0] = d0; raw[1] = d1; raw[2] = d2; raw[3] = d3; }
                else {
                    raw[0] = d0 & 0xffffu; raw[1] = d0 >> 16; raw[2] = d1 & 0xffffu; raw[3] = d1 >> 16;
                    raw[4] = d2 & 0xffffu; raw[5] = d2 >> 16; raw[6] = d3 & 0xffffu; raw[7] = d3 >> 16;
                }
                const uint32_t prune = thr_prune_bits<DT, true>(raw, item_valid, item_index, thr);
                if constexpr (VEC == 4) {
                    d0 = (prune & 1u) ? 0u : d0; d1 = (prune & 2u) ? 0u : d1; d2 = (prune & 4u) ? 0u : d2; d3 = (prune & 8u) ? 0u : d3;
                } else {
                    auto m = [](uint32_t pr) { return ((pr & 1u) ? 0u : 0xffffu) | ((pr & 2u) ? 0u : 0xffff0000u); };
                    d0 &= m(prune); d1 &= m(prune >> 2); d2 &= m(prune >> 4); d3 &= m(prune >> 6);
                }
            }
        } else if constexpr (NM == 8 && VEC == 4) {
            // fp32: a group of 8 is two adjacent lane items (even lane: elements 0-3, odd lane: 4-7; item parity = lane
            // parity because the sweep stride is a multiple of 256).  The partner's four keys come over by DPP (swap of
            // adjacent lanes); every lane then counts, for its own four elements, the smaller and the equal keys among
            // all eight, and the group decides as in the 16-bit path: certainly pruned / certainly kept / look the weak
            // ordering up (or replay nth_element) when ties straddle the cut.
            auto swp = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); };
            const uint32_t own[4] = {mag_key<DT>(d0), mag_key<DT>(d1), mag_key<DT>(d2), mag_key<DT>(d3)};
            const uint32_t oth[4] = {swp(own[0]), swp(own[1]), swp(own[2]), swp(own[3])};
            const bool odd = (threadIdx.x & 1) != 0;
            const int P = 8 - a.N;
            uint32_t less4 = 0, prune = 0;
            bool amb = false;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int less = 0, eq = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    less += (own[j] < own[i]) + (oth[j] < own[i]);
                    eq += (own[j] == own[i]) + (oth[j] == own[i]);
                }
                less4 |= (uint32_t)less << (3 * i);
                if (less + eq <= P) prune |= 1u << i;
                else if (less < P) amb = true;
            }
            const uint32_t amb_group = (uint32_t)amb | swp((uint32_t)amb);
            if (amb_group) {
                const uint32_t other_less = swp(less4);
                const uint32_t lo = odd ? other_less : less4, hi = odd ? less4 : other_less;   // elements 0-3 | 4-7
                uint32_t mask8;
                if (a.nm_lut) mask8 = a.nm_lut[lo | (hi << 12)];
                else {
                    KvView view{s_kv + threadIdx.x, kThreads};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        view.set(odd ? 4 + i : i, ((uint64_t)own[i] << 8) | (uint64_t)(odd ? 4 + i : i));
                        view.set(odd ? i : 4 + i, ((uint64_t)oth[i] << 8) | (uint64_t)(odd ? i : 4 + i));
                    }
                    mask8 = (uint32_t)nm_prune_mask(view, a.N, 8);
                }
                prune = odd ? (mask8 >> 4) & 0xfu : mask8 & 0xfu;
            }
            d0 = (prune & 1u) ? 0u : d0; d1 = (prune & 2u) ? 0u : d1; d2 = (prune & 4u) ? 0u : d2; d3 = (prune & 8u) ? 0u : d3;
        } else if constexpr (NM == 8) {
            // one lane item = one group of 8 (16-bit dtypes).  Count, per element, the smaller and the equal keys (28 pair
            // comparisons in registers): less + equal <= P -> certainly pruned, less >= P -> certainly kept; only a group
            // whose ties straddle the cut needs the reference's tie order (libstdc++ nth_element replay on an LDS column)
            // Packed: the item's four dwords ARE the key pairs.  c = clamp(k_i - k_j, -1, 1) for two pairs per instruction;
            // per element S = sum_j c = 2 less + eq - 8 and A = sum_j c^2 = 8 - eq, so
            //   pruned  <=> less + eq <= P <=> S - A <= 2P - 16,      certainly kept <=> less >= P <=> S + A >= 2P
            const uint32_t absm = T::ABS | (T::ABS << 16), nanc = (T::INF + 1u) | ((T::INF + 1u) << 16);
            uint32_t D[4] = {d0, d1, d2, d3}, K[4], Kr[4], S[4] = {0, 0, 0, 0}, A[4] = {0, 0, 0, 0};
            auto rot = [](uint32_t x) { return __builtin_amdgcn_alignbit(x, x, 16); };
            auto c3 = [&](uint32_t x, uint32_t y) { return pk_min_i16_s(pk_max_i16_s(pk_sub_i16(x, y), 0xffffffffu), 0x00010001u); };
#pragma unroll
            for (int x = 0; x < 4; x++) { K[x] = pk_min_i16_s(D[x] & absm, nanc); Kr[x] = rot(K[x]); }
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint32_t c = c3(K[x], Kr[x]);                       // the two elements of one dword against each other
                S[x] = pk_add_i16(S[x], c); A[x] = pk_mad_i16(c, c, A[x]);
#pragma unroll
                for (int y = x + 1; y < 4; y++) {
                    const uint32_t c1 = c3(K[x], K[y]);                   // (x0 - y0, x1 - y1)
                    S[x] = pk_add_i16(S[x], c1); A[x] = pk_mad_i16(c1, c1, A[x]);
                    S[y] = pk_sub_i16(S[y], c1); A[y] = pk_mad_i16(c1, c1, A[y]);
                    const uint32_t c2 = c3(K[x], Kr[y]), c2r = rot(c2);   // (x0 - y1, x1 - y0); rotated: indexed by y's lanes
                    S[x] = pk_add_i16(S[x], c2); A[x] = pk_mad_i16(c2, c2, A[x]);
                    S[y] = pk_sub_i16(S[y], c2r); A[y] = pk_mad_i16(c2r, c2r, A[y]);
                }
            }
            const int P = 8 - a.N;
            const uint32_t cp = (uint32_t)((2 * P - 16) & 0xffff) * 0x10001u, pp = (uint32_t)(2 * P) * 0x10001u;
            uint32_t keep[4], amb = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                keep[x] = pk_ashr_i16_s(pk_sub_i16(pk_add_i16(A[x], cp), S[x]), 0x000f000fu);        // 0xffff where S - A > 2P - 16
                amb |= keep[x] & pk_ashr_i16_s(pk_sub_i16(pk_add_i16(S[x], A[x]), pp), 0x000f000fu);  // ... and S + A < 2P
            }
            if (amb) {                                                    // ties straddle the cut: the reference's tie order decides
                uint32_t prune;
                if (a.nm_lut) {
                    // what nth_element does depends only on the weak ordering of the 8 keys, i.e. on the vector of
                    // "number of smaller keys" less_i = (S_i + A_i) / 2: 8 x 3 bits index a host-built 16 MiB table
                    // (bfpq_nm8_lut_host) that only these few lanes touch
                    uint32_t idx = 0;
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const uint32_t l2 = pk_add_i16(S[x], A[x]);       // 2 * less per half, 0..14
                        idx |= (((l2 >> 1) & 7u) | (((l2 >> 17) & 7u) << 3)) << (6 * x);
                    }
                    prune = a.nm_lut[idx];
                } else {
                    KvView view{s_kv + threadIdx.x, kThreads};
#pragma unroll
                    for (int i = 0; i < 8; i++) view.set(i, ((uint64_t)((K[i >> 1] >> (16 * (i & 1))) & 0xffffu) << 8) | (uint64_t)i);
                    prune = (uint32_t)nm_prune_mask(view, a.N, 8);
                }
#pragma unroll
                for (int x = 0; x < 4; x++)
                    keep[x] = (((prune >> (2 * x)) & 1u) ? 0u : 0xffffu) | (((prune >> (2 * x + 1)) & 1u) ? 0u : 0xffff0000u);
            }
            d0 &= keep[0]; d1 &= keep[1]; d2 &= keep[2]; d3 &= keep[3];
        } else if constexpr (NM == 4 && VEC == 8) {
            // A_i = (group0.elem_i | group1.elem_i << 16): both groups go through one packed instruction
            const uint32_t absm = T::ABS | (T::ABS << 16), nanc = (T::INF + 1u) | ((T::INF + 1u) << 16);
            auto key = [&](uint32_t hi, uint32_t lo, uint32_t sel) {      // keys <= 0x7fff: signed min == unsigned min
                return pk_min_i16_s(__builtin_amdgcn_perm(hi, lo, sel) & absm, nanc);
            };
            const uint32_t k0 = key(d2, d0, 0x05040100u), k1 = key(d2, d0, 0x07060302u);
            const uint32_t k2 = key(d3, d1, 0x05040100u), k3 = key(d3, d1, 0x07060302u);
            // 3-way comparison of both groups at once: clamp(k_i - k_j, -1, 1); index = 364 + sum c_p 3^p
            auto c3 = [&](uint32_t x, uint32_t y) { return pk_min_i16_s(pk_max_i16_s(pk_sub_i16(x, y), 0xffffffffu), 0x00010001u); };
            uint32_t ip = 0x016c016cu;
            ip = pk_mad_i16_s(c3(k0, k1), 0x00010001u, ip);
            ip = pk_mad_i16_s(c3(k0, k2), 0x00030003u, ip);
            ip = pk_mad_i16_s(c3(k0, k3), 0x00090009u, ip);
            ip = pk_mad_i16_s(c3(k1, k2), 0x001b001bu, ip);
            ip = pk_mad_i16_s(c3(k1, k3), 0x00510051u, ip);
            ip = pk_mad_i16_s(c3(k2, k3), 0x00f300f3u, ip);
            const uint2 m0 = s_mask[ip & 0xffffu], m1 = s_mask[ip >> 16];
            d0 &= m0.x; d1 &= m0.y; d2 &= m1.x; d3 &= m1.y;
        } else if constexpr (NM == 4) {
            const uint32_t keep = s_keep[nm4_index(mag_key<DT>(d0), mag_key<DT>(d1), mag_key<DT>(d2), mag_key<DT>(d3))];
            d0 = (keep & 1u) ? d0 : 0u; d1 = (keep & 2u) ? d1 : 0u; d2 = (keep & 4u) ? d2 : 0u; d3 = (keep & 8u) ? d3 : 0u;
        } else if constexpr (NM == 2 && VEC == 8) {
            auto pair = [&](uint32_t& d) {
                const uint32_t keep = nm2_keep(mag_key<DT>(d & 0xffffu), mag_key<DT>(d >> 16), a.N);
                d &= ((keep & 1u) ? 0xffffu : 0u) | ((keep & 2u) ? 0xffff0000u : 0u);
            };
            pair(d0); pair(d1); pair(d2); pair(d3);
        } else if constexpr (NM == 2) {
            const uint32_t ka = nm2_keep(mag_key<DT>(d0), mag_key<DT>(d1), a.N), kb2 = nm2_keep(mag_key<DT>(d2), mag_key<DT>(d3), a.N);
            d0 = (ka & 1u) ? d0 : 0u; d1 = (ka & 2u) ? d1 : 0u; d2 = (kb2 & 1u) ? d2 : 0u; d3 = (kb2 & 2u) ? d3 : 0u;
        }
    };

    // One lane item.  GUARD = false in the main loop (every lane of the grid holds a real item: no
    // branch around any memory operation, so hipcc can emit counted vmcnt waits and keep the prefetches
    // and the previous store in flight); GUARD = true only in the ragged last sweep.
    // (batched mode: the tensor's own output pointer, item count and N:M switch; else the launch's)
    void* out_deq = a.out_deq;
    int64_t n_limit = a.n_items;
    [[maybe_unused]] bool nm_on = true;
    // (A/B knob BFPQ_USE_BUF, off: drop-in instantiations whose arithmetic does not depend on the position of an item address
    // the two streams through buffer descriptors -- the item offset is ONE 32-bit register advanced by one add per two sweeps,
    // the look-ahead distance sits in the instruction's scalar offset, out-of-range lanes read zeros / have their stores
    // dropped by the hardware: no 64-bit index arithmetic, no clamp, no guarded tail; a tensor above 3.9 GB is cut into
    // several launches by the host.  Measured slower than plain global loads/stores, see the knob.)
    constexpr bool USE_BUF = BFPQ_USE_BUF && DEQ_ONLY && !STOCH && NM != -1 && !BATCHED;
    [[maybe_unused]] uint4 buf_res;
    auto body = [&](auto guard_tag, const int64_t item, const uint4 cur) __attribute__((always_inline)) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        const bool valid = !GUARD || item < n_limit;
        item_valid = valid;
        item_index = item;
        uint32_t d0 = cur.x, d1 = cur.y, d2 = cur.z, d3 = cur.w;

#ifdef BFPQ_COPYONLY          /* A/B knob: same loop, loads and stores only (ceiling for this launch geometry) */
        if (valid && a.out_deq) stream_store(reinterpret_cast<uint4*>(a.out_deq) + item, make_uint4(d0, d1, d2, d3));
        return;
#endif
        if constexpr (NM != 0 && SFIRST) {                                     // S before Q (bfp_ops.py:141-144)
            if constexpr (BATCHED) { if (nm_on) nm_mask(d0, d1, d2, d3); }
            else nm_mask(d0, d1, d2, d3);
        }

        uint32_t o0 = d0, o1 = d1, o2 = d2, o3 = d3;
        float code[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) code[j] = 0.f;
        int e_blk = 0;
        bool nan_blk = false;
        if (do_quant) {
            // block max of |v| as integer max of magnitude bits
            uint32_t mx;
            if constexpr (VEC == 8) {
                const uint32_t absm = T::ABS | (T::ABS << 16);
                const uint32_t mp = pk_max_u16(pk_max_u16(d0 & absm, d1 & absm), pk_max_u16(d2 & absm, d3 & absm));
                mx = (mp & 0xffffu) > (mp >> 16) ? (mp & 0xffffu) : (mp >> 16);
            } else {
                const uint32_t m01 = (d0 & T::ABS) > (d1 & T::ABS) ? (d0 & T::ABS) : (d1 & T::ABS);
                const uint32_t m23 = (d2 & T::ABS) > (d3 & T::ABS) ? (d2 & T::ABS) : (d3 & T::ABS);
                mx = m01 > m23 ? m01 : m23;
            }
            mx = group_max<LPBT>(mx, a.lpb);
            bool hot = false;
            [[maybe_unused]] Hot16 h16;
            if constexpr (VEC == 8 && !STOCH && DEQ_ONLY) {
                h16 = hot16_scale<DT>(mx, a, s_win);
                hot = !__any(!h16.ok);
            }
            if (__builtin_expect(hot, 1)) {
                if constexpr (VEC == 8 && !STOCH && DEQ_ONLY) {
                    const uint32_t absm = T::ABS | (T::ABS << 16);
                    const uint32_t dd[4] = {d0, d1, d2, d3};
                    uint32_t oo[4];
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const uint32_t am = pk_min_u16(dd[x] & absm, h16.maxv2);          // clamped magnitudes of two elements
                        typedef float float2v __attribute__((ext_vector_type(2)));
                        const float2v C2 = {h16.C, h16.C};
                        float2v v;
                        if constexpr (DT == BFPQ_BF16) v = (float2v){u2f(am << 16), u2f(am & 0xffff0000u)} + C2;   // v_pk_add_f32
                        else v = (float2v){fma_mix_f16<false>(am, h16.C), fma_mix_f16<true>(am, h16.C)};            // (float)half + C, fused
                        v -= C2;
                        uint32_t pk;
                        if constexpr (DT == BFPQ_BF16) pk = __builtin_amdgcn_perm(f2u(v.y), f2u(v.x), 0x07060302u);  // exact: upper halves
                        else pk = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(v.x, v.y));                // exact in fp16
                        oo[x] = bfi_b32(absm, pk, dd[x]);                                      // magnitude from pk, signs from the input
                    }
                    o0 = oo[0]; o1 = oo[1]; o2 = oo[2]; o3 = oo[3];
                }
            } else {
                const FastScale fs = fast_scale<DT>(mx, a.mant_bits, a.eps_dt, s_win);
                e_blk = fs.e;
                uint32_t raw[VEC];
                if constexpr (VEC == 4) { raw[0] = d0; raw[1] = d1; raw[2] = d2; raw[3] = d3; }
                else {
                    raw[0] = d0 & 0xffffu; raw[1] = d0 >> 16; raw[2] = d1 & 0xffffu; raw[3] = d1 >> 16;
                    raw[4] = d2 & 0xffffu; raw[5] = d2 >> 16; raw[6] = d3 & 0xffffu; raw[7] = d3 >> 16;
                }
                const bool slow = (a.force_slow != 0) || !fs.ok;
                if (__builtin_expect(__any(slow), 0)) {
                    // cold: replay the reference's op sequence step by step (exact for every block)
                    const BlockScale bs = block_scale<DT>(mx, a.mant_bits, a.eps_dt, s_win);
                    e_blk = bs.e;
                    nan_blk = bs.mode == 2;
                    uint32_t outraw[VEC];
    #pragma unroll
                    for (int j = 0; j < VEC; j++) {
                        const float dither = STOCH ? uniform24k(rng_item_key(a.seed, (uint64_t)item * VEC), (uint32_t)j) - 0.5f : 0.f;
                        const float yv = quant_elem<DT>(raw_to_f32<DT>(raw[j]), bs, STOCH, dither, &code[j]);
                        outraw[j] = f32_to_raw<DT>(yv);
                    }
                    if constexpr (VEC == 4) { o0 = outraw[0]; o1 = outraw[1]; o2 = outraw[2]; o3 = outraw[3]; }
                    else {
                        o0 = outraw[0] | (outraw[1] << 16); o1 = outraw[2] | (outraw[3] << 16);
                        o2 = outraw[4] | (outraw[5] << 16); o3 = outraw[6] | (outraw[7] << 16);
                    }
                } else {
                    typedef float float2v __attribute__((ext_vector_type(2)));
                    float y[VEC];
                    [[maybe_unused]] const uint32_t rkey = STOCH ? rng_item_key(a.seed, (uint64_t)item * VEC) : 0u;
    #pragma unroll
                    for (int j = 0; j < VEC; j += 2) {                     // two elements per v_pk_mul_f32
                        float2v x;
                        if constexpr (DT == BFPQ_BF16) {
                            const uint32_t d = j < 2 ? d0 : (j < 4 ? d1 : (j < 6 ? d2 : d3));
                            x = (float2v){u2f(d << 16), u2f(d & 0xffff0000u)};
                        } else x = (float2v){raw_to_f32<DT>(raw[j]), raw_to_f32<DT>(raw[j + 1])};
                        float2v t = x * (float2v){fs.inv, fs.inv};
                        if constexpr (STOCH) {
                            t.x += uniform24k(rkey, (uint32_t)j) - 0.5f;
                            t.y += uniform24k(rkey, (uint32_t)j + 1u) - 0.5f;
                        }
                        float2v q = {__builtin_amdgcn_fmed3f(rintf(t.x), -fs.qmax, fs.qmax), __builtin_amdgcn_fmed3f(rintf(t.y), -fs.qmax, fs.qmax)};
                        code[j] = q.x; code[j + 1] = q.y;
                        const float2v yy = q * (float2v){fs.interval, fs.interval};
                        y[j] = yy.x; y[j + 1] = yy.y;
                    }
                    if constexpr (DT == BFPQ_F32) { o0 = f2u(y[0]); o1 = f2u(y[1]); o2 = f2u(y[2]); o3 = f2u(y[3]); }
                    else if constexpr (DT == BFPQ_BF16) {                 // exact: the bf16 image is the upper half
                        o0 = __builtin_amdgcn_perm(f2u(y[1]), f2u(y[0]), 0x07060302u);
                        o1 = __builtin_amdgcn_perm(f2u(y[3]), f2u(y[2]), 0x07060302u);
                        o2 = __builtin_amdgcn_perm(f2u(y[5]), f2u(y[4]), 0x07060302u);
                        o3 = __builtin_amdgcn_perm(f2u(y[7]), f2u(y[6]), 0x07060302u);
                    } else {                                               // exact in fp16: any rounding mode packs it
                        o0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y[0], y[1]));
                        o1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y[2], y[3]));
                        o2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y[4], y[5]));
                        o3 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(y[6], y[7]));
                    }
                }
            }
        }
        if constexpr (NM != 0 && !SFIRST) {                                   // Q before S (bfp_ops.py:146-149)
            if constexpr (BATCHED) { if (nm_on) nm_mask(o0, o1, o2, o3); }
            else nm_mask(o0, o1, o2, o3);
            if (a.out_codes) {                                                // a pruned element has code 0
                if constexpr (VEC == 4) {
                    code[0] = (o0 & T::ABS) ? code[0] : 0.f; code[1] = (o1 & T::ABS) ? code[1] : 0.f;
                    code[2] = (o2 & T::ABS) ? code[2] : 0.f; code[3] = (o3 & T::ABS) ? code[3] : 0.f;
                } else {
                    const uint32_t od[4] = {o0, o1, o2, o3};
#pragma unroll
                    for (int j = 0; j < VEC; j++) code[j] = ((od[j >> 1] >> (16 * (j & 1))) & T::ABS) ? code[j] : 0.f;
                }
            }
        }
        if constexpr (USE_BUF) { buf_res = make_uint4(o0, o1, o2, o3); return; }   // (the sweep stores it)
        if constexpr (DEQ_ONLY) {                                           // hot mode: exactly one store per item
            if (valid) stream_store(reinterpret_cast<uint4*>(out_deq) + item, make_uint4(o0, o1, o2, o3));
            return;
        }
        if (!valid) return;
        if (a.out_deq) stream_store(reinterpret_cast<uint4*>(a.out_deq) + item, make_uint4(o0, o1, o2, o3));
        if (a.out_codes) {
            int c[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j++) c[j] = (int)code[j];
            if (a.code_bits == 4) {
                uint32_t w = 0;
#pragma unroll
                for (int j = 0; j < VEC; j++) w |= ((uint32_t)c[j] & 0xfu) << (4 * j);
                if constexpr (VEC == 8) reinterpret_cast<uint32_t*>(a.out_codes)[item] = w;
                else reinterpret_cast<uint16_t*>(a.out_codes)[item] = (uint16_t)w;
            } else if (a.code_bits == 32) {                              // fp32 image of the dequantised values
                const uint32_t od[4] = {o0, o1, o2, o3};
                uint32_t f[VEC];
#pragma unroll
                for (int j = 0; j < VEC; j++)
                    f[j] = VEC == 4 ? od[j] : f2u(raw_to_f32<DT>((od[j >> 1] >> (16 * (j & 1))) & 0xffffu));
                uint4* dst = reinterpret_cast<uint4*>(a.out_codes) + item * (VEC / 4);
                dst[0] = make_uint4(f[0], f[1], f[2], f[3]);
                if constexpr (VEC == 8) dst[1] = make_uint4(f[4], f[5], f[6], f[7]);
            } else if (a.code_bits == 8) {
                uint32_t w0 = 0, w1 = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) w0 |= ((uint32_t)c[j] & 0xffu) << (8 * j);
                if constexpr (VEC == 8) {
#pragma unroll
                    for (int j = 0; j < 4; j++) w1 |= ((uint32_t)c[4 + j] & 0xffu) << (8 * j);
                    reinterpret_cast<uint2*>(a.out_codes)[item] = make_uint2(w0, w1);
                } else reinterpret_cast<uint32_t*>(a.out_codes)[item] = w0;
            } else {
                uint32_t w[VEC / 2];
#pragma unroll
                for (int j = 0; j < VEC / 2; j++) w[j] = ((uint32_t)c[2 * j] & 0xffffu) | (((uint32_t)c[2 * j + 1] & 0xffffu) << 16);
                if constexpr (VEC == 8) reinterpret_cast<uint4*>(a.out_codes)[item] = make_uint4(w[0], w[1], w[2], w[3]);
                else reinterpret_cast<uint2*>(a.out_codes)[item] = make_uint2(w[0], w[1]);
            }
        }
        if (a.out_exp && do_quant && (item % a.lpb) == 0) {
            const int es = e_blk < -127 ? -127 : (e_blk > 127 ? 127 : e_blk);
            a.out_exp[item / a.lpb] = nan_blk ? (int8_t)-128 : (int8_t)es;
        }
    };

    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    auto u4 = [](const u4v v) __attribute__((always_inline)) { return make_uint4(v.x, v.y, v.z, v.w); };
    if constexpr (BATCHED) {
        // chunk c of the list = chunk (c - chunk0[t]) of tensor t; a workgroup's chunks only move forward, so t does too
        __shared__ uint32_t s_chunk0[kMaxBatch + 1];
        if ((int)threadIdx.x <= b->n) s_chunk0[threadIdx.x] = (int)threadIdx.x < b->n ? b->d[threadIdx.x].chunk0 : b->total_chunks;
        __syncthreads();
        struct Cur { const uint4* in; void* out; int64_t n; int64_t item; bool nm; };
        // the descriptor of the current tensor is re-read (scalar loads from the kernel arguments) only when a chunk
        // crosses into another tensor: the address of the next load must not wait for a descriptor load every sweep
        int t = 0;
        const uint4* d_in = reinterpret_cast<const uint4*>(b->d[0].in);
        void* d_out = b->d[0].out;
        int64_t d_n = b->d[0].n_items;
        uint32_t d_c0 = b->d[0].chunk0, d_c1 = b->n > 1 ? b->d[1].chunk0 : b->total_chunks;
        bool d_nm = (b->d[0].flags & 1u) != 0;
        auto locate = [&](uint32_t c) __attribute__((always_inline)) {
            if (c >= d_c1) {                                   // (wave-uniform: c depends on blockIdx only)
                while (t + 1 < b->n && c >= s_chunk0[t + 1]) t++;
                t = __builtin_amdgcn_readfirstlane(t);
                const BatchDesc& d = b->d[t];
                d_in = reinterpret_cast<const uint4*>(d.in); d_out = d.out; d_n = d.n_items; d_nm = (d.flags & 1u) != 0;
                d_c0 = d.chunk0; d_c1 = s_chunk0[t + 1];
            }
            Cur r;
            r.in = d_in; r.out = d_out; r.n = d_n; r.nm = d_nm;
            r.item = (int64_t)(c - d_c0) * kThreads + threadIdx.x;
            return r;
        };
        auto fetchb = [&](const Cur& r) __attribute__((always_inline)) {     // (every chunk of a batched tensor is full: no clamp)
            return __builtin_nontemporal_load(reinterpret_cast<const u4v*>(r.in + r.item));
        };
        auto use = [&](const Cur& r) __attribute__((always_inline)) { out_deq = r.out; n_limit = r.n; nm_on = r.nm; };
        const uint32_t total = b->total_chunks, last_c = total - 1, G = gridDim.x;
        uint32_t cc = blockIdx.x;
        Cur cA = locate(cc < total ? cc : last_c);             // (more workgroups than chunks: they load the last chunk, store nothing)
        u4v vA = fetchb(cA);
        // (the tables are filled by the code below in the flat mode; here, behind the first load as well)
        {
            const int tt = threadIdx.x;
            for (int i = tt; i < 512; i += kThreads) s_win[i] = (a.exp_win && i < BFPQ_EXP_WIN_ENTRIES) ? a.exp_win[i] : 0;
            if constexpr (NM == 4) {
                for (int i = tt; i < BFPQ_NM4_LUT_ENTRIES; i += kThreads) {
                    const uint32_t k = a.nm_lut[i];
                    if constexpr (VEC == 8)
                        s_mask[i] = make_uint2(((k & 1u) ? 0xffffu : 0u) | ((k & 2u) ? 0xffff0000u : 0u),
                                               ((k & 4u) ? 0xffffu : 0u) | ((k & 8u) ? 0xffff0000u : 0u));
                    else s_keep[i] = (uint8_t)k;
                }
            }
        }
        __syncthreads();
        // two chunks per trip, register sets alternating by name (as the flat sweep below), bodies unguarded
        for (; cc < total && cc + G < total; cc += 2 * G) {
            const Cur cB = locate(cc + G);
            const u4v vB = fetchb(cB);
            use(cA);
            body(std::false_type{}, cA.item, u4(vA));
            const uint32_t c3 = cc + 2 * G < total ? cc + 2 * G : last_c;
            cA = locate(c3);
            vA = fetchb(cA);
            use(cB);
            body(std::false_type{}, cB.item, u4(vB));
        }
        if (cc < total) { use(cA); body(std::false_type{}, cA.item, u4(vA)); }
        return;
    }
    if constexpr (USE_BUF) {
        typedef unsigned int v4u __attribute__((vector_size(16)));
        const uint32_t n_bytes = (uint32_t)(a.n_items * 16);
        const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)n_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(a.out_deq, 0, (int)n_bytes, 0x00020000);
        const uint32_t sb = (uint32_t)stride * 16u;            // one sweep, in bytes
        uint32_t voff = ((uint32_t)blockIdx.x * kThreads + threadIdx.x) * 16u;
        auto bload = [&](uint32_t soff) __attribute__((always_inline)) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)voff, (int)soff, 2 /* nt */);
            return make_uint4(v[0], v[1], v[2], v[3]);
        };
        auto bstore = [&](const uint4 o, uint32_t soff) __attribute__((always_inline)) {
            const v4u v = {o.x, o.y, o.z, o.w};
            __builtin_amdgcn_raw_buffer_store_b128(v, r_out, (int)voff, (int)soff, 2 /* nt */);
        };
        uint4 c0 = bload(0);
        {
            // tables -> LDS behind the first tile's load, one dword per thread (see the flat sweep below)
            const int t = threadIdx.x;
            const bool win_al = a.exp_win && (reinterpret_cast<uintptr_t>(a.exp_win) & 3u) == 0;
            const bool lut_al = NM == 4 && (reinterpret_cast<uintptr_t>(a.nm_lut) & 3u) == 0;
            uint32_t w = 0, k4 = 0;
            if (win_al && t < BFPQ_EXP_WIN_ENTRIES / 4) w = reinterpret_cast<const uint32_t*>(a.exp_win)[t];
            if constexpr (NM == 4) {
                if (lut_al && t < BFPQ_NM4_LUT_ENTRIES / 4) k4 = reinterpret_cast<const uint32_t*>(a.nm_lut)[t];
                else if (lut_al && t == BFPQ_NM4_LUT_ENTRIES / 4) k4 = a.nm_lut[BFPQ_NM4_LUT_ENTRIES - 1];
            }
            if (win_al) { if (t < 128) reinterpret_cast<uint32_t*>(s_win)[t] = w; }
            else
                for (int i = t; i < 512; i += kThreads) s_win[i] = (a.exp_win && i < BFPQ_EXP_WIN_ENTRIES) ? a.exp_win[i] : 0;
            if constexpr (NM == 4) {
                auto put = [&](int i, uint32_t k) __attribute__((always_inline)) {
                    if constexpr (VEC == 8)
                        s_mask[i] = make_uint2(((k & 1u) ? 0xffffu : 0u) | ((k & 2u) ? 0xffff0000u : 0u),
                                               ((k & 4u) ? 0xffffu : 0u) | ((k & 8u) ? 0xffff0000u : 0u));
                    else s_keep[i] = (uint8_t)k;
                };
                if (lut_al) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (4 * t + j < BFPQ_NM4_LUT_ENTRIES) put(4 * t + j, (k4 >> (8 * j)) & 0xffu);
                } else
                    for (int i = t; i < BFPQ_NM4_LUT_ENTRIES; i += kThreads) put(i, a.nm_lut[i]);
            }
        }
        __syncthreads();
        // (one more memory operation behind the first load, result unused: see the flat sweep below -- with [load, dummy] on
        // the entry edge and [load, store] on the back edge the loop-top wait can leave the previous store in flight)
#if BFPQ_BUF_DUMMY
        asm volatile("" ::: "memory");
        const uint32_t dummy = *reinterpret_cast<const uint32_t*>(a.in);
        asm volatile("" ::: "memory");
#endif
        const int64_t pairs = ((a.n_items + stride - 1) / stride + 1) / 2;     // sweeps, two per trip (a sweep past the end is all out of range)
        for (int64_t p = 0; p < pairs; p++) {
            const uint4 c1 = bload(sb);
            body(std::false_type{}, 0, c0);
            bstore(buf_res, 0);
            c0 = bload(2 * sb);
            body(std::false_type{}, 0, c1);
            bstore(buf_res, sb);
            voff += 2 * sb;
        }
#if BFPQ_BUF_DUMMY
        asm volatile("" : : "v"(dummy));
#endif
        return;
    }
    // Sweep: item = sweep * stride + global thread id.  Loads run two sweeps ahead of the item being
    // processed (index clamped to the last item, never conditional).
    const int64_t last = a.n_items - 1;
    auto fetch = [&](int64_t i) __attribute__((always_inline)) {
        return __builtin_nontemporal_load(reinterpret_cast<const u4v*>(src + (i < last ? i : last)));
    };
    const int64_t full = a.n_items / stride;                               // sweeps in which every thread has an item
    int64_t item = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    u4v c0 = fetch(item);
    // Tables -> LDS, issued BEHIND the first tile's load and as one dword per thread.  The byte-per-thread loops this
    // replaces were 5 dependent global round trips (3 for the 729-byte N:M table, 2 for the window table) in front of the
    // first load: invisible on a 90 MB tensor (the CU's other workgroups cover it), a large part of the run time of the
    // launch-latency-sized tensors (OPT-125m, ViT).
    {
        const int t = threadIdx.x;
        auto put = [&](int i, uint32_t k) __attribute__((always_inline)) {
            if constexpr (NM == 4 && VEC == 8)
                s_mask[i] = make_uint2(((k & 1u) ? 0xffffu : 0u) | ((k & 2u) ? 0xffff0000u : 0u),
                                       ((k & 4u) ? 0xffffu : 0u) | ((k & 8u) ? 0xffff0000u : 0u));
            else if constexpr (NM == 4) s_keep[i] = (uint8_t)k;
        };
        const bool win_al = a.exp_win && (reinterpret_cast<uintptr_t>(a.exp_win) & 3u) == 0;
        const bool lut_al = NM == 4 && (reinterpret_cast<uintptr_t>(a.nm_lut) & 3u) == 0;
        uint32_t w = 0, k4 = 0;
        if (win_al && t < BFPQ_EXP_WIN_ENTRIES / 4) w = reinterpret_cast<const uint32_t*>(a.exp_win)[t];
        if constexpr (NM == 4) {
            if (lut_al && t < BFPQ_NM4_LUT_ENTRIES / 4) k4 = reinterpret_cast<const uint32_t*>(a.nm_lut)[t];
            else if (lut_al && t == BFPQ_NM4_LUT_ENTRIES / 4) k4 = a.nm_lut[BFPQ_NM4_LUT_ENTRIES - 1];    // 729 = 4 * 182 + 1
        }
        if (win_al) { if (t < 128) reinterpret_cast<uint32_t*>(s_win)[t] = w; }
        else
            for (int i = t; i < 512; i += kThreads) s_win[i] = (a.exp_win && i < BFPQ_EXP_WIN_ENTRIES) ? a.exp_win[i] : 0;
        if constexpr (NM == 4) {
            if (lut_al) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (4 * t + j < BFPQ_NM4_LUT_ENTRIES) put(4 * t + j, (k4 >> (8 * j)) & 0xffu);
            } else
                for (int i = t; i < BFPQ_NM4_LUT_ENTRIES; i += kThreads) put(i, a.nm_lut[i]);
        }
    }
    __syncthreads();
    if constexpr (NM == -1) {
        __shared__ uint32_t s_part[16];
        __shared__ uint32_t s_res[8];
        thr_setup<DT>(thr, a.selws, a.in, a.n_items * VEC, a.n_items, s_part, s_res);
    }
    // One more memory op behind the first load, result unused.  At the loop top the back edge arrives with [load, store]
    // outstanding and the entry edge with [load] only; one s_waitcnt immediate must serve both edges, so the compiler
    // emitted vmcnt(0) and every wave waited for its just-issued store once per iteration.  With [load, dummy] on the
    // entry edge both edges need vmcnt(1) and the store stays in flight across the loop top (A/B: 32.5 -> 31.95 us).
    // NB the scheduler still hoists the tile's first v_perm above the next prefetch, so a wave has ONE load in flight,
    // issued when the previous arrives; pinning the prefetch in front of that wait (two loads in flight) measured
    // SLOWER (33.25 us) -- like every other variant with more reads in flight per wave on this part.
    asm volatile("" ::: "memory");                                         // (pins the dummy between the first load and the loop)
    const uint32_t dummy = *reinterpret_cast<const uint32_t*>(src);
    asm volatile("" ::: "memory");
    int64_t sweep = 0;
    // main loop: load one sweep ahead; unrolled by two so that the two register sets alternate by NAME
    // (copying a register that a load in flight will write forces vmcnt(0)); with nothing conditional in
    // the body the waits are counted and the previous store stays in flight across the loop top
    for (; sweep + 2 <= full; sweep += 2, item += 2 * stride) {
        const u4v c1 = fetch(item + stride);
        body(std::false_type{}, item, u4(c0));
        c0 = fetch(item + 2 * stride);
        body(std::false_type{}, item + stride, u4(c1));
    }
    // remaining full sweep (0..1) and the ragged last one: guarded, rolled (block-uniform trip count)
    for (; item < n_round; item += stride) {
        const u4v c1 = fetch(item + stride);
        body(std::true_type{}, item, u4(c0));
        c0 = c1;
    }
    asm volatile("" : : "v"(dummy));                                       // the dummy's only "use": after all the work
}

template <int DT, int NM, bool SFIRST, bool STOCH, int LPBT, bool DEQ_ONLY>
__global__ void __launch_bounds__(kThreads) k_fused_flat(const FusedArgs a)
{
    fused_flat_body<DT, NM, SFIRST, STOCH, LPBT, DEQ_ONLY, false>(a, nullptr);
}

// the same item pipeline over a list of tensors (drop-in mode, round-half-even, dense or N:4)
template <int DT, int NM, bool SFIRST, int LPBT>
__global__ void __launch_bounds__(kThreads) k_fused_batched(const FusedArgs a, const BatchArgs b)
{
    // The descriptor list is indexed with a run-time (wave-uniform) index.  Taking the address of the by-value parameter
    // would make hipcc copy all 2 KB of it into per-lane scratch; reading it where it already lies -- in the kernel
    // argument segment, explicit arguments in order at their natural alignment -- keeps the accesses scalar loads.
    constexpr size_t kOff = (sizeof(FusedArgs) + alignof(BatchArgs) - 1) / alignof(BatchArgs) * alignof(BatchArgs);
#if defined(__HIP_DEVICE_COMPILE__)
    const BatchArgs* bp = (const BatchArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kOff);
#else
    const BatchArgs* bp = &b;
    (void)kOff;
#endif
    (void)b;
    fused_flat_body<DT, NM, SFIRST, false, LPBT, true, true>(a, bp);
}

// ---------------------------------------------------------------------------------------------
// k_nm_rows: one thread per N:M group, general M, ragged rows (tail group padded with zeros as
// F.pad does, bfp_ops.py:79-82).  codes (optional) are zeroed where an element is pruned.
// ---------------------------------------------------------------------------------------------
// Prune mask of one group whose (key << 8 | index) pairs sit in `view`.  Which elements the reference's topk drops is
// only a question of ITS tie order when equal magnitudes straddle the cut; otherwise "the M-N smallest" is unambiguous.
// So: count, for every element, the keys below it and the keys equal to it (M^2 uniform LDS reads, no branches); an
// element with less + equal <= P is certainly pruned, one with less >= P certainly kept.  Only a group that has an
// element in between replays libstdc++'s nth_element (nm_prune_mask) -- rare on real-valued weights, common on inputs
// that are already quantized.
__device__ __forceinline__ uint64_t nm_prune_mask_counted(KvView& view, int N, int M)
{
    const int P = M - N;
    uint64_t prune = 0;
    bool ambiguous = false;
    for (int i = 0; i < M; i++) {
        const uint64_t ki = view.get(i) >> 8;
        int less = 0, eq = 0;
        for (int j = 0; j < M; j++) {
            const uint64_t kj = view.get(j) >> 8;
            less += kj < ki;
            eq += kj == ki;
        }
        if (less + eq <= P) prune |= 1ull << i;
        else if (less < P) ambiguous = true;
    }
    return ambiguous ? nm_prune_mask(view, N, M) : prune;
}

template <int DT>
__global__ void __launch_bounds__(128) k_nm_rows(const void* in, void* out, void* codes, int code_bits,
                                                 int64_t rows, int64_t cols, int N, int M, const uint8_t* lut8)
{
    using raw_t = typename Traits<DT>::raw_t;
    extern __shared__ uint64_t s_kv[];                    // [M][blockDim.x]
    const int64_t ngrp = (cols + M - 1) / M;
    const int64_t total = rows * ngrp;
    const raw_t* src = reinterpret_cast<const raw_t*>(in);
    raw_t* dst = reinterpret_cast<raw_t*>(out);
    KvView view{s_kv + threadIdx.x, (int)blockDim.x};
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = g / ngrp, gi = g - row * ngrp;
        const int64_t c0 = gi * M;
        const int n = (int)((cols - c0) < M ? (cols - c0) : M);
        const int64_t base = row * cols + c0;
        constexpr int EPV = 16 / (int)sizeof(raw_t);                       // elements per 16-byte vector
        // whole groups that are 16-byte multiples at 16-byte aligned addresses move as vectors
        const bool vec_io = !codes && (M % EPV) == 0 && (cols % M) == 0 &&
                            ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
        if (vec_io && M == 8) {
            // the common general case (N:8) entirely in registers: 28 pair comparisons give every element its count of
            // smaller and of equal keys; LDS and the nth_element replay only for a group whose ties straddle the cut
            constexpr int Q = 8 / EPV;                                      // 16-byte vectors per group (1 or 2)
            uint32_t d[Q][4], key[8];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const uint4 v = reinterpret_cast<const uint4*>(src + base)[q];
                d[q][0] = v.x; d[q][1] = v.y; d[q][2] = v.z; d[q][3] = v.w;
#pragma unroll
                for (int e = 0; e < EPV; e++)
                    key[q * EPV + e] = mag_key<DT>(sizeof(raw_t) == 4 ? d[q][e] : ((d[q][e >> 1] >> (16 * (e & 1))) & 0xffffu));
            }
            int less[8] = {0, 0, 0, 0, 0, 0, 0, 0}, eq[8] = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = i + 1; j < 8; j++) {
                    const int lt = key[i] < key[j], e = key[i] == key[j];
                    less[j] += lt; less[i] += 1 - lt - e; eq[i] += e; eq[j] += e;
                }
            const int P = 8 - N;
            uint32_t prune = 0;
            bool ambiguous = false;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (less[i] + eq[i] <= P) prune |= 1u << i;
                else if (less[i] < P) ambiguous = true;
            }
            if (ambiguous && lut8) {                                        // the weak ordering indexes the host-built table
                uint32_t idx = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) idx |= (uint32_t)less[i] << (3 * i);
                prune = lut8[idx];
            } else if (ambiguous) {
#pragma unroll
                for (int i = 0; i < 8; i++) view.set(i, ((uint64_t)key[i] << 8) | (uint64_t)i);
                prune = (uint32_t)nm_prune_mask(view, N, M);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
#pragma unroll
                for (int e = 0; e < EPV; e++) {
                    if ((prune >> (q * EPV + e)) & 1u) {
                        if (sizeof(raw_t) == 4) d[q][e] = 0u;
                        else d[q][e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
                    }
                }
                reinterpret_cast<uint4*>(dst + base)[q] = make_uint4(d[q][0], d[q][1], d[q][2], d[q][3]);
            }
            continue;
        }
        if (vec_io) {
            for (int q = 0; q < M / EPV; q++) {
                const uint4 v = reinterpret_cast<const uint4*>(src + base)[q];
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < EPV; e++) {
                    const uint32_t r = sizeof(raw_t) == 4 ? d[e] : ((d[e >> 1] >> (16 * (e & 1))) & 0xffffu);
                    view.set(q * EPV + e, ((uint64_t)mag_key<DT>(r) << 8) | (uint64_t)(q * EPV + e));
                }
            }
            const uint64_t prune = nm_prune_mask_counted(view, N, M);
            for (int q = 0; q < M / EPV; q++) {
                uint4 v = reinterpret_cast<const uint4*>(src + base)[q];
                uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < EPV; e++) {
                    if ((prune >> (q * EPV + e)) & 1ull) {
                        if (sizeof(raw_t) == 4) d[e] = 0u;
                        else d[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
                    }
                }
                reinterpret_cast<uint4*>(dst + base)[q] = make_uint4(d[0], d[1], d[2], d[3]);
            }
            continue;
        }
        for (int i = 0; i < M; i++) {
            const uint32_t key = i < n ? mag_key<DT>((uint32_t)src[base + i]) : 0u;
            view.set(i, ((uint64_t)key << 8) | (uint64_t)i);
        }
        const uint64_t prune = nm_prune_mask_counted(view, N, M);
        for (int i = 0; i < n; i++) {
            const bool p = (prune >> i) & 1ull;
            dst[base + i] = p ? (raw_t)0 : src[base + i];
            if (codes && p) {
                const int64_t e = base + i;
                if (code_bits == 4) {
                    uint8_t* cb = reinterpret_cast<uint8_t*>(codes) + row * ((cols + 1) / 2) + (c0 + i) / 2;
                    *cb = ((c0 + i) & 1) ? (uint8_t)(*cb & 0x0f) : (uint8_t)(*cb & 0xf0);
                } else if (code_bits == 8) reinterpret_cast<int8_t*>(codes)[e] = 0;
                else reinterpret_cast<int16_t*>(codes)[e] = 0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_quant_rows: general HBFP quantizer.  A group of G lanes (power of two, <= 64) owns one block
// of `block` elements of one row (the tail block of a row is shorter: the zero pad of
// bfp_ops.py:50-53 does not change the max and is cut off again).
// ---------------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(kThreads) k_quant_rows(const void* in, void* out_deq, void* out_codes, int8_t* out_exp,
                                                         int64_t rows, int64_t cols, int block, int G,
                                                         int mant_bits, float eps_dt, int code_bits, uint64_t seed,
                                                         const uint8_t* exp_win)
{
    using T = Traits<DT>;
    using raw_t = typename T::raw_t;
    __shared__ uint8_t s_win[BFPQ_EXP_WIN_ENTRIES];
    for (int i = threadIdx.x; i < BFPQ_EXP_WIN_ENTRIES; i += kThreads) s_win[i] = exp_win[i];
    __syncthreads();
    const bool stoch = seed != 0;
    const int64_t nblk = (cols + block - 1) / block;
    const int64_t total = rows * nblk;
    const int64_t groups_per_grid = (int64_t)gridDim.x * kThreads / G;
    const int lig = threadIdx.x % G;
    const int64_t total_round = (total + groups_per_grid - 1) / groups_per_grid * groups_per_grid;
    const raw_t* src = reinterpret_cast<const raw_t*>(in);
    for (int64_t b = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / G; b < total_round; b += groups_per_grid) {
        const bool valid = b < total;
        const int64_t row = valid ? b / nblk : 0, bi = valid ? b - row * nblk : 0;
        const int64_t c0 = bi * block;
        const int len = valid ? (int)((cols - c0) < block ? (cols - c0) : block) : 0;
        const int64_t base = row * cols + c0;
        uint32_t mx = 0;
        for (int i = lig; i < len; i += G) { const uint32_t k = (uint32_t)src[base + i] & T::ABS; mx = k > mx ? k : mx; }
        for (int o = 1; o < G; o <<= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)mx, o, 64); mx = other > mx ? other : mx; }
        const BlockScale bs = block_scale<DT>(mx, mant_bits, eps_dt, s_win);
        // two elements per step so that a 4-bit code byte is written by one lane (block is even then)
        for (int i = 2 * lig; i < len; i += 2 * G) {
            float c2[2] = {0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (i + h >= len) break;
                const float dither = stoch ? uniform24(seed, (uint64_t)(base + i + h)) - 0.5f : 0.f;
                const float yv = quant_elem<DT>(raw_to_f32<DT>((uint32_t)src[base + i + h]), bs, stoch, dither, &c2[h]);
                if (out_deq) reinterpret_cast<raw_t*>(out_deq)[base + i + h] = (raw_t)f32_to_raw<DT>(yv);
                if (out_codes && code_bits == 8) reinterpret_cast<int8_t*>(out_codes)[base + i + h] = (int8_t)(int)c2[h];
                if (out_codes && code_bits == 16) reinterpret_cast<int16_t*>(out_codes)[base + i + h] = (int16_t)(int)c2[h];
            }
            if (out_codes && code_bits == 4) {
                uint8_t* cb = reinterpret_cast<uint8_t*>(out_codes) + row * ((cols + 1) / 2) + (c0 + i) / 2;
                *cb = (uint8_t)(((uint32_t)(int)c2[0] & 0xfu) | (((uint32_t)(int)c2[1] & 0xfu) << 4));
            }
        }
        if (out_exp && valid && lig == 0) out_exp[b] = sat_exp(bs);
    }
}

// ---------------------------------------------------------------------------------------------
// k_quant_rows_tiled: drop-in HBFP quantizer for rows whose length is a multiple of the vector width but NOT of the
// block (the reference's per-row F.pad, bfp_ops.py:50-53), block = a power-of-two number of lane items.  A workgroup
// walks whole rows (no division anywhere), its waves the 64-item tiles of the row; lanes past the end of the row hold the
// pad zeros.  Same three tiers of arithmetic as the flat kernel (lean 16-bit path / exact power-of-two path / step replay).
// ---------------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(kThreads) k_quant_rows_tiled(const FusedArgs a, int64_t rows, int64_t ipr)
{
    using T = Traits<DT>;
    constexpr int VEC = T::VEC;
    __shared__ __attribute__((aligned(16))) uint8_t s_win[512];
    for (int i = threadIdx.x; i < 512; i += kThreads) s_win[i] = i < BFPQ_EXP_WIN_ENTRIES ? a.exp_win[i] : 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t tpr = (ipr + 63) / 64;                          // tiles per row
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(a.in);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(a.out_deq);
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    // (row, tile) of this wave, advanced tile by tile; the next tile's load is issued before the current one is processed
    int64_t row = blockIdx.x, tile = w;
    auto advance = [&](int64_t& r, int64_t& t) __attribute__((always_inline)) {
        t += kThreads / 64;
        if (t >= tpr) { t = w; r += gridDim.x; }
    };
    auto fetch = [&](int64_t r, int64_t t) __attribute__((always_inline)) {     // lanes past the end of the row hold the pad zeros
        const int64_t it = t * 64 + lane;
        u4v v = {0u, 0u, 0u, 0u};
        if (r < rows && it < ipr) v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(src + r * ipr + it));
        return v;
    };
    if (tile >= tpr) { tile = w; row = rows; }                    // (more waves than tiles in a row: idle wave)
    u4v cur = fetch(row, tile);
    // (measured: an unconditional clamped load + two tiles per trip, as in the flat kernel, is SLOWER here: 41.4 vs 38.6 us)
    while (row < rows) {
        int64_t nrow = row, ntile = tile;
        advance(nrow, ntile);
        const u4v nxt = fetch(nrow, ntile);
        const u4v c = cur;
        const int64_t it = tile * 64 + lane;
        const bool valid = it < ipr;
        uint32_t d0 = c.x, d1 = c.y, d2 = c.z, d3 = c.w, o0, o1, o2, o3;
        uint32_t mx;
        if constexpr (VEC == 8) {
            const uint32_t absm = T::ABS | (T::ABS << 16);
            const uint32_t mp = pk_max_u16(pk_max_u16(d0 & absm, d1 & absm), pk_max_u16(d2 & absm, d3 & absm));
            mx = (mp & 0xffffu) > (mp >> 16) ? (mp & 0xffffu) : (mp >> 16);
        } else {
            const uint32_t m01 = (d0 & T::ABS) > (d1 & T::ABS) ? (d0 & T::ABS) : (d1 & T::ABS);
            const uint32_t m23 = (d2 & T::ABS) > (d3 & T::ABS) ? (d2 & T::ABS) : (d3 & T::ABS);
            mx = m01 > m23 ? m01 : m23;
        }
        mx = group_max<-1>(mx, a.lpb);
        bool hot = false;
        [[maybe_unused]] Hot16 h16;
        if constexpr (VEC == 8) { h16 = hot16_scale<DT>(mx, a, s_win); hot = !__any(!h16.ok); }
        if (hot) {
            if constexpr (VEC == 8) {
                const uint32_t absm = T::ABS | (T::ABS << 16);
                const uint32_t dd[4] = {d0, d1, d2, d3};
                uint32_t oo[4];
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const uint32_t am = pk_min_u16(dd[x] & absm, h16.maxv2);
                    typedef float float2v __attribute__((ext_vector_type(2)));
                    const float2v C2 = {h16.C, h16.C};
                    float2v v;
                    if constexpr (DT == BFPQ_BF16) v = (float2v){u2f(am << 16), u2f(am & 0xffff0000u)} + C2;
                    else v = (float2v){fma_mix_f16<false>(am, h16.C), fma_mix_f16<true>(am, h16.C)};
                    v -= C2;
                    uint32_t pk;
                    if constexpr (DT == BFPQ_BF16) pk = __builtin_amdgcn_perm(f2u(v.y), f2u(v.x), 0x07060302u);
                    else pk = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(v.x, v.y));
                    oo[x] = bfi_b32(absm, pk, dd[x]);
                }
                o0 = oo[0]; o1 = oo[1]; o2 = oo[2]; o3 = oo[3];
            }
        } else {
            uint32_t raw[VEC], outraw[VEC];
            if constexpr (VEC == 4) { raw[0] = d0; raw[1] = d1; raw[2] = d2; raw[3] = d3; }
            else {
                raw[0] = d0 & 0xffffu; raw[1] = d0 >> 16; raw[2] = d1 & 0xffffu; raw[3] = d1 >> 16;
                raw[4] = d2 & 0xffffu; raw[5] = d2 >> 16; raw[6] = d3 & 0xffffu; raw[7] = d3 >> 16;
            }
            const FastScale fs = fast_scale<DT>(mx, a.mant_bits, a.eps_dt, s_win);
            if (__any(a.force_slow != 0 || !fs.ok)) {
                const BlockScale bs = block_scale<DT>(mx, a.mant_bits, a.eps_dt, s_win);
                float code;
#pragma unroll
                for (int j = 0; j < VEC; j++) outraw[j] = f32_to_raw<DT>(quant_elem<DT>(raw_to_f32<DT>(raw[j]), bs, false, 0.f, &code));
            } else {
#pragma unroll
                for (int j = 0; j < VEC; j++)
                    outraw[j] = f32_to_raw<DT>(__builtin_amdgcn_fmed3f(rintf(raw_to_f32<DT>(raw[j]) * fs.inv), -fs.qmax, fs.qmax) * fs.interval);
            }
            if constexpr (VEC == 4) { o0 = outraw[0]; o1 = outraw[1]; o2 = outraw[2]; o3 = outraw[3]; }
            else {
                o0 = outraw[0] | (outraw[1] << 16); o1 = outraw[2] | (outraw[3] << 16);
                o2 = outraw[4] | (outraw[5] << 16); o3 = outraw[6] | (outraw[7] << 16);
            }
        }
        if (valid) stream_store(dst + row * ipr + it, make_uint4(o0, o1, o2, o3));
        row = nrow; tile = ntile; cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------
// k_quant_rows_vec: HBFP quantizer for rows that are 16-byte aligned (cols % VEC == 0, block % VEC == 0,
// aligned pointers) but whose length is NOT a multiple of the block, or whose block is not a power-of-two
// number of lane items.  A block owns GP = pow2ceil(block / VEC) adjacent lanes; lane j of the group holds
// lane item j of the block when that item exists in the row (else zeros: the reference's F.pad).  One
// pass, registers only.  Virtual item v = (row * nblk + blk) * GP + j.
// ---------------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(kThreads) k_quant_rows_vec(const void* in, void* out_deq, void* out_codes, int8_t* out_exp,
                                                             int64_t rows, int64_t cols, int block, int GP,
                                                             int mant_bits, float eps_dt, int code_bits, uint64_t seed,
                                                             const uint8_t* exp_win)
{
    using T = Traits<DT>;
    constexpr int VEC = T::VEC;
    __shared__ uint8_t s_win[BFPQ_EXP_WIN_ENTRIES];
    for (int i = threadIdx.x; i < BFPQ_EXP_WIN_ENTRIES; i += kThreads) s_win[i] = exp_win[i];
    __syncthreads();
    const bool stoch = seed != 0;
    const int64_t ipr = cols / VEC;                         // lane items per row
    const int ipb = block / VEC;                            // lane items per (full) block
    const int64_t nblk = (cols + block - 1) / block;
    const int64_t total = rows * nblk * GP;                 // virtual items
    const int64_t total_round = (total + 63) / 64 * 64;
    const int j = threadIdx.x % GP;
    for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < total_round; v += (int64_t)gridDim.x * kThreads) {
        const int64_t b = v / GP;                           // global block index (GP is a power of two: a shift)
        const int64_t row = b / nblk, blk = b - row * nblk;
        const int64_t it = blk * ipb + j;                   // lane item inside the row
        const bool valid = v < total && j < ipb && it < ipr;
        const int64_t item = row * ipr + it;
        uint32_t raw[VEC];
        {
            uint4 q = make_uint4(0, 0, 0, 0);
            if (valid) q = reinterpret_cast<const uint4*>(in)[item];
            if constexpr (VEC == 4) { raw[0] = q.x; raw[1] = q.y; raw[2] = q.z; raw[3] = q.w; }
            else {
                raw[0] = q.x & 0xffffu; raw[1] = q.x >> 16; raw[2] = q.y & 0xffffu; raw[3] = q.y >> 16;
                raw[4] = q.z & 0xffffu; raw[5] = q.z >> 16; raw[6] = q.w & 0xffffu; raw[7] = q.w >> 16;
            }
        }
        uint32_t mx = 0;
#pragma unroll
        for (int e = 0; e < VEC; e++) { const uint32_t k = raw[e] & T::ABS; mx = k > mx ? k : mx; }
        for (int o = 1; o < GP; o <<= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)mx, o, 64); mx = other > mx ? other : mx; }
        // same split as the flat kernel: exact power-of-two arithmetic unless some block of the wave is unusual
        // (zero / non-finite max, scale outside the normal range, mantissa wider than the dtype) or rounding is stochastic
        const FastScale fs = fast_scale<DT>(mx, mant_bits, eps_dt, s_win);
        float y[VEC], code[VEC];
        int8_t e_sat;
        if (__builtin_expect(__any(stoch || !fs.ok), 0)) {
            const BlockScale bs = block_scale<DT>(mx, mant_bits, eps_dt, s_win);
            e_sat = sat_exp(bs);
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                const float dither = stoch ? uniform24(seed, (uint64_t)item * VEC + e) - 0.5f : 0.f;
                y[e] = quant_elem<DT>(raw_to_f32<DT>(raw[e]), bs, stoch, dither, &code[e]);
            }
        } else {
            e_sat = (int8_t)(fs.e < -127 ? -127 : (fs.e > 127 ? 127 : fs.e));
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                code[e] = __builtin_amdgcn_fmed3f(rintf(raw_to_f32<DT>(raw[e]) * fs.inv), -fs.qmax, fs.qmax);
                y[e] = code[e] * fs.interval;
            }
        }
        if (!valid) continue;
        if (out_deq) {
            uint32_t o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e++) o[e] = f32_to_raw<DT>(y[e]);
            uint4 w;
            if constexpr (VEC == 4) w = make_uint4(o[0], o[1], o[2], o[3]);
            else w = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
            reinterpret_cast<uint4*>(out_deq)[item] = w;
        }
        if (out_codes) {
            int c[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e++) c[e] = (int)code[e];
            if (code_bits == 4) {
                uint32_t w = 0;
#pragma unroll
                for (int e = 0; e < VEC; e++) w |= ((uint32_t)c[e] & 0xfu) << (4 * e);
                if constexpr (VEC == 8) reinterpret_cast<uint32_t*>(out_codes)[item] = w;
                else reinterpret_cast<uint16_t*>(out_codes)[item] = (uint16_t)w;
            } else if (code_bits == 8) {
#pragma unroll
                for (int e = 0; e < VEC; e++) reinterpret_cast<int8_t*>(out_codes)[item * VEC + e] = (int8_t)c[e];
            } else {
#pragma unroll
                for (int e = 0; e < VEC; e++) reinterpret_cast<int16_t*>(out_codes)[item * VEC + e] = (int16_t)c[e];
            }
        }
        if (out_exp && j == 0) out_exp[b] = e_sat;
    }
}

template <int DT, int NM, bool SFIRST, bool STOCH, bool DEQ_ONLY>
int launch_fused_o(const FusedArgs& a0, hipStream_t s)
{
    FusedArgs a = a0;
    if constexpr (DEQ_ONLY && !STOCH && NM != -1) {
        // these instantiations address the tensor through 32-bit buffer offsets (items + three sweeps of look-ahead must stay
        // below 4 GB): a larger tensor goes in pieces of 2^27 items (2 GB), whole blocks and whole chunks of 256 items each
        const int64_t piece = (int64_t)1 << 27;
        if (a0.n_items > piece + (piece >> 1)) {
            for (int64_t i0 = 0; i0 < a0.n_items; i0 += piece) {
                FusedArgs b = a0;
                b.in = reinterpret_cast<const char*>(a0.in) + i0 * 16;
                b.out_deq = reinterpret_cast<char*>(a0.out_deq) + i0 * 16;
                b.n_items = a0.n_items - i0 < piece ? a0.n_items - i0 : piece;
                const int rc = launch_fused_o<DT, NM, SFIRST, STOCH, DEQ_ONLY>(b, s);
                if (rc) return rc;
            }
            return 0;
        }
    }
    const dim3 grid(grid_for(a.n_items)), block(kThreads);
    if constexpr (!STOCH && NM != 2) {           // the shapes that matter get a compile-time lane group
        switch (a.lpb) {
            case 2: hipLaunchKernelGGL((k_fused_flat<DT, NM, SFIRST, STOCH, 2, DEQ_ONLY>), grid, block, 0, s, a); return (int)hipGetLastError();
            case 4: hipLaunchKernelGGL((k_fused_flat<DT, NM, SFIRST, STOCH, 4, DEQ_ONLY>), grid, block, 0, s, a); return (int)hipGetLastError();
            case 8: hipLaunchKernelGGL((k_fused_flat<DT, NM, SFIRST, STOCH, 8, DEQ_ONLY>), grid, block, 0, s, a); return (int)hipGetLastError();
            case 16: hipLaunchKernelGGL((k_fused_flat<DT, NM, SFIRST, STOCH, 16, DEQ_ONLY>), grid, block, 0, s, a); return (int)hipGetLastError();
            default: break;
        }
    }
    hipLaunchKernelGGL((k_fused_flat<DT, NM, SFIRST, STOCH, -1, DEQ_ONLY>), grid, block, 0, s, a);
    return (int)hipGetLastError();
}

template <int DT, int NM, bool SFIRST, bool STOCH>
int launch_fused_l(const FusedArgs& a, hipStream_t s)
{
    const bool deq_only = a.out_deq && !a.out_codes && !a.out_exp;
    if (deq_only) return launch_fused_o<DT, NM, SFIRST, STOCH, true>(a, s);
    return launch_fused_o<DT, NM, SFIRST, STOCH, false>(a, s);
}

template <int DT>
int launch_fused_threshold(const FusedArgs& a, hipStream_t s)
{
    const dim3 grid(grid_for(a.n_items)), block(kThreads);
    const bool deq_only = a.out_deq && !a.out_codes && !a.out_exp;
    if (a.seed) {
        hipLaunchKernelGGL((k_fused_flat<DT, -1, true, true, -1, false>), grid, block, 0, s, a);
    } else if (deq_only) {
        if (a.lpb == 8) hipLaunchKernelGGL((k_fused_flat<DT, -1, true, false, 8, true>), grid, block, 0, s, a);
        else if (a.lpb == 4) hipLaunchKernelGGL((k_fused_flat<DT, -1, true, false, 4, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_fused_flat<DT, -1, true, false, -1, true>), grid, block, 0, s, a);
    } else hipLaunchKernelGGL((k_fused_flat<DT, -1, true, false, -1, false>), grid, block, 0, s, a);
    return (int)hipGetLastError();
}

// N:8 (16-bit dtypes): few instantiations -- lane groups 8 (block 64) or run-time
template <int DT, bool SFIRST, bool STOCH>
int launch_fused_nm8(const FusedArgs& a, hipStream_t s)
{
    const bool deq_only0 = a.out_deq && !a.out_codes && !a.out_exp;
    const int64_t piece = (int64_t)1 << 27;                   // (see launch_fused_o: 32-bit buffer offsets in the drop-in instantiations)
    if (!STOCH && deq_only0 && a.n_items > piece + (piece >> 1)) {
        for (int64_t i0 = 0; i0 < a.n_items; i0 += piece) {
            FusedArgs b = a;
            b.in = reinterpret_cast<const char*>(a.in) + i0 * 16;
            b.out_deq = reinterpret_cast<char*>(a.out_deq) + i0 * 16;
            b.n_items = a.n_items - i0 < piece ? a.n_items - i0 : piece;
            const int rc = launch_fused_nm8<DT, SFIRST, STOCH>(b, s);
            if (rc) return rc;
        }
        return 0;
    }
    const dim3 grid(grid_for(a.n_items)), block(kThreads);
    const bool deq_only = a.out_deq && !a.out_codes && !a.out_exp;
    if constexpr (STOCH) hipLaunchKernelGGL((k_fused_flat<DT, 8, SFIRST, true, -1, false>), grid, block, 0, s, a);
    else if constexpr (Traits<DT>::VEC == 8) {
        if (deq_only && a.lpb == 8) hipLaunchKernelGGL((k_fused_flat<DT, 8, SFIRST, false, 8, true>), grid, block, 0, s, a);
        else if (deq_only) hipLaunchKernelGGL((k_fused_flat<DT, 8, SFIRST, false, -1, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_fused_flat<DT, 8, SFIRST, false, -1, false>), grid, block, 0, s, a);
    } else {
        if (deq_only) hipLaunchKernelGGL((k_fused_flat<DT, 8, SFIRST, false, -1, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_fused_flat<DT, 8, SFIRST, false, -1, false>), grid, block, 0, s, a);
    }
    return (int)hipGetLastError();
}

template <int DT, bool STOCH>
int launch_fused_s(const FusedArgs& a, int M, bool sfirst, hipStream_t s)
{
    if (M == 8) return sfirst ? launch_fused_nm8<DT, true, STOCH>(a, s) : launch_fused_nm8<DT, false, STOCH>(a, s);
    if (M == 0) return launch_fused_l<DT, 0, true, STOCH>(a, s);
    if (M == 2) return sfirst ? launch_fused_l<DT, 2, true, STOCH>(a, s) : launch_fused_l<DT, 2, false, STOCH>(a, s);
    return sfirst ? launch_fused_l<DT, 4, true, STOCH>(a, s) : launch_fused_l<DT, 4, false, STOCH>(a, s);
}

template <int DT>
int launch_fused(const FusedArgs& a, int M, bool sfirst, hipStream_t s)
{
    return a.seed ? launch_fused_s<DT, true>(a, M, sfirst, s) : launch_fused_s<DT, false>(a, M, sfirst, s);
}

int launch_nm_rows(const void* in, void* out, void* codes, int code_bits, int64_t rows, int64_t cols, int dtype, int N, int M, const uint8_t* lut8, hipStream_t s)
{
    const int threads = 128;
    const int64_t total = rows * ((cols + M - 1) / M);
    if (total == 0) return 0;
    int64_t g = (total + threads - 1) / threads;
    const int grid = (int)(g > 4096 ? 4096 : g);
    const size_t lds = (size_t)M * threads * sizeof(uint64_t);
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)k_nm_rows<BFPQ_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_nm_rows<BFPQ_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_nm_rows<BFPQ_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (dtype == BFPQ_F32) hipLaunchKernelGGL((k_nm_rows<BFPQ_F32>), dim3(grid), dim3(threads), lds, s, in, out, codes, code_bits, rows, cols, N, M, lut8);
    else if (dtype == BFPQ_F16) hipLaunchKernelGGL((k_nm_rows<BFPQ_F16>), dim3(grid), dim3(threads), lds, s, in, out, codes, code_bits, rows, cols, N, M, lut8);
    else hipLaunchKernelGGL((k_nm_rows<BFPQ_BF16>), dim3(grid), dim3(threads), lds, s, in, out, codes, code_bits, rows, cols, N, M, lut8);
    return (int)hipGetLastError();
}

void set_hot16(FusedArgs& a, int dtype, int mant_bits, float eps_dt);

int launch_quant_rows(const void* in, void* out_deq, void* out_codes, int8_t* out_exp, int64_t rows, int64_t cols, int dtype,
                      int block, int mant_bits, float eps_dt, int code_bits, uint64_t seed, const uint8_t* exp_win, hipStream_t s)
{
    const int64_t total = rows * ((cols + block - 1) / block);
    if (total == 0) return 0;
    const int vec = dtype_vec(dtype);
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out_deq) | reinterpret_cast<uintptr_t>(out_codes)) & 15u) == 0;
    if (aligned && out_deq && !out_codes && !out_exp && seed == 0 && cols % vec == 0 && block % vec == 0 && is_pow2(block / vec) && block / vec <= 64) {
        FusedArgs a;
        a.in = in; a.out_deq = out_deq; a.out_codes = nullptr; a.out_exp = nullptr; a.n_items = rows * cols / vec;
        a.exp_win = exp_win; a.nm_lut = nullptr; a.seed = 0; a.eps_dt = eps_dt; a.lpb = block / vec;
        a.mant_bits = mant_bits; a.N = 0; a.code_bits = 0;
        a.force_slow = mant_bits > (dtype == BFPQ_F32 ? 24 : (dtype == BFPQ_F16 ? 11 : 8));
        set_hot16(a, dtype, mant_bits, eps_dt);
        a.selws = nullptr;
        const int64_t ipr = cols / vec;
        const int grid = (int)(rows < 2048 ? rows : 2048);
        if (dtype == BFPQ_F32) hipLaunchKernelGGL((k_quant_rows_tiled<BFPQ_F32>), dim3(grid), dim3(kThreads), 0, s, a, rows, ipr);
        else if (dtype == BFPQ_F16) hipLaunchKernelGGL((k_quant_rows_tiled<BFPQ_F16>), dim3(grid), dim3(kThreads), 0, s, a, rows, ipr);
        else hipLaunchKernelGGL((k_quant_rows_tiled<BFPQ_BF16>), dim3(grid), dim3(kThreads), 0, s, a, rows, ipr);
        return (int)hipGetLastError();
    }
    if (aligned && cols % vec == 0 && block % vec == 0 && block / vec <= 64 && (code_bits != 4 || cols % 2 == 0)) {
        int GP = 1;
        while (GP < block / vec) GP <<= 1;
        const int grid = grid_for(total * GP);
        if (dtype == BFPQ_F32) hipLaunchKernelGGL((k_quant_rows_vec<BFPQ_F32>), dim3(grid), dim3(kThreads), 0, s, in, out_deq, out_codes, out_exp, rows, cols, block, GP, mant_bits, eps_dt, code_bits, seed, exp_win);
        else if (dtype == BFPQ_F16) hipLaunchKernelGGL((k_quant_rows_vec<BFPQ_F16>), dim3(grid), dim3(kThreads), 0, s, in, out_deq, out_codes, out_exp, rows, cols, block, GP, mant_bits, eps_dt, code_bits, seed, exp_win);
        else hipLaunchKernelGGL((k_quant_rows_vec<BFPQ_BF16>), dim3(grid), dim3(kThreads), 0, s, in, out_deq, out_codes, out_exp, rows, cols, block, GP, mant_bits, eps_dt, code_bits, seed, exp_win);
        return (int)hipGetLastError();
    }
    int G = 1;
    while (G < 64 && 2 * G < block) G <<= 1;       // two elements per lane per step
    const int grid = grid_for(total * G);
    if (dtype == BFPQ_F32) hipLaunchKernelGGL((k_quant_rows<BFPQ_F32>), dim3(grid), dim3(kThreads), 0, s, in, out_deq, out_codes, out_exp, rows, cols, block, G, mant_bits, eps_dt, code_bits, seed, exp_win);
    else if (dtype == BFPQ_F16) hipLaunchKernelGGL((k_quant_rows<BFPQ_F16>), dim3(grid), dim3(kThreads), 0, s, in, out_deq, out_codes, out_exp, rows, cols, block, G, mant_bits, eps_dt, code_bits, seed, exp_win);
    else hipLaunchKernelGGL((k_quant_rows<BFPQ_BF16>), dim3(grid), dim3(kThreads), 0, s, in, out_deq, out_codes, out_exp, rows, cols, block, G, mant_bits, eps_dt, code_bits, seed, exp_win);
    return (int)hipGetLastError();
}

// range of block maxima (dtype exponent field) that the hot16 path of k_fused_flat takes, and its max_v constant; see hot16_scale
void set_hot16(FusedArgs& a, int dtype, int mant_bits, float eps_dt)
{
    a.kb_lo = 1000; a.kb_span = 0; a.maxv_c = 0;            // never
    if (dtype == BFPQ_F32 || a.force_slow || mant_bits < 1 || mant_bits > (dtype == BFPQ_F16 ? 11 : 8)) return;
    const int mb = dtype == BFPQ_F16 ? 10 : 7, eoff = dtype == BFPQ_F16 ? 112 : 0;
    int lo = 1 + eoff, hi = (dtype == BFPQ_F16 ? 30 : 254) + eoff;     // fp32-biased exponent field of the block max: normal, finite
    while (lo <= hi && !(eps_dt < ldexpf(1.0f, lo - 127 - mb - 1))) lo++;   // max + epsilon rounds back to max
    if (lo < mant_bits + 3) lo = mant_bits + 3;              // interval >= 2^-124
    if (lo < eoff + 2) lo = eoff + 2;                        // max_v = (2^m - 1) 2^(e-m) >= 2^(e-1) is a NORMAL number of the dtype
    if (dtype == BFPQ_F16 && lo < 103 + mant_bits) lo = 103 + mant_bits;    // interval >= 2^-24, the smallest fp16
    if (hi > 253) hi = 253;                                  // e <= max exponent + 1 stays finite
    if (hi > 230 + mant_bits) hi = 230 + mant_bits;          // the magic constant 1.5 * 2^(23 + e - m) stays finite
    if (dtype == BFPQ_F16 && hi > 141) hi = 141;             // 2^e finite in fp16
    if (lo > hi) return;
    a.kb_lo = lo - eoff; a.kb_span = hi - lo;
    const uint32_t mpat = ((1u << (mant_bits - 1)) - 1u) << (mb - (mant_bits - 1));
    a.maxv_c = mpat - ((uint32_t)(1 + eoff) << mb);
}

bool fused_shape_ok(int64_t rows, int64_t cols, int dtype, int block_size, int N, int M)
{
    (void)N;
    const int vec = dtype_vec(dtype);
    const int64_t numel = rows * cols;
    if (numel == 0 || numel % vec != 0) return false;
    if (!(M == 0 || M == 2 || M == 4 || M == 8)) return false;   // N:8 = one 16-byte item of a 16-bit dtype, two adjacent items of fp32
    if (M != 0 && cols % M != 0) return false;
    if (block_size == 0) return M != 0;                              // sparsify only
    if (cols % block_size != 0 || block_size % vec != 0) return false;
    const int lpb = block_size / vec;
    return is_pow2(lpb) && lpb <= 64;
}

template <int DT, int NM, bool SFIRST>
int launch_batched(const FusedArgs& a, const BatchArgs& b, hipStream_t s)
{
    const dim3 grid(grid_for((int64_t)b.total_chunks * kThreads)), block(kThreads);
    if (a.lpb == 4) hipLaunchKernelGGL((k_fused_batched<DT, NM, SFIRST, 4>), grid, block, 0, s, a, b);
    else if (a.lpb == 8) hipLaunchKernelGGL((k_fused_batched<DT, NM, SFIRST, 8>), grid, block, 0, s, a, b);
    else hipLaunchKernelGGL((k_fused_batched<DT, NM, SFIRST, -1>), grid, block, 0, s, a, b);
    return (int)hipGetLastError();
}

template <int DT>
int launch_batched_dt(const FusedArgs& a, const BatchArgs& b, int M, bool sfirst, hipStream_t s)
{
    if (M == 0) return launch_batched<DT, 0, true>(a, b, s);
    return sfirst ? launch_batched<DT, 4, true>(a, b, s) : launch_batched<DT, 4, false>(a, b, s);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int bfpq_version(void) { return BFPQ_VERSION; }

int bfpq_tune(int key, int value)
{
    if (key == BFPQ_TUNE_MAX_GRID && value >= 1 && value <= 65535) { bfpq_g_max_grid = value; return 0; }
    if (key == BFPQ_TUNE_GEMM_ROW_TILES && (value == 0 || value == 1 || value == 2 || value == 4)) { bfpq_g_gemm_rt = value; return 0; }
    return BFPQ_E_ARG;
}

const char* bfpq_error_string(int code)
{
    switch (code) {
        case 0: return "ok";
        case BFPQ_E_ARG: return "bfpq: invalid argument";
        case BFPQ_E_UNSUPPORTED: return "bfpq: unsupported configuration";
        case BFPQ_E_ALIGN: return "bfpq: pointer not 16-byte aligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "bfpq: unknown error";
    }
}

int bfpq_exp_window_host(int dtype, uint8_t* table)
{
    if (!table || dtype < 0 || dtype > 2) return BFPQ_E_ARG;
    const int mbits = dtype == BFPQ_F32 ? 23 : (dtype == BFPQ_F16 ? 10 : 7);
    for (int idx = 0; idx < BFPQ_EXP_WIN_ENTRIES; idx++) {
        const int k = idx - 160;
        int win = 0;
        // e(s) for s = 2^k (1 + j 2^-mbits) is k up to some j and k+1 beyond (monotone): scan j upward
        for (int j = 1; j < 256 && j < (1 << mbits); j++) {
            const double s = ldexp(1.0 + ldexp((double)j, -mbits), k);
            const float l32 = (float)log2(s);                         // fp32 log2, correctly rounded
            const float l = h_round(l32, dtype);
            if (ceilf(l) > (float)k) break;
            win = j;
        }
        table[idx] = (uint8_t)win;
    }
    return 0;
}

uint64_t bfpq_nm_prune_mask_host(const uint32_t* keys, int N, int M)
{
    if (!keys || !(N > 0 && M > 0 && N <= M && M <= 64)) return 0;
    uint64_t kv[64];
    for (int i = 0; i < M; i++) kv[i] = ((uint64_t)keys[i] << 8) | (uint64_t)i;
    KvView v{kv, 1};
    return nm_prune_mask(v, N, M);
}

int bfpq_nm4_lut_host(int N, uint8_t* lut)
{
    if (!lut || N < 1 || N > 4) return BFPQ_E_ARG;
    memset(lut, 0x0f, BFPQ_NM4_LUT_ENTRIES);              // unreachable signatures keep everything
    for (uint32_t v = 0; v < 256; v++) {                  // every weak ordering of 4 values appears over {0..3}^4
        const uint32_t k[4] = {v & 3u, (v >> 2) & 3u, (v >> 4) & 3u, (v >> 6) & 3u};
        auto c3 = [](uint32_t a, uint32_t b) { return (uint32_t)(a > b) + (uint32_t)(a >= b); };
        const uint32_t idx = c3(k[0], k[1]) + 3u * c3(k[0], k[2]) + 9u * c3(k[0], k[3]) + 27u * c3(k[1], k[2]) +
                             81u * c3(k[1], k[3]) + 243u * c3(k[2], k[3]);
        const uint64_t prune = bfpq_nm_prune_mask_host(k, N, 4);
        lut[idx] = (uint8_t)(~prune & 0xfu);
    }
    return 0;
}

// every weak ordering of 8 elements (545 835 of them), generated level by level: the elements of the next level all have
// `placed` strictly smaller elements
static void nm8_enumerate(uint32_t remaining, int placed, uint32_t idx, uint32_t* less, int N, uint8_t* lut)
{
    if (!remaining) {
        lut[idx] = (uint8_t)bfpq_nm_prune_mask_host(less, N, 8);       // keys = ranks: same ordering, same ties
        return;
    }
    for (uint32_t sub = remaining; sub; sub = (sub - 1) & remaining) {
        uint32_t id = idx;
        int cnt = 0;
        for (int i = 0; i < 8; i++)
            if ((sub >> i) & 1u) { less[i] = (uint32_t)placed; id |= (uint32_t)placed << (3 * i); cnt++; }
        nm8_enumerate(remaining ^ sub, placed + cnt, id, less, N, lut);
    }
}

int bfpq_nm8_lut_host(int N, uint8_t* lut)
{
    if (!lut || N < 1 || N > 8) return BFPQ_E_ARG;
    memset(lut, 0, BFPQ_NM8_LUT_ENTRIES);
    uint32_t less[8];
    nm8_enumerate(0xffu, 0, 0u, less, N, lut);
    return 0;
}

int bfpq_is_fused(int64_t rows, int64_t cols, int dtype, int block_size, int N, int M)
{
    if (dtype < 0 || dtype > 2) return 0;
    return fused_shape_ok(rows, cols, dtype, block_size, N, M) ? 1 : 0;
}

int bfpq_quantize_nm(const void* in, void* out_deq, void* out_codes, int8_t* out_exp,
                     int64_t rows, int64_t cols, int dtype, int block_size, int mant_bits, double epsilon,
                     int N, int M, int sparsify_first, int code_bits, uint64_t stoch_seed,
                     const uint8_t* exp_win, const uint8_t* nm4_lut, void* scratch, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (dtype < 0 || dtype > 2 || rows < 0 || cols < 0 || block_size < 0) return BFPQ_E_ARG;
    if (M < 0 || N < 0 || (M > 0 && !(N > 0 && N <= M)) || M > 64) return M > 64 ? BFPQ_E_UNSUPPORTED : BFPQ_E_ARG;
    if (rows * cols == 0) return 0;
    if (!in || (!out_deq && !out_codes && !out_exp)) return BFPQ_E_ARG;
    if (block_size > 0 && (mant_bits < 0 || mant_bits > 23 || !exp_win)) return BFPQ_E_ARG;
    if (out_codes && !(code_bits == 4 || code_bits == 8 || code_bits == 16 || code_bits == 32)) return BFPQ_E_ARG;
    if (out_codes && block_size == 0) return BFPQ_E_ARG;
    if (out_codes && ((code_bits == 4 && mant_bits > 3) || (code_bits == 8 && mant_bits > 7) || (code_bits == 16 && mant_bits > 15))) return BFPQ_E_ARG;
    if (block_size == 0 && M == 0) return BFPQ_E_ARG;                   // identity: the caller returns its input
    const float eps_dt = h_round((float)epsilon, dtype);
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out_deq) |
                           reinterpret_cast<uintptr_t>(out_codes)) & 15u) == 0;

    if (aligned && fused_shape_ok(rows, cols, dtype, block_size, N, M) && (M != 4 || nm4_lut)) {
        FusedArgs a;
        a.in = in; a.out_deq = out_deq; a.out_codes = out_codes; a.out_exp = out_exp;
        a.n_items = rows * cols / dtype_vec(dtype);
        a.exp_win = exp_win; a.nm_lut = nm4_lut; a.seed = stoch_seed; a.eps_dt = eps_dt;
        a.lpb = block_size ? block_size / dtype_vec(dtype) : 0;
        a.mant_bits = mant_bits; a.N = N; a.code_bits = code_bits;
        a.force_slow = mant_bits > (dtype == BFPQ_F32 ? 24 : (dtype == BFPQ_F16 ? 11 : 8));
        set_hot16(a, dtype, mant_bits, eps_dt);
        a.selws = nullptr;
        if (dtype == BFPQ_F32) return launch_fused<BFPQ_F32>(a, M, sparsify_first != 0, s);
        if (dtype == BFPQ_F16) return launch_fused<BFPQ_F16>(a, M, sparsify_first != 0, s);
        return launch_fused<BFPQ_BF16>(a, M, sparsify_first != 0, s);
    }

    if (out_codes && code_bits == 32) return BFPQ_E_UNSUPPORTED;       // fp32 image: fused kernel only
    // general path: separate launches.  The quantize stage still takes the flat fused kernel when the shape
    // allows it (e.g. M = 8 on a regular weight: only the N:M replay needs the general kernel).
    if (out_codes && code_bits == 4 && ((block_size & 1) || (M & 1))) return BFPQ_E_UNSUPPORTED;
    auto quantize_stage = [&](const void* src, void* deq) -> int {
        const bool al = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(deq) | reinterpret_cast<uintptr_t>(out_codes)) & 15u) == 0;
        if (al && fused_shape_ok(rows, cols, dtype, block_size, 0, 0)) {
            FusedArgs a;
            a.in = src; a.out_deq = deq; a.out_codes = out_codes; a.out_exp = out_exp;
            a.n_items = rows * cols / dtype_vec(dtype);
            a.exp_win = exp_win; a.nm_lut = nullptr; a.seed = stoch_seed; a.eps_dt = eps_dt;
            a.lpb = block_size / dtype_vec(dtype);
            a.mant_bits = mant_bits; a.N = 0; a.code_bits = code_bits;
            a.force_slow = mant_bits > (dtype == BFPQ_F32 ? 24 : (dtype == BFPQ_F16 ? 11 : 8));
            set_hot16(a, dtype, mant_bits, eps_dt);
            a.selws = nullptr;
            if (dtype == BFPQ_F32) return launch_fused<BFPQ_F32>(a, 0, true, s);
            if (dtype == BFPQ_F16) return launch_fused<BFPQ_F16>(a, 0, true, s);
            return launch_fused<BFPQ_BF16>(a, 0, true, s);
        }
        return launch_quant_rows(src, deq, out_codes, out_exp, rows, cols, dtype, block_size, mant_bits, eps_dt, code_bits, stoch_seed, exp_win, s);
    };
    if (block_size == 0) {                                               // sparsify only
        if (!out_deq) return BFPQ_E_ARG;
        return launch_nm_rows(in, out_deq, nullptr, 0, rows, cols, dtype, N, M, M == 8 ? nm4_lut : nullptr, s);
    }
    if (M == 0) return quantize_stage(in, out_deq);
    void* tmp = out_deq ? out_deq : scratch;
    if (!tmp) return BFPQ_E_ARG;
    int rc;
    if (sparsify_first) {
        rc = launch_nm_rows(in, tmp, nullptr, 0, rows, cols, dtype, N, M, M == 8 ? nm4_lut : nullptr, s);
        if (rc) return rc;
        return quantize_stage(tmp, out_deq);
    }
    rc = quantize_stage(in, tmp);
    if (rc) return rc;
    return launch_nm_rows(tmp, tmp, out_codes, code_bits, rows, cols, dtype, N, M, M == 8 ? nm4_lut : nullptr, s);
}

int bfpq_fake_quantize(const bfpq_plan* p, const void* in, void* out, int64_t rows, int64_t cols, void* stream)
{
    if (!p) return BFPQ_E_ARG;
    return bfpq_quantize_nm(in, out, nullptr, nullptr, rows, cols, p->dtype, p->block_size, p->mant_bits, p->epsilon, p->N, p->M,
                            p->sparsify_first, 0, 0, p->exp_win_dev, p->nm_lut_dev, nullptr, stream);
}

int bfpq_fake_quantize_batched(const bfpq_plan* p, const bfpq_tensor_desc* descs, int n, void* stream)
{
    if (!p || (n > 0 && !descs) || n < 0) return BFPQ_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int dtype = p->dtype;
    if (dtype < 0 || dtype > 2 || p->block_size < 0 || (p->block_size > 0 && !p->exp_win_dev)) return BFPQ_E_ARG;
    const bool any_nm_cfg = p->M > 0;
    // tensors the single-pass kernel takes go into launches of up to kMaxBatch; the rest (ragged shapes, N:8, ...) one by one
    BatchArgs b;
    b.n = 0; b.total_chunks = 0;
    FusedArgs a;
    a.in = nullptr; a.out_deq = nullptr; a.out_codes = nullptr; a.out_exp = nullptr; a.n_items = 0;
    a.exp_win = p->exp_win_dev; a.nm_lut = p->nm_lut_dev; a.seed = 0; a.eps_dt = h_round((float)p->epsilon, dtype);
    a.lpb = p->block_size ? p->block_size / dtype_vec(dtype) : 0;
    a.mant_bits = p->mant_bits; a.N = p->N; a.code_bits = 0;
    a.force_slow = p->mant_bits > (dtype == BFPQ_F32 ? 24 : (dtype == BFPQ_F16 ? 11 : 8));
    set_hot16(a, dtype, p->mant_bits, a.eps_dt);
    a.selws = nullptr;
    bool batch_has_nm = false;
    auto flush = [&]() -> int {
        if (b.n == 0) return 0;
        const int M = batch_has_nm ? 4 : 0;
        int rc;
        if (dtype == BFPQ_F32) rc = launch_batched_dt<BFPQ_F32>(a, b, M, p->sparsify_first != 0, s);
        else if (dtype == BFPQ_F16) rc = launch_batched_dt<BFPQ_F16>(a, b, M, p->sparsify_first != 0, s);
        else rc = launch_batched_dt<BFPQ_BF16>(a, b, M, p->sparsify_first != 0, s);
        b.n = 0; b.total_chunks = 0; batch_has_nm = false;
        return rc;
    };
    for (int i = 0; i < n; i++) {
        const bfpq_tensor_desc& d = descs[i];
        if (d.rows < 0 || d.cols < 0) return BFPQ_E_ARG;
        if (d.rows * d.cols == 0) continue;
        if (!d.in_dev || !d.out_dev) return BFPQ_E_ARG;
        const bool nm = any_nm_cfg && d.apply_nm != 0;
        const int N = nm ? p->N : 0, M = nm ? p->M : 0;
        if (p->block_size == 0 && M == 0) return BFPQ_E_ARG;                // identity: the caller keeps its tensor
        const bool aligned = ((reinterpret_cast<uintptr_t>(d.in_dev) | reinterpret_cast<uintptr_t>(d.out_dev)) & 15u) == 0;
        const bool ok = aligned && (M == 0 || (M == 4 && p->nm_lut_dev)) && p->block_size > 0 &&
                        fused_shape_ok(d.rows, d.cols, dtype, p->block_size, N, M) &&
                        (d.rows * d.cols / dtype_vec(dtype)) % kThreads == 0 &&          // whole chunks of 256 lane items only

                        (d.rows * d.cols / dtype_vec(dtype) + kThreads - 1) / kThreads < ((int64_t)1 << 31);
        if (!ok) {
            const int rc = bfpq_quantize_nm(d.in_dev, d.out_dev, nullptr, nullptr, d.rows, d.cols, dtype, p->block_size, p->mant_bits, p->epsilon,
                                            N, M, p->sparsify_first, 0, 0, p->exp_win_dev, p->nm_lut_dev, nullptr, stream);
            if (rc) return rc;
            continue;
        }
        const int64_t items = d.rows * d.cols / dtype_vec(dtype);
        const int64_t chunks = (items + kThreads - 1) / kThreads;
        if (b.n == kMaxBatch || (int64_t)b.total_chunks + chunks >= ((int64_t)1 << 32)) { const int rc = flush(); if (rc) return rc; }
        BatchDesc& o = b.d[b.n++];
        o.in = d.in_dev; o.out = d.out_dev; o.n_items = items; o.chunk0 = b.total_chunks; o.flags = nm ? 1u : 0u;
        b.total_chunks += (uint32_t)chunks;
        batch_has_nm = batch_has_nm || nm;
    }
    return flush();
}

int bfpq_nm_sparsify(const void* in, void* out, int64_t rows, int64_t cols, int dtype, int N, int M,
                     const uint8_t* nm4_lut, void* stream)
{
    if (!in || !out || dtype < 0 || dtype > 2 || rows < 0 || cols < 0) return BFPQ_E_ARG;
    if (!(N > 0 && M > 0 && N <= M)) return BFPQ_E_ARG;
    if (M > 64) return BFPQ_E_UNSUPPORTED;
    return bfpq_quantize_nm(in, out, nullptr, nullptr, rows, cols, dtype, 0, 0, 0.0, N, M, 1, 0, 0, nullptr, nm4_lut, nullptr, stream);
}

int bfpq_quantize_threshold(const void* in, void* out_deq, void* out_codes, int8_t* out_exp,
                            int64_t rows, int64_t cols, int dtype, int block_size, int mant_bits, double epsilon,
                            int code_bits, uint64_t stoch_seed, const uint8_t* exp_win,
                            void* ws, void* scratch, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (dtype < 0 || dtype > 2 || rows < 0 || cols < 0 || block_size <= 0) return BFPQ_E_ARG;
    if (rows * cols == 0) return 0;
    if (!in || !ws || !exp_win || (!out_deq && !out_codes && !out_exp)) return BFPQ_E_ARG;
    if (mant_bits < 0 || mant_bits > 23) return BFPQ_E_ARG;
    if (out_codes && !(code_bits == 4 || code_bits == 8 || code_bits == 16)) return BFPQ_E_ARG;
    if (out_codes && ((code_bits == 4 && mant_bits > 3) || (code_bits == 8 && mant_bits > 7) || (code_bits == 16 && mant_bits > 15))) return BFPQ_E_ARG;
    const float eps_dt = h_round((float)epsilon, dtype);
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out_deq) |
                           reinterpret_cast<uintptr_t>(out_codes)) & 15u) == 0;
    if (aligned && fused_shape_ok(rows, cols, dtype, block_size, 0, 0)) {
        FusedArgs a;
        a.in = in; a.out_deq = out_deq; a.out_codes = out_codes; a.out_exp = out_exp;
        a.n_items = rows * cols / dtype_vec(dtype);
        a.exp_win = exp_win; a.nm_lut = nullptr; a.seed = stoch_seed; a.eps_dt = eps_dt;
        a.lpb = block_size / dtype_vec(dtype);
        a.mant_bits = mant_bits; a.N = 0; a.code_bits = code_bits;
        a.force_slow = mant_bits > (dtype == BFPQ_F32 ? 24 : (dtype == BFPQ_F16 ? 11 : 8));
        set_hot16(a, dtype, mant_bits, eps_dt);
        a.selws = (SelWs*)ws;
        if (dtype == BFPQ_F32) return launch_fused_threshold<BFPQ_F32>(a, s);
        if (dtype == BFPQ_F16) return launch_fused_threshold<BFPQ_F16>(a, s);
        return launch_fused_threshold<BFPQ_BF16>(a, s);
    }
    void* tmp = out_deq ? out_deq : scratch;
    if (!tmp) return BFPQ_E_ARG;
    if (out_codes && code_bits == 4 && (block_size & 1)) return BFPQ_E_UNSUPPORTED;
    int rc = bfpq_threshold_apply(in, tmp, rows * cols, dtype, ws, stream);
    if (rc) return rc;
    return launch_quant_rows(tmp, out_deq, out_codes, out_exp, rows, cols, dtype, block_size, mant_bits, eps_dt, code_bits, stoch_seed, exp_win, s);
}

}  // extern "C"
