// bfpq_fused.h -- the fused single-pass kernel (k_fused_flat / k_fused_batched) and its launchers, as templates on the dtype.
// Included by bfpq_fused_dt.hip, which is compiled once per dtype (BFPQ_FUSED_DT = 0 f32, 1 f16, 2 bf16) so that the
// ~90 instantiations build side by side; bfpq_kernels.hip dispatches to the three translation units.
#pragma once
#include <type_traits>
#include "bfpq_quant_math.h"
#include "nm_select.h"

namespace bfpq_dev {

// ---------------------------------------------------------------------------------------------
// k_fused_flat: the tensor is a flat array of 16-byte lane items; rows do not matter because
// cols % block == 0 (and cols % M == 0).  NM in {0,2,4}.  One HBM read, one HBM write per output.
//   hot path per item: [packed 3-way comparisons -> 729-entry LDS table -> AND masks], packed abs/max,
//   DPP group max, exponent from a 320-byte LDS table, mul / rndne / med3 / mul per element, pack.
//   Anything unusual in a block (non-finite or zero max, scale outside the normal range, mantissa
//   wider than the dtype) makes the whole wavefront replay that item through the step-by-step
//   emulation (quant_elem); the branch is wave-uniform and never taken on ordinary weights.
// ---------------------------------------------------------------------------------------------
template <int DT, int NM, bool SFIRST, bool STOCH, int LPBT, bool DEQ_ONLY, bool BATCHED, bool F32IMG = false, int PACK = 0>
__device__ __forceinline__ void fused_flat_body(const FusedArgs& a, [[maybe_unused]] const BatchArgs* b)
{
    using T = Traits<DT>;
    constexpr int VEC = T::VEC;
    constexpr bool PACK4 = PACK != 0;                            // packed-only output: 4-bit codes + int8 exponents (PACK == 4) or
    constexpr bool MX8 = PACK == 8;                              // e4m3 mantissas + E8M0 block scales (PACK == 8: the matrix unit's image)
    __shared__ __attribute__((aligned(16))) uint8_t s_win[512];
    __shared__ uint2 s_mask[NM == 4 && VEC == 8 ? 736 : 1];      // 16-bit dtypes: AND masks for the two dwords of a group
    __shared__ uint8_t s_keep[NM == 4 && VEC == 4 ? 736 : 1];    // fp32: 4-bit keep mask
    __shared__ uint64_t s_kv[NM == 8 ? 8 * kThreads : 1];        // N:8: column per thread for the nth_element replay (rare)
    __shared__ uint4 s_f32t[F32IMG ? kThreads / 64 : 1][F32IMG ? 128 : 1];   // fp32 image: a wave's 2 KB on their way to hole-free stores
    // (the tables are filled further down, behind the first tile's load: one memory round trip for everything)

    const bool do_quant = a.lpb > 0;
    constexpr int kLead = NM == -1 ? kCutWGs : 0;                 // threshold mode: the first workgroups of the grid own the cut segment
    const int64_t stride = (int64_t)(gridDim.x - kLead) * kThreads;
    const int64_t n_round = (a.n_items + kThreads - 1) / kThreads * kThreads;   // uniform trip count per block
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(a.in);

    // N:M mask on the 4 dwords of an item (16-bit dtypes: 2 groups of 4; fp32: 1 group)
    ThrCtx thr;
    bool item_valid = true;
    auto nm_mask = [&](uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3) __attribute__((always_inline)) {
        if constexpr (NM == -1) {                       // global magnitude threshold (unstructured, bfp_ops.py:61-71)
            if (__builtin_expect(!thr.ranked, 1)) {
                // ordinary workgroups: the ties of this tile all go (in front of the cut segment) or all stay, i.e. one
                // comparison against tau + 1 or tau (thr.teff, set per tile by the caller; tau == 0 when nothing is pruned)
                const uint32_t teff = thr.teff;
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
            } else {                                    // cut workgroups: rank the ties of the tile
                uint32_t raw[VEC];
                if constexpr (VEC == 4) { raw[0] = d0; raw[1] = d1; raw[2] = d2; raw[3] = d3; }
                else {
                    raw[0] = d0 & 0xffffu; raw[1] = d0 >> 16; raw[2] = d1 & 0xffffu; raw[3] = d1 >> 16;
                    raw[4] = d2 & 0xffffu; raw[5] = d2 >> 16; raw[6] = d3 & 0xffffu; raw[7] = d3 >> 16;
                }
                const uint32_t prune = thr_rank_bits<DT>(raw, item_valid, thr);
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
        uint32_t d0 = cur.x, d1 = cur.y, d2 = cur.z, d3 = cur.w;
        [[maybe_unused]] bool cut_skip = false;                   // NM == -1, ordinary workgroup on a tile of the cut segment: its store goes to the dump
        if constexpr (NM == -1) {
            if (!thr.ranked) {
                const int64_t tile0 = uniform64(item - (threadIdx.x & 63));     // the wave's 64 lanes hold one aligned tile
                thr.teff = thr.tau + (tile0 < thr.cut_lo ? 1u : 0u);
                cut_skip = tile0 >= thr.cut_lo && tile0 < thr.cut_hi;
            }
        }

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
        [[maybe_unused]] uint32_t pack_w = 0, pack_w2 = 0;        // PACK4: the item's eight 4-bit codes (MX8: eight e4m3 bytes), the block's exponent byte
        [[maybe_unused]] int pack_e = 0;
        [[maybe_unused]] bool pack_hot = false;
        // packed item of the general tiers, from code[] / e_blk / nan_blk.  Called INSIDE their branch where nothing comes between
        // (dense or sparsify-first): with the packing behind the merge, the eight codes were live across it and the hot path paid
        // eight register moves per item for values it never uses.
        [[maybe_unused]] auto pack_cold = [&]() __attribute__((always_inline)) {
            if constexpr (MX8) {
                auto b8 = [](float cf) { const int c = (int)cf; const uint32_t m = (uint32_t)(c < 0 ? -c : c);
                                         const uint32_t v = m >= 8 ? 0x48u + m : (m >= 4 ? 0x40u + 2u * m : (m >= 2 ? 0x38u + 4u * m : 0x38u));
                                         return (m ? v : 0u) | (c < 0 ? 0x80u : 0u); };
                pack_w = pack_w2 = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { pack_w |= b8(code[j]) << (8 * j); pack_w2 |= b8(code[(4 + j) % VEC]) << (8 * j); }
                pack_e = nan_blk ? 0x7fffffff : e_blk;                       // (the E8M0 conversion below maps the marker to 0xff)
            } else if constexpr (PACK4) {
                uint32_t w = 0;
#pragma unroll
                for (int j = 0; j < VEC; j++) w |= ((uint32_t)(int)code[j] & 0xfu) << (4 * j);
                pack_w = w;
                pack_e = nan_blk ? -128 : (e_blk < -127 ? -127 : (e_blk > 127 ? 127 : e_blk));
            }
        };
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
            if constexpr (VEC == 8 && !STOCH && (DEQ_ONLY || PACK4)) {
                h16 = hot16_scale<DT>(mx, a, s_win);
                hot = !__any(!h16.ok);
            }
            pack_hot = hot;
            if (__builtin_expect(hot, 1)) {
                if constexpr (VEC == 8 && !STOCH && PACK4) {
                    // Packed output from the lean arithmetic: min(|x|, max_v) + C has the integer mantissa in the LOW bits of
                    // its float image (ulp(C) = interval and the low 16 bits of C's image are zero), so the magnitude codes of
                    // two elements are one v_perm of the two sums; sign -> two's complement on packed halves; eight nibbles
                    // -> one dword.  The eight block exponents of the wave leave as ONE 8-byte store.
                    static_assert(LPBT == 8, "packed 4-bit instantiation: block = 8 lane items");
                    const uint32_t absm = T::ABS | (T::ABS << 16);
                    uint32_t dd[4] = {d0, d1, d2, d3};
                    if constexpr (!MX8 && (NM == 0 || SFIRST)) {
                        // 4-bit codes straight from the SIGNED values: clamp(x, +-max_v) + C with C = 1.5 * 2^23 * interval leaves
                        // round(x / interval) as a two's-complement integer in the low bits of the float image (the classic
                        // float -> int constant), so the low nibble of each sum IS the code: no sign fix-up afterwards.  The
                        // clamp: two v_med3_f32 per dword (bf16, after the unpack) or v_pk_max/min_f16 on the packed halves (fp16).
                        typedef float float2v __attribute__((ext_vector_type(2)));
                        const float2v C2 = {h16.C, h16.C};
                        uint32_t g[4];
#pragma unroll
                        for (int x = 0; x < 4; x++) {
                            float2v v;
                            if constexpr (DT == BFPQ_BF16) {
                                const float mv = u2f(h16.maxv2 << 16);
                                v = (float2v){__builtin_amdgcn_fmed3f(u2f(dd[x] << 16), -mv, mv), __builtin_amdgcn_fmed3f(u2f(dd[x] & 0xffff0000u), -mv, mv)} + C2;
                            } else {
                                uint32_t cl;
                                asm("v_pk_min_f16 %0, %1, %2" : "=v"(cl) : "v"(dd[x]), "v"(h16.maxv2));
                                asm("v_pk_max_f16 %0, %1, %2" : "=v"(cl) : "v"(cl), "v"(h16.maxv2 | 0x80008000u));
                                v = (float2v){fma_mix_f16<false>(cl, h16.C), fma_mix_f16<true>(cl, h16.C)};
                            }
                            g[x] = __builtin_amdgcn_perm(f2u(v.y), f2u(v.x), 0x0c0c0400u);        // [lo.byte0, hi.byte0, 0, 0]
                        }
                        uint32_t p0 = __builtin_amdgcn_perm(g[1], g[0], 0x05040100u) & 0x0f0f0f0fu;    // the low nibbles of elements 0..3 / 4..7, one per byte
                        uint32_t p1 = __builtin_amdgcn_perm(g[3], g[2], 0x05040100u) & 0x0f0f0f0fu;
                        p0 |= p0 >> 4; p1 |= p1 >> 4;                                                  // bytes 0 and 2 now hold two nibbles each
                        pack_w = __builtin_amdgcn_perm(p1, p0, 0x06040200u);
                        pack_e = h16.e;
                    } else {
                    uint32_t q[4];
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const uint32_t am = pk_min_u16(dd[x] & absm, h16.maxv2);
                        typedef float float2v __attribute__((ext_vector_type(2)));
                        const float2v C2 = {h16.C, h16.C};
                        float2v v;
                        if constexpr (DT == BFPQ_BF16) v = (float2v){u2f(am << 16), u2f(am & 0xffff0000u)} + C2;
                        else v = (float2v){fma_mix_f16<false>(am, h16.C), fma_mix_f16<true>(am, h16.C)};
                        q[x] = __builtin_amdgcn_perm(f2u(v.y), f2u(v.x), 0x05040100u);             // (q_hi << 16) | q_lo
                    }
                    if constexpr (NM != 0 && !SFIRST) nm_mask(q[0], q[1], q[2], q[3]);   // Q before S: inside a block codes order like values
                    if constexpr (MX8) {
                        // e4m3 byte of an integer magnitude m: one v_perm into an 8-byte table (m <= 7), 0x48 + m above;
                        // the sign bits are the high bytes of the input halves
                        auto img = [&](uint32_t qa, uint32_t qb, uint32_t da, uint32_t db) __attribute__((always_inline)) {
                            const uint32_t m = __builtin_amdgcn_perm(qb, qa, 0x06040200u) & 0x0f0f0f0fu;
                            uint32_t e = __builtin_amdgcn_perm(0x4e4c4a48u, 0x44403800u, m & 0x07070707u);
                            if (a.mant_bits > 3) e = bfi_b32(((m >> 3) & 0x01010101u) * 0xffu, m + 0x48484848u, e);
                            return e | (__builtin_amdgcn_perm(db, da, 0x07050301u) & 0x80808080u);
                        };
                        pack_w = img(q[0], q[1], dd[0], dd[1]);
                        pack_w2 = img(q[2], q[3], dd[2], dd[3]);
                        pack_e = h16.e;
                    } else {
                    uint32_t c[4];
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const uint32_t m = pk_ashr_i16_s(dd[x], 0x000f000fu);                     // 0xffff where the element is negative
                        c[x] = pk_sub_i16(q[x] ^ m, m);                                            // two's complement per half
                    }
                    // eight low nibbles -> one dword: low bytes of the halves side by side, then nibbles of adjacent bytes together
                    uint32_t p0 = __builtin_amdgcn_perm(c[1], c[0], 0x06040200u) & 0x0f0f0f0fu;   // [c0.lo, c0.hi, c1.lo, c1.hi]
                    uint32_t p1 = __builtin_amdgcn_perm(c[3], c[2], 0x06040200u) & 0x0f0f0f0fu;
                    p0 |= p0 >> 4; p1 |= p1 >> 4;                                                  // bytes 0 and 2 now hold two nibbles each
                    pack_w = __builtin_amdgcn_perm(p1, p0, 0x06040200u);
                    pack_e = h16.e;
                    }
                    }
                } else if constexpr (VEC == 8 && !STOCH && DEQ_ONLY) {
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
                    // stochastic rounding: the per-item key goes once through the two-multiply mixer, then one xorshift32 step
                    // yields the two 16-bit dithers of a pair of elements (32-bit integer multiplies are quarter rate on this
                    // part: the per-element mixer made 'stoc' VALU-bound); a dither is built as the float 1.f + u 2^-16 and the
                    // -1.5 that centres it rides in the scaling FMA
                    [[maybe_unused]] uint32_t rstate = 0u;
                    if constexpr (STOCH) {
                        uint32_t x = rng_item_key(a.seed, (uint64_t)item * VEC);
                        x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
                        rstate = x | 1u;
                    }
    #pragma unroll
                    for (int j = 0; j < VEC; j += 2) {                     // two elements per v_pk_mul_f32
                        float2v x;
                        if constexpr (DT == BFPQ_BF16) {
                            const uint32_t d = j < 2 ? d0 : (j < 4 ? d1 : (j < 6 ? d2 : d3));
                            x = (float2v){u2f(d << 16), u2f(d & 0xffff0000u)};
                        } else x = (float2v){raw_to_f32<DT>(raw[j]), raw_to_f32<DT>(raw[j + 1])};
                        float2v t;
                        if constexpr (STOCH) {
                            rstate ^= rstate << 13; rstate ^= rstate >> 17; rstate ^= rstate << 5;
                            const float2v u = {u2f(0x3f800000u | ((rstate & 0xffffu) << 7)), u2f(0x3f800000u | ((rstate >> 16) << 7))};   // [1, 2)
                            t = __builtin_elementwise_fma(x, (float2v){fs.inv, fs.inv}, (float2v){-1.5f, -1.5f}) + u;   // x / interval + uniform[-0.5, 0.5)
                        } else t = x * (float2v){fs.inv, fs.inv};
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
                if constexpr (PACK4 && (NM == 0 || SFIRST)) pack_cold();
            }
        }
        if constexpr (NM != 0 && !SFIRST) if (!(PACK4 && pack_hot)) {                // Q before S (bfp_ops.py:146-149)
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
        if constexpr (PACK4) {
            // the packed item from whichever tier ran; then stores that are unconditional in form (a store inside a branch makes
            // hipcc drain the memory queue at the loop top): every lane its code dword; the block's exponent byte from the first
            // lane of each block
            if constexpr (NM != 0 && !SFIRST) { if (!pack_hot) pack_cold(); }       // (else: done inside the general tier's branch)
            if constexpr (MX8) {                                  // E8M0 scale of the block: 2^(e - mant_bits), NaN block -> 0xff
                if (pack_e == 0x7fffffff) pack_e = 0xff;
                else { const int sc = pack_e - a.mant_bits + 127; pack_e = sc < 0 ? 0 : (sc > 254 ? 254 : sc); }
            }
            if constexpr (!GUARD) {
                // (32-bit byte offsets from a uniform base: one shift per address; the launcher cuts tensors beyond 2^28 items)
                // (A/B on one box, tools_dev/ab_packed.py: without the code stores 22.3 us, without the exponent stores no change,
                //  non-temporal code stores +0.7 us, three sweeps of loads ahead instead of two +1.3 us)
                if constexpr (MX8) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.out_codes) + (uint32_t)item * 8u) = make_uint2(pack_w, pack_w2);
                else *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.out_codes) + (uint32_t)item * 4u) = pack_w;
                // the block's exponent byte: ONE byte store per wave-instruction with only the first lane of each block enabled
                // (exec narrowed around it by hand -- a branch would make hipcc drain the memory queue at the loop top; the eight
                // v_readlane + scalar packing this replaces cost 9 vector instructions per item)
                {
                    unsigned long long saved;
                    const unsigned long long lead = 0x0101010101010101ull;
                    const uint32_t eoff = (uint32_t)item >> 3;
                    asm volatile("s_and_saveexec_b64 %0, %4\n\tglobal_store_byte %1, %2, %3\n\ts_mov_b64 exec, %0"
                                 : "=&s"(saved) : "v"(eoff), "v"(pack_e), "s"(a.out_exp), "s"(lead) : "memory");                 // (exec is restored inside)
                }
            } else if (valid) {
                if constexpr (MX8) reinterpret_cast<uint2*>(a.out_codes)[item] = make_uint2(pack_w, pack_w2);
                else reinterpret_cast<uint32_t*>(a.out_codes)[item] = pack_w;
                if ((item & 7) == 0) a.out_exp[item >> 3] = (int8_t)pack_e;
            }
            return;
        }
        if constexpr (USE_BUF) { buf_res = make_uint4(o0, o1, o2, o3); return; }   // (the sweep stores it)
        if constexpr (F32IMG && NM == -1) { if (cut_skip) return; }
        if constexpr (F32IMG) {
            // the fp32 image of the result and nothing else (what 'stoc' rounding of a half tensor returns): two
            // unconditional-in-form stores per item, like the drop-in mode's one
            static_assert(VEC == 8, "fp32 image: 16-bit inputs");
            const uint32_t od[4] = {o0, o1, o2, o3};
            uint32_t f[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j++) f[j] = f2u(raw_to_f32<DT>((od[j >> 1] >> (16 * (j & 1))) & 0xffffu));
            // A lane's eight results are 32 bytes; stored as they lie, each of the two store instructions of a wave writes 16 bytes of
            // every 32 (measured on the 'int' format's kernels: launches up to 1.7 x slower, at random).  The wave's 2 KB go through a
            // wave-private LDS tile instead and leave as two instructions of 1 KB without holes (a wave's items are consecutive:
            // its tile starts at item - lane; LDS operations of one wave execute in order).
            uint4* tile = &s_f32t[threadIdx.x >> 6][0];
            const int lane = threadIdx.x & 63;
            tile[2 * lane] = make_uint4(f[0], f[1], f[2], f[3]);
            tile[2 * lane + 1] = make_uint4(f[4], f[5], f[6], f[7]);
            const uint4 lo = tile[lane], hi = tile[64 + lane];        // lane l: bytes [16 l, 16 l + 16) of the first and of the second KB
            uint4* dst = reinterpret_cast<uint4*>(a.out_codes) + (item - lane) * 2;
            // the lane that holds item j's halves: lanes 2 (j - first) and 2 (j - first) + 1 of the first KB for j - first < 32, ...
            const int64_t n_it = a.n_items;
            const int64_t it_lo = item - lane + (lane >> 1), it_hi = it_lo + 32;
            if (it_lo < n_it) dst[lane] = lo;
            if (it_hi < n_it) dst[64 + lane] = hi;
            return;
        }
        if constexpr (DEQ_ONLY) {                                           // hot mode: exactly one store per item
            uint4* dst = reinterpret_cast<uint4*>(out_deq) + item;
            if constexpr (NM == -1) dst = cut_skip ? thr.dump + (threadIdx.x & 63) : dst;       // (keeps the store unconditional in form)
            if (valid) stream_store(dst, make_uint4(o0, o1, o2, o3));
            return;
        }
        if (!valid) return;
        if constexpr (NM == -1) { if (cut_skip) return; }
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
            } else if (a.code_bits == kCodeBitsE4M3) {                  // e4m3 image bytes (fp32 tensors; the 16-bit dtypes have the lean PACK = 8 instantiation)
                auto b8 = [](int cc) { const uint32_t m = (uint32_t)(cc < 0 ? -cc : cc);
                                       const uint32_t v = m >= 8 ? 0x48u + m : (m >= 4 ? 0x40u + 2u * m : (m >= 2 ? 0x38u + 4u * m : 0x38u));
                                       return (m ? v : 0u) | (cc < 0 ? 0x80u : 0u); };
                uint32_t w0 = 0, w1 = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) w0 |= b8(c[j]) << (8 * j);
                if constexpr (VEC == 8) {
#pragma unroll
                    for (int j = 0; j < 4; j++) w1 |= b8(c[4 + j]) << (8 * j);
                    reinterpret_cast<uint2*>(a.out_codes)[item] = make_uint2(w0, w1);
                } else reinterpret_cast<uint32_t*>(a.out_codes)[item] = w0;
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
            if (a.code_bits == kCodeBitsE4M3) {                           // E8M0 scale of the block: 2^(e - mant_bits); NaN block -> 0xff
                const int sc = e_blk - a.mant_bits + 127;
                a.out_exp[item / a.lpb] = (int8_t)(nan_blk ? 0xff : (sc < 0 ? 0 : (sc > 254 ? 254 : sc)));
            } else {
                const int es = e_blk < -127 ? -127 : (e_blk > 127 ? 127 : e_blk);
                a.out_exp[item / a.lpb] = nan_blk ? (int8_t)-128 : (int8_t)es;
            }
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
        // (the flat sweep's loop-top trick -- one more memory operation behind the first load, so that the loop-top wait leaves the previous
        // store in flight -- measured SLOWER here: ViT-L's 144 weights 434 against 417 us, 64 x [4096,11008] 2238 against 2113, tools_dev/ab_list.py)
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
    // (the packed-output instantiations are bound by vector-instruction issue, not by memory: they index with 32 bits -- byte
    // offsets from a uniform base, one shift per address instead of 64-bit compare / select / shift-add chains; the launcher
    // cuts a tensor beyond 2^28 lane items into pieces)
    constexpr bool IDX32 = PACK4 && !BATCHED;
    using idx_t = std::conditional_t<IDX32, uint32_t, int64_t>;
    const idx_t last = (idx_t)(a.n_items - 1);
    auto fetch = [&](idx_t i) __attribute__((always_inline)) {
        const idx_t ic = i < last ? i : last;
        if constexpr (IDX32) return __builtin_nontemporal_load(reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(src) + (uint32_t)(ic * 16u)));
        else return __builtin_nontemporal_load(reinterpret_cast<const u4v*>(src + ic));
    };
    const idx_t strd = (idx_t)stride, n_rnd = (idx_t)n_round;
    const idx_t full = (idx_t)a.n_items / strd;                            // sweeps in which every thread has an item
    const bool cut_wg = NM == -1 && (int)blockIdx.x < kLead;
    idx_t item = cut_wg ? (idx_t)threadIdx.x : (idx_t)((int)blockIdx.x - kLead) * kThreads + threadIdx.x;
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
        thr_setup<DT>(thr, a.selws, a.n_items, s_part, s_res);
        if (cut_wg) {
            // this workgroup owns a share of the cut segment's tiles: ties in front of them counted, then each tile ranked
            cut_wg_run<DT, true>(thr, a.in, a.n_items * VEC, a.n_items, s_part, [&](int64_t it, const uint32_t* raw, bool, uint32_t before) __attribute__((always_inline)) {
                thr.ranked = true; thr.before = before;
                uint4 cur;
                if constexpr (VEC == 4) cur = make_uint4(raw[0], raw[1], raw[2], raw[3]);
                else cur = make_uint4(raw[0] | (raw[1] << 16), raw[2] | (raw[3] << 16), raw[4] | (raw[5] << 16), raw[6] | (raw[7] << 16));
                body(std::true_type{}, it, cur);
            });
            return;
        }
    }
    // One more memory op behind the first load, result unused.  At the loop top the back edge arrives with [load, store]
    // outstanding and the entry edge with [load] only; one s_waitcnt immediate must serve both edges, so the compiler
    // emitted vmcnt(0) and every wave waited for its just-issued store once per iteration.  With [load, dummy] on the
    // entry edge both edges need vmcnt(1) and the store stays in flight across the loop top (A/B: 32.5 -> 31.95 us).
    // NB the scheduler still hoists the tile's first v_perm above the next prefetch, so a wave has ONE load in flight,
    // issued when the previous arrives; pinning the prefetch in front of that wait (two loads in flight) measured
    // SLOWER (33.25 us) -- like every other variant with more reads in flight per wave on this part.  (r03: also the second tile's
    // load issued once, in front of the table set-up, to fill the ramp of the launch: 31.90 against 31.54 us interleaved.)
    asm volatile("" ::: "memory");                                         // (pins the dummy between the first load and the loop)
    const uint32_t dummy = *reinterpret_cast<const uint32_t*>(src);
    asm volatile("" ::: "memory");
    idx_t sweep = 0;
    if constexpr (IDX32) {                  // (r03: the same two-sweeps-ahead loop for the N:8 instantiation: 43.60 against 43.41 us, not adopted)
        // The packed-output instantiations are not at the memory pipe's limit (a quarter / half of the bytes written) but at
        // what ONE 1-KB load in flight per wave sustains against ~2 us of latency (8 waves x 4 SIMDs x 256 CUs x 1 KB = 8.4 MB in
        // flight ~ 4 TB/s, which is what they measured).  So: TWO sweeps ahead, three register sets alternating by name.
        u4v c1 = fetch(item + strd);
        for (; sweep + 3 <= full; sweep += 3, item += 3 * strd) {
            const u4v c2 = fetch(item + 2 * strd);
            body(std::false_type{}, (int64_t)item, u4(c0));
            c0 = fetch(item + 3 * strd);
            body(std::false_type{}, (int64_t)(item + strd), u4(c1));
            c1 = fetch(item + 4 * strd);
            body(std::false_type{}, (int64_t)(item + 2 * strd), u4(c2));
        }
        for (; item < n_rnd; item += strd) {
            const u4v c2 = fetch(item + 2 * strd);
            body(std::true_type{}, (int64_t)item, u4(c0));
            c0 = c1; c1 = c2;
        }
        asm volatile("" : : "v"(dummy));
        return;
    }
    // main loop: load one sweep ahead; unrolled by two so that the two register sets alternate by NAME
    // (copying a register that a load in flight will write forces vmcnt(0)); with nothing conditional in
    // the body the waits are counted and the previous store stays in flight across the loop top
    for (; sweep + 2 <= full; sweep += 2, item += 2 * strd) {
        const u4v c1 = fetch(item + strd);
        body(std::false_type{}, (int64_t)item, u4(c0));
        c0 = fetch(item + 2 * strd);
        body(std::false_type{}, (int64_t)(item + strd), u4(c1));
    }
    // remaining full sweep (0..1) and the ragged last one: guarded, rolled (block-uniform trip count)
    for (; item < n_rnd; item += strd) {
        const u4v c1 = fetch(item + strd);
        body(std::true_type{}, (int64_t)item, u4(c0));
        c0 = c1;
    }
    asm volatile("" : : "v"(dummy));                                       // the dummy's only "use": after all the work
}

template <int DT, int NM, bool SFIRST, bool STOCH, int LPBT, bool DEQ_ONLY, bool F32IMG = false, int PACK = 0>
__global__ void __launch_bounds__(kThreads) k_fused_flat(const FusedArgs a)
{
    fused_flat_body<DT, NM, SFIRST, STOCH, LPBT, DEQ_ONLY, false, F32IMG, PACK>(a, nullptr);
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
    if constexpr (!STOCH && Traits<DT>::VEC == 8 && NM != 2 && NM != 8) {
        // packed 4-bit codes + exponents only, block = 8 lane items: the lean packed instantiation
        if (!a.out_deq && a.out_codes && a.code_bits == 4 && a.out_exp && a.lpb == 8) {
            // (32-bit indexing inside: pieces of 2^27 lane items -- whole blocks, whole exponent bytes)
            const int64_t piece = (int64_t)1 << 27;
            for (int64_t i0 = 0; i0 < a.n_items; i0 += piece) {
                FusedArgs b = a;
                b.in = reinterpret_cast<const char*>(a.in) + i0 * 16;
                b.out_codes = reinterpret_cast<char*>(a.out_codes) + i0 * 4;
                b.out_exp = a.out_exp + i0 / 8;
                b.n_items = a.n_items - i0 < piece ? a.n_items - i0 : piece;
                const dim3 grid(grid_for_packed(b.n_items)), block(kThreads);
                hipLaunchKernelGGL((k_fused_flat<DT, NM, SFIRST, false, 8, false, false, 4>), grid, block, 0, s, b);
            }
            return (int)hipGetLastError();
        }
    }
    if constexpr (STOCH && Traits<DT>::VEC == 8) {
        if (!a.out_deq && a.out_codes && a.code_bits == 32 && !a.out_exp) {      // fp32 image only: its own store-lean instantiation
            const dim3 grid(grid_for(a.n_items)), block(kThreads);
            hipLaunchKernelGGL((k_fused_flat<DT, NM, SFIRST, true, -1, false, true>), grid, block, 0, s, a);
            return (int)hipGetLastError();
        }
    }
    return launch_fused_o<DT, NM, SFIRST, STOCH, false>(a, s);
}

// dense quantize of a 16-bit tensor straight into the block-scaled matrix unit's operand image (e4m3 + E8M0), block = 8 lane items
template <int DT>
int launch_fused_mx8(const FusedArgs& a, hipStream_t s)
{
    if constexpr (Traits<DT>::VEC == 8) {
        const int64_t piece = (int64_t)1 << 27;               // (32-bit indexing inside: see launch_fused_l)
        for (int64_t i0 = 0; i0 < a.n_items; i0 += piece) {
            FusedArgs b = a;
            b.in = reinterpret_cast<const char*>(a.in) + i0 * 16;
            b.out_codes = reinterpret_cast<char*>(a.out_codes) + i0 * 8;
            b.out_exp = a.out_exp + i0 / 8;
            b.n_items = a.n_items - i0 < piece ? a.n_items - i0 : piece;
            const dim3 grid(grid_for_cap(b.n_items, (int64_t)kMaxGrid * 43 / 16)), block(kThreads);       // (the image writes twice the codes' bytes: 2752 measured best, 14.3 vs 14.8 us at 2048)
            hipLaunchKernelGGL((k_fused_flat<DT, 0, true, false, 8, false, false, 8>), grid, block, 0, s, b);
        }
        return (int)hipGetLastError();
    } else return BFPQ_E_UNSUPPORTED;
}

template <int DT>
int launch_fused_threshold(const FusedArgs& a, hipStream_t s)
{
    const dim3 grid(kCutWGs + grid_for(a.n_items)), block(kThreads);        // (the cut segment's workgroups come first)
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
    // N:8 on a 16-bit tensor, pruned before it is quantized, is bound by its ~150 packed comparisons per item, not by HBM: twice the workgroups
    // of the drop-in grid keep the VALU fed while others wait for memory ([4096,11008] bf16 4:8 s, sweep of the cap on one box: 1024 -> 52.0 us,
    // 1536 -> 50.2, 2048 -> 47.7, 2752 -> 48.2).  Quantize-first inputs (every group ties at the cut and reads the 16 MiB rank table) and fp32
    // (already at 63 % of HBM) lose 3 % with the larger grid and keep the drop-in one.
    const dim3 grid(grid_for_cap(a.n_items, (int64_t)kMaxGrid * ((Traits<DT>::VEC == 8 && SFIRST) ? 2 : 1))), block(kThreads);
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

}  // namespace bfpq_dev
