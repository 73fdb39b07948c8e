// bfpq_kernels.hip -- gfx950 kernels + C-ABI launchers of libbfpq.so (see include/bfpq.h).
//
// Reference path replaced: src/transformers/bfp/bfp_ops.py:16-149.
//
// Kernels
//   k_fused_flat   the hot kernel: [N:M mask] + shared exponent + mantissa rounding in ONE pass.
//                  16 B per lane per access, a block of `block_size` elements lives in `lpb`
//                  adjacent lanes of one wavefront (block 64 bf16 = 8 lanes), block max by
//                  cross-lane xor-shuffles, N:M keep-mask from a 729-entry LDS table, no atomics,
//                  no LDS staging of data, no second read.  HBM-bound: 2 x sizeof(dtype) B/elem.
//   k_nm_rows      general N:M (any M <= 64, ragged rows): one thread per group, the group's
//                  (key,index) pairs in an LDS column, libstdc++ introselect replayed (nm_select.h)
//   k_quant_rows   general HBFP quantizer (any block size, ragged rows): a power-of-two lane group
//                  per block, two sweeps (max, then quantize)
//   k_select_*     radix select of the k-th smallest magnitude (LDS histogram, then a 1-block scan)
//   k_tie_count / k_tie_scan / k_threshold_apply   ordered tie ranks + zeroing for unstructured pruning
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <type_traits>
#include "bfpq.h"
#include "bfpq_common.h"
#include "nm_select.h"

using namespace bfpq;

extern "C" __attribute__((visibility("hidden"))) int bfpq_g_gemm_rt;   // bfpq_gemm.hip

namespace {

#ifndef BFPQ_MAXGRID
#define BFPQ_MAXGRID 1024          // 256 CUs x 4 workgroups, grid-stride beyond that (A/B over 5 shapes: 1024 best or tied)
#endif
#ifndef BFPQ_NT
#define BFPQ_NT 1                  // non-temporal loads/stores on the once-touched streams (A/B: +6..8 %)
#endif
constexpr int kThreads = 256;
int g_max_grid = BFPQ_MAXGRID;           // tuning knob (bfpq_tune), process-wide
#define kMaxGrid g_max_grid

__device__ __forceinline__ uint4 stream_load(const uint4* p)
{
#if BFPQ_NT
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void stream_store(uint4* p, uint4 v)
{
#if BFPQ_NT
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u4v*>(p));
#else
    *p = v;
#endif
}

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
typedef short short2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cmp3(uint32_t a, uint32_t b) { return (uint32_t)(a > b) + (uint32_t)(a >= b); }

// signature index of one group of 4 magnitude keys (see bfpq_nm4_lut_host): sum_p c_p 3^p, c in {0,1,2}
__device__ __forceinline__ uint32_t nm4_index(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3)
{
    // keys < 2^31: the signed difference clamped to [-1,1] is the 3-way comparison
    auto c = [](uint32_t a, uint32_t b) { const int d = (int)a - (int)b; return d < -1 ? -1 : (d > 1 ? 1 : d); };
    return (uint32_t)(364 + c(k0, k1) + 3 * c(k0, k2) + 9 * c(k0, k3) + 27 * c(k1, k2) + 81 * c(k1, k3) + 243 * c(k2, k3));
}

// keep-mask of one group of 2 (keep 1): stable insertion sort of two -> index 0 goes on a tie
__device__ __forceinline__ uint32_t nm2_keep(uint32_t k0, uint32_t k1, int N)
{
    if (N >= 2) return 3u;
    return (k1 < k0) ? 1u : 2u;
}

// packed 16-bit VALU ops, spelled out: hipcc scalarises a clamp written with vector builtins into
// per-half v_cmp / v_cndmask chains (seen in the ISA of the first version of this kernel)
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_mad_i16(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ uint32_t pk_add_i16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_add_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// same with the constant operand in an SGPR (one scalar operand per VALU instruction is allowed)
__device__ __forceinline__ uint32_t pk_ashr_i16_s(uint32_t a, uint32_t sh) { uint32_t d; asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(d) : "s"(sh), "v"(a)); return d; }
__device__ __forceinline__ uint32_t pk_max_i16_s(uint32_t a, uint32_t k) { uint32_t d; asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k)); return d; }
__device__ __forceinline__ uint32_t pk_min_i16_s(uint32_t a, uint32_t k) { uint32_t d; asm("v_pk_min_i16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k)); return d; }
__device__ __forceinline__ uint32_t pk_mad_i16_s(uint32_t a, uint32_t k, uint32_t c) { uint32_t d; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c)); return d; }

// max over the 2^n adjacent lanes that share one block, by DPP where the ISA has a pattern for it
template <int CTRL> __device__ __forceinline__ uint32_t dpp_max(uint32_t v)
{
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
    return o > v ? o : v;
}
__device__ __forceinline__ uint32_t shfl_max(uint32_t v, int o)
{
    const uint32_t other = (uint32_t)__shfl_xor((int)v, o, 64);
    return other > v ? other : v;
}
// LPBT > 0: lanes per block known at compile time; LPBT < 0: runtime value lpb
template <int LPBT> __device__ __forceinline__ uint32_t group_max(uint32_t v, int lpb)
{
    const int n = LPBT > 0 ? LPBT : lpb;
    if (n >= 2) v = dpp_max<0xB1>(v);        // quad_perm [1,0,3,2]
    if (n >= 4) v = dpp_max<0x4E>(v);        // quad_perm [2,3,0,1]
    if (n >= 8) v = dpp_max<0x141>(v);       // row_half_mirror: the other quad of the 8-lane half
    if (n >= 16) v = dpp_max<0x140>(v);      // row_mirror: the other half of the 16-lane row
    if (n >= 32) v = shfl_max(v, 16);
    if (n >= 64) v = shfl_max(v, 32);
    return v;
}

struct FusedArgs {
    const void* in;
    void* out_deq;
    void* out_codes;
    int8_t* out_exp;
    int64_t n_items;          // numel / VEC
    const uint8_t* exp_win;   // global, BFPQ_EXP_WIN_ENTRIES
    const uint8_t* nm_lut;    // global, BFPQ_NM4_LUT_ENTRIES (NM == 4)
    uint64_t seed;
    float eps_dt;
    int lpb;                  // lanes per block (power of two <= 64); 0 = no quantization
    int mant_bits;
    int N;
    int code_bits;
    int force_slow;           // mant_bits wider than the dtype significand: always emulate step by step
    const bfpq_select_state* sel;   // NM == -1 (global magnitude threshold): select result,
    const uint32_t* tie_counts;     //   ties per wave-chunk (k_tie_count),
    const int64_t* tie_base;        //   ties held by lower ranks (nullable)
    unsigned long long* unit_status;  // NM == -1, one-pass mode: flag + tie count per unit of kThreads x 8 items (zeroed); else null
    int* unit_error;                  //   set to 1 if a look-back spin ran into its cap (never in a healthy run)
};

// wave-level inclusive scan (lane order)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)v, o, 64);
        if (lane >= o) v += up;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o, 64);
    return v;
}

// Tie ranks without block barriers and without a fixed traversal order: the tensor is cut into "wave
// tiles" of 64 lane items (64 x 16 B, exactly what one wavefront handles per step).  k_tie_count writes
// the number of threshold ties of every tile, k_tie_chunk_sum / k_tie_scan turn that array into exclusive prefixes (one
// workgroup; the array is L2-sized), and a consumer wave reads prefix[tile] -- unconditionally, once per
// tile, so that no load sits inside a branch of the streaming loop -- and adds a wave scan of its own
// lanes' tie counts when the tile holds a tie at all.
__host__ __device__ inline int64_t tie_tiles(int64_t n_items) { return (n_items + 63) / 64; }
// workspace layout (uint32): [0, cpad) exclusive prefix over chunks of 64 tiles; [cpad, cpad + 64*chunks) ties per tile
struct TieLayout { int64_t n_tiles, n_chunks, cpad; };
__host__ __device__ inline TieLayout tie_layout(int64_t n_items)
{
    TieLayout l;
    l.n_tiles = tie_tiles(n_items);
    l.n_chunks = (l.n_tiles + 63) / 64;
    l.cpad = (l.n_chunks + 63) / 64 * 64;
    if (l.cpad < 64) l.cpad = 64;
    return l;
}

// threshold state as wave-uniform scalars
struct ThrCtx {
    uint32_t tau; unsigned long long need, base; bool on, ranked, allties;
    const uint32_t* coarse; const uint32_t* counts;
    __device__ __forceinline__ void load(const bfpq_select_state* st, const uint32_t* tie_ws, const int64_t* tie_base, int64_t n_items)
    {
        tau = st->tau; need = (unsigned long long)st->need;
        on = st->k > 0; ranked = on && st->need != 0 && st->need != st->ties; allties = st->need == st->ties;
        coarse = tie_ws; counts = tie_ws + tie_layout(n_items).cpad;
        base = tie_base ? (unsigned long long)*tie_base : 0ull;
    }
};

// The two unconditional loads a consumer wave makes per tile (nothing is loaded inside a branch of the
// streaming loop): ties before the tile's chunk of 64 tiles, and this lane's entry of the chunk's counts.
struct TileTies { uint32_t chunk_prefix, tile_prefix; };
__device__ __forceinline__ TileTies tile_ties(int64_t item, const ThrCtx& t)
{
    const int64_t tile = item >> 6;                        // wave-uniform: both loads are broadcasts
    TileTies r;
    r.chunk_prefix = t.coarse[tile >> 6];
    r.tile_prefix = t.counts[tile];
    return r;
}

// prune bits of one lane item (bit j = element j goes)
template <int DT>
__device__ __forceinline__ uint32_t thr_prune_bits(const uint32_t* raw, bool valid, int64_t item, const TileTies tt, const ThrCtx& t)
{
    constexpr int VEC = Traits<DT>::VEC;
    uint32_t ltm = 0, eqm = 0;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        const uint32_t key = mag_key<DT>(raw[j]);
        ltm |= (uint32_t)(key < t.tau) << j;
        eqm |= (uint32_t)(key == t.tau) << j;
    }
    if (!valid) eqm = 0;
    if (!t.on) return 0;
    uint32_t prune = ltm;
    if (t.ranked) {
        if (__ballot(eqm != 0)) {                          // most wave tiles hold no element equal to tau
            const uint32_t cnt = __popc(eqm);
            const uint32_t incl = wave_incl_scan(cnt);
            unsigned long long r = t.base + tt.chunk_prefix + tt.tile_prefix + (incl - cnt);
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                if ((eqm >> j) & 1u) { if (r < t.need) prune |= 1u << j; r++; }
            }
        }
    } else if (t.allties) prune |= eqm;
    return prune;
}

// scale of a block on the branch-free path; ok == false -> the caller emulates step by step instead
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

// ---------------------------------------------------------------------------------------------
// k_fused_flat: the tensor is a flat array of 16-byte lane items; rows do not matter because
// cols % block == 0 (and cols % M == 0).  NM in {0,2,4}.  One HBM read, one HBM write per output.
//   hot path per item: [packed 3-way comparisons -> 729-entry LDS table -> AND masks], packed abs/max,
//   DPP group max, exponent from a 320-byte LDS table, mul / rndne / med3 / mul per element, pack.
//   Anything unusual in a block (non-finite or zero max, scale outside the normal range, mantissa
//   wider than the dtype) makes the whole wavefront replay that item through the step-by-step
//   emulation (quant_elem); the branch is wave-uniform and never taken on ordinary weights.
// ---------------------------------------------------------------------------------------------
template <int DT, int NM, bool SFIRST, bool STOCH, int LPBT, bool DEQ_ONLY>
__global__ void __launch_bounds__(kThreads) k_fused_flat(const FusedArgs a)
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
    bool unit_mode = false;
    bool item_valid = true;
    int64_t item_index = 0;
    if constexpr (NM == -1) thr.load(a.sel, a.tie_counts, a.tie_base, a.n_items);
    auto nm_mask = [&](uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3) __attribute__((always_inline)) {
        if constexpr (NM == -1) {                       // global magnitude threshold (unstructured, bfp_ops.py:61-71)
            uint32_t raw[VEC];
            if constexpr (VEC == 4) { raw[0] = d0; raw[1] = d1; raw[2] = d2; raw[3] = d3; }
            else {
                raw[0] = d0 & 0xffffu; raw[1] = d0 >> 16; raw[2] = d1 & 0xffffu; raw[3] = d1 >> 16;
                raw[4] = d2 & 0xffffu; raw[5] = d2 >> 16; raw[6] = d3 & 0xffffu; raw[7] = d3 >> 16;
            }
            const TileTies tt = unit_mode ? TileTies{0u, 0u} : tile_ties(item_index, thr);   // unit mode: thr.base carries the rank
            const uint32_t prune = thr_prune_bits<DT>(raw, item_valid, item_index, tt, thr);
            if constexpr (VEC == 4) {
                d0 = (prune & 1u) ? 0u : d0; d1 = (prune & 2u) ? 0u : d1; d2 = (prune & 4u) ? 0u : d2; d3 = (prune & 8u) ? 0u : d3;
            } else {
                auto m = [](uint32_t pr) { return ((pr & 1u) ? 0u : 0xffffu) | ((pr & 2u) ? 0u : 0xffff0000u); };
                d0 &= m(prune); d1 &= m(prune >> 2); d2 &= m(prune >> 4); d3 &= m(prune >> 6);
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
    auto body = [&](auto guard_tag, const int64_t item, const uint4 cur) __attribute__((always_inline)) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        const bool valid = !GUARD || item < a.n_items;
        item_valid = valid;
        item_index = item;
        uint32_t d0 = cur.x, d1 = cur.y, d2 = cur.z, d3 = cur.w;

#ifdef BFPQ_COPYONLY          /* A/B knob: same loop, loads and stores only (ceiling for this launch geometry) */
        if (valid && a.out_deq) stream_store(reinterpret_cast<uint4*>(a.out_deq) + item, make_uint4(d0, d1, d2, d3));
        return;
#endif
        if constexpr (NM != 0 && SFIRST) nm_mask(d0, d1, d2, d3);             // S before Q (bfp_ops.py:141-144)

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
        if constexpr (NM != 0 && !SFIRST) {                                   // Q before S (bfp_ops.py:146-149)
            nm_mask(o0, o1, o2, o3);
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
        if constexpr (DEQ_ONLY) {                                           // hot mode: exactly one store per item
            if (valid) stream_store(reinterpret_cast<uint4*>(a.out_deq) + item, make_uint4(o0, o1, o2, o3));
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

    // Sweep: item = sweep * stride + global thread id.  Loads run two sweeps ahead of the item being
    // processed (index clamped to the last item, never conditional).
    const int64_t last = a.n_items - 1;
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    auto fetch = [&](int64_t i) __attribute__((always_inline)) {
        return __builtin_nontemporal_load(reinterpret_cast<const u4v*>(src + (i < last ? i : last)));
    };
    auto u4 = [](const u4v v) __attribute__((always_inline)) { return make_uint4(v.x, v.y, v.z, v.w); };
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
        if (a.unit_status != nullptr && thr.ranked) {
            // ONE-PASS unstructured apply.  A workgroup takes units of kThreads x IPT consecutive lane items (32 KiB of
            // bf16), keeps a unit in registers, counts its threshold ties per wave tile, publishes the unit's count in a
            // packed (flag, value) word and obtains the number of ties in all earlier units by decoupled look-back over
            // the lower-numbered units (device-scope atomics on those words only -- no __threadfence(), which is an L2
            // write-back on this part); then prunes with exact flat-order ranks, quantizes and stores from the registers.
            // The tensor is read once instead of twice (no k_tie_count pass) and three launches disappear.
            // Progress: the launcher bounds the grid by the resident capacity, so every unit waited for belongs to a
            // workgroup that is running; spins are capped all the same (a.unit_error).
            constexpr int IPT = 8, NW = kThreads / 64;
            constexpr unsigned long long F_AGG = 1ull << 62, F_PRE = 2ull << 62, VMASK = (1ull << 62) - 1;
            __shared__ uint32_t s_cnt[IPT * NW];                                // ties per wave tile, order (i, wave)
            __shared__ uint4 s_items[IPT][kThreads];                            // the unit, parked per thread between count and apply
                                                                                // (in registers the 8 inlined bodies cost 160 VGPRs)
            __shared__ unsigned long long s_unit_base;
            unit_mode = true;
            const unsigned long long base0 = thr.base;
            const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const int64_t unit_items = (int64_t)kThreads * IPT;
            const int64_t n_units = (a.n_items + unit_items - 1) / unit_items;
            u4v d[IPT];
            {
                const int64_t f0 = (int64_t)blockIdx.x * unit_items + threadIdx.x;
#pragma unroll
                for (int i = 0; i < IPT; i++) d[i] = fetch(f0 + (int64_t)i * kThreads);
            }
            for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
                const int64_t first = u * unit_items + threadIdx.x;
#pragma unroll
                for (int i = 0; i < IPT; i++) {
                    const uint32_t dw[4] = {d[i].x, d[i].y, d[i].z, d[i].w};
                    uint32_t cnt = 0;
#pragma unroll
                    for (int j = 0; j < VEC; j++) {
                        const uint32_t r = VEC == 4 ? dw[j] : ((dw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
                        cnt += mag_key<DT>(r) == thr.tau;
                    }
                    if (first + (int64_t)i * kThreads >= a.n_items) cnt = 0;
                    for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, 64);
                    if (lane == 0) s_cnt[i * NW + w] = cnt;
                    s_items[i][threadIdx.x] = u4(d[i]);
                }
                __syncthreads();
                {                                                   // the next unit's loads fly during the look-back and the apply
                    const int64_t fn = (u + gridDim.x) * unit_items + threadIdx.x;      // (clamped by fetch past the end)
#pragma unroll
                    for (int i = 0; i < IPT; i++) d[i] = fetch(fn + (int64_t)i * kThreads);
                }
                if (w == 0) {
                    const uint32_t v = lane < IPT * NW ? s_cnt[lane] : 0u;
                    const uint32_t incl = wave_incl_scan(v);
                    const unsigned long long total = (unsigned long long)(uint32_t)__shfl((int)incl, IPT * NW - 1, 64);
                    if (lane == 0)
                        __hip_atomic_store(&a.unit_status[u], (u == 0 ? F_PRE : F_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long excl = 0;
                    if (u > 0) {
                        // One poll reads LB x 64 status words.  Wider polls were tried (LB = 4, 16: the whole grid at once) and
                        // measured SLOWER (cfg 4: 61.9 us with 1, 64.2 with 4, 68.3 with 16): the uncached device-scope loads
                        // cost more than the look-back steps they save.
                        constexpr int LB = 1;
                        int64_t look = u - 1;
                        int spins = 0;
                        bool done = false;
                        while (!done) {
                            unsigned long long st[LB];
#pragma unroll
                            for (int j = 0; j < LB; j++) {
                                const int64_t idx = look - (int64_t)j * 64 - lane;
                                st[j] = idx >= 0 ? __hip_atomic_load(&a.unit_status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                 : F_PRE;                               // in front of unit 0: prefix 0
                            }
                            unsigned long long part = 0;
                            bool wait = false;
#pragma unroll
                            for (int j = 0; j < LB; j++) {
                                if (done || wait) continue;                             // (wave-uniform flags)
                                const uint32_t flag = (uint32_t)(st[j] >> 62);
                                const unsigned long long m_pre = __ballot(flag == 2u), m_empty = __ballot(flag == 0u);
                                const int fp = m_pre ? __ffsll((long long)m_pre) - 1 : 64;  // nearest predecessor with a prefix
                                const unsigned long long nearer = fp >= 64 ? ~0ull : ((1ull << fp) - 1ull);
                                if (m_empty & nearer) { wait = true; continue; }        // a nearer unit has not published yet
                                part += wave_sum64(lane <= fp ? (st[j] & VMASK) : 0ull);
                                if (fp < 64) done = true;
                            }
                            if (wait) {                                                 // poll the same window again
                                if (++spins > (1 << 20)) { if (lane == 0 && a.unit_error) *a.unit_error = 1; break; }
                                __builtin_amdgcn_s_sleep(1);
                                continue;
                            }
                            excl += part;
                            look -= (int64_t)LB * 64;
                        }
                        if (lane == 0)
                            __hip_atomic_store(&a.unit_status[u], F_PRE | ((excl + total) & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane < IPT * NW) s_cnt[lane] = incl - v;                        // ties of the unit's earlier wave tiles
                    if (lane == 0) s_unit_base = excl;
                }
                __syncthreads();
                const unsigned long long ubase = base0 + s_unit_base;
#pragma unroll 1
                for (int i = 0; i < IPT; i++) {
                    thr.base = ubase + s_cnt[i * NW + w];
                    body(std::true_type{}, first + (int64_t)i * kThreads, s_items[i][threadIdx.x]);
                }
                __syncthreads();                                                        // s_cnt is rewritten by the next unit
            }
            return;
        }
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

// ---------------------------------------------------------------------------------------------
// Unstructured: radix select on magnitude keys.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline void select_digit(int dtype, int pass, int* shift, int* nbits)
{
    if (dtype == BFPQ_F32) {
        if (pass == 0) { *shift = 20; *nbits = 11; }
        else if (pass == 1) { *shift = 9; *nbits = 11; }
        else { *shift = 0; *nbits = 9; }
    } else { *shift = 0; *nbits = 15; }
}

template <int DT>
__global__ void __launch_bounds__(1024) k_select_hist(const void* in, int64_t numel, int shift, int nbits, int first,
                                                      const bfpq_select_state* st, uint32_t* hist,
                                                      unsigned long long* zero_ptr, int64_t zero_n)
{
    // (the one-pass apply that follows needs its unit status words zeroed: done here, for free, instead of a memset node)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += (int64_t)gridDim.x * blockDim.x) zero_ptr[i] = 0ull;
    using T = Traits<DT>;
    using raw_t = typename T::raw_t;
    constexpr int VEC = T::VEC;
    extern __shared__ uint32_t s_hist[];
    const int nbins = 1 << nbits;
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const uint32_t pmask = first ? 0u : st->prefix_mask, pval = first ? 0u : st->prefix;   // pass 0 reads no state
    const uint32_t dmask = (uint32_t)nbins - 1u;
    const raw_t* src = reinterpret_cast<const raw_t*>(in);
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
    const int64_t n_items = aligned ? numel / VEC : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (n_items > 0) {
        // one-ahead prefetch with clamped, unconditional loads (a lone load per wave is latency-bound)
        const int64_t last = n_items - 1;
        int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        uint4 v = reinterpret_cast<const uint4*>(in)[item < last ? item : last];
        for (; item < n_items; item += stride) {
            const int64_t pf = item + stride;
            const uint4 nv = reinterpret_cast<const uint4*>(in)[pf < last ? pf : last];
            if constexpr (VEC == 8) {
                // 16-bit dtypes: one pass over the whole 15-bit key (shift 0, no prefix to match): two keys per packed
                // and/min, unconditional LDS atomics
                const uint32_t absm = T::ABS | (T::ABS << 16), nanc = (T::INF + 1u) | ((T::INF + 1u) << 16);
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t k2 = pk_min_i16_s(d[j] & absm, nanc);
                    atomicAdd(&s_hist[k2 & 0xffffu], 1u);
                    atomicAdd(&s_hist[k2 >> 16], 1u);
                }
            } else {
                const uint32_t raw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t key = mag_key<DT>(raw[j]);
                    if ((key & pmask) == pval) atomicAdd(&s_hist[(key >> shift) & dmask], 1u);
                }
            }
            v = nv;
        }
    }
    for (int64_t i = n_items * VEC + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
        const uint32_t key = mag_key<DT>((uint32_t)src[i]);
        if ((key & pmask) == pval) atomicAdd(&s_hist[(key >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

// One block of 1024 threads (16 waves): first bin whose inclusive prefix count reaches k_rem.
// Wave w owns a contiguous segment of nbins/16 bins, lane l the PER = seg/64 contiguous bins
// [l*PER, (l+1)*PER) of it -- all of a lane's loads are issued together (one memory latency), the
// rest is register arithmetic plus one wave scan in the wave that holds the crossing.
// Leaves the histogram zeroed for the next pass / call.
template <int PER>
__global__ void __launch_bounds__(1024) k_select_scan(bfpq_select_state* st, uint32_t* hist, int shift, int nbits, int last, int first, int64_t k_first)
{
    __shared__ unsigned long long s_tot[16];
    const int nbins = 1 << nbits;
    const int seg = nbins / 16;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool active = lane * PER < seg;                 // seg < 64 (512-bin digit): upper lanes idle
    uint32_t v[PER];
    if constexpr (PER == 32) {
        // A lane owns 32 contiguous bins (128 B); loading them directly makes every load instruction touch 64 different
        // 128-byte lines and the 8 loads of a wave touch the same 64 lines again -- with 16 waves that is 4x the L1, so
        // the lines come from L2 up to 8 times (measured 8.8 us for 128 KiB).  Instead the wave reads its 8 KiB segment
        // with coalesced 16-byte loads and transposes it through LDS (row stride 36 dwords: 16-byte aligned rows).
        extern __shared__ uint32_t s_seg[];                     // [16 waves][64 lanes][36]
        uint32_t* mine_seg = s_seg + w * (64 * 36);
        const uint4* src4 = reinterpret_cast<const uint4*>(hist + w * seg);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int b4 = j * 64 + lane;                          // 16-byte item inside the segment: bins 4 b4 .. 4 b4 + 3
            const uint4 q = src4[b4];
            *reinterpret_cast<uint4*>(mine_seg + (b4 >> 3) * 36 + (b4 & 7) * 4) = q;   // owner lane b4 / 8, offset (b4 % 8) * 4
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 32; j += 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(mine_seg + lane * 36 + j);
            v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
        }
    } else if constexpr (PER % 4 == 0) {
#pragma unroll
        for (int j = 0; j < PER; j += 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(hist + w * seg + lane * PER + j);
            v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; j++) v[j] = active ? hist[w * seg + lane * PER + j] : 0u;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) mine += v[j];
    const unsigned long long wtot = wave_sum64(mine);
    if (lane == 0) s_tot[w] = wtot;
    __syncthreads();
    // the first pass starts the selection: k comes as an argument, the state is (re)initialised below
    const unsigned long long k_rem = first ? (unsigned long long)k_first : (unsigned long long)st->k_rem;
    unsigned long long before = 0;
    int W = 15;                                          // k_rem beyond the total cannot happen (k <= numel)
    for (int i = 0; i < 16; i++) {
        if (k_rem <= before + s_tot[i]) { W = i; break; }
        if (i < 15) before += s_tot[i];
    }
    if (w == W) {
        const uint32_t incl = wave_incl_scan(mine);       // counts fit 32 bits per wave segment? no: use 64-bit compare below
        const unsigned long long lane_before = before + (unsigned long long)(incl - mine);
        const unsigned long long m = __ballot(active && lane_before + mine >= k_rem);
        int l = m ? __ffsll((long long)m) - 1 : 63;
        if (k_rem == 0) l = 0;
        if (lane == l) {
            unsigned long long run = lane_before;
            int bin = W * seg + lane * PER + PER - 1;
            uint32_t tie = v[PER - 1];
            bool found = false;
#pragma unroll
            for (int j = 0; j < PER; j++) {
                if (!found) {
                    if (k_rem <= run + v[j]) { bin = W * seg + lane * PER + j; tie = v[j]; found = true; }
                    else run += v[j];
                }
            }
            if (k_rem == 0) { bin = 0; run = 0; }
            if (first) {
                st->prefix = 0; st->prefix_mask = 0; st->tau = 0; st->done = 0; st->need = 0; st->ties = 0;
                st->k = k_first; st->reserved[0] = 0; st->reserved[1] = 0;
            }
            st->prefix |= (uint32_t)bin << shift;
            st->prefix_mask |= ((uint32_t)nbins - 1u) << shift;
            st->k_rem = (int64_t)(k_rem - run);
            if (last) {
                st->tau = st->prefix;
                st->need = st->k_rem;
                st->ties = (int64_t)tie;
                st->done = 1;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins / 4; i += 1024) reinterpret_cast<uint4*>(hist)[i] = make_uint4(0, 0, 0, 0);   // nbins % 4 == 0
}

template <int DT> __device__ __forceinline__ void load_raw_vec(const void* in, int64_t item, uint32_t* raw)
{
    constexpr int VEC = Traits<DT>::VEC;
    const uint4 v = reinterpret_cast<const uint4*>(in)[item];
    if constexpr (VEC == 4) { raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w; }
    else {
        raw[0] = v.x & 0xffffu; raw[1] = v.x >> 16; raw[2] = v.y & 0xffffu; raw[3] = v.y >> 16;
        raw[4] = v.z & 0xffffu; raw[5] = v.z >> 16; raw[6] = v.w & 0xffffu; raw[7] = v.w >> 16;
    }
}

// item-based grid-stride sweep with a one-ahead prefetch.  FAST: pointer 16-B aligned and numel a
// multiple of the vector width -> unconditional vector loads (index clamped), so the prefetch stays in
// flight; otherwise element loads with bounds checks.
template <int DT, bool FAST>
__device__ __forceinline__ void sweep_load(const void* in, int64_t item, int64_t n_items, int64_t numel, uint32_t* raw)
{
    using raw_t = typename Traits<DT>::raw_t;
    constexpr int VEC = Traits<DT>::VEC;
    if constexpr (FAST) load_raw_vec<DT>(in, item < n_items ? item : n_items - 1, raw);
    else {
        const int64_t e0 = item * VEC;
#pragma unroll
        for (int j = 0; j < VEC; j++)                          // past-the-end elements: a key that is never < or == tau
            raw[j] = (e0 + j < numel) ? (uint32_t)reinterpret_cast<const raw_t*>(in)[e0 + j] : (Traits<DT>::INF + 2u);
    }
}

// ties per wave tile (+ their sum per chunk of 64 tiles, integer atomics: deterministic); skipped on the
// device when ranks are not needed
template <int DT, bool FAST>
__global__ void __launch_bounds__(kThreads) k_tie_count(const void* in, int64_t numel, const bfpq_select_state* st, uint32_t* tie_ws)
{
    constexpr int VEC = Traits<DT>::VEC;
    const int64_t n_items = (numel + VEC - 1) / VEC;
    ThrCtx t; t.load(st, tie_ws, nullptr, n_items);
    if (!t.ranked) return;
    uint32_t* counts = tie_ws + tie_layout(n_items).cpad;
    const int64_t n_round = (n_items + 63) / 64 * 64;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    int64_t item = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    uint32_t cur[VEC], nxt[VEC];
    sweep_load<DT, FAST>(in, item, n_items, numel, cur);
    for (; item < n_round; item += stride) {                   // wave-uniform trip count
        sweep_load<DT, FAST>(in, item + stride, n_items, numel, nxt);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < VEC; j++) cnt += mag_key<DT>(cur[j]) == t.tau;
        if (item >= n_items) cnt = 0;
        for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, 64);
        if ((threadIdx.x & 63) == 0) counts[item >> 6] = cnt;
#pragma unroll
        for (int j = 0; j < VEC; j++) cur[j] = nxt[j];
    }
}

// sum of the 64 tile counts of every chunk: one wave per chunk, coalesced (no memory op inside a branch of
// k_tie_count's streaming loop: a conditional atomic there tripled its run time)
__global__ void __launch_bounds__(kThreads) k_tie_chunk_sum(const bfpq_select_state* st, uint32_t* tie_ws, int64_t n_tiles, int64_t n_chunks, int64_t cpad)
{
    const bool ranked = st->k > 0 && st->need != 0 && st->need != st->ties;
    if (!ranked) return;
    const int64_t c = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
    if (c >= n_chunks) return;
    const int lane = threadIdx.x & 63;
    const int64_t tile = c * 64 + lane;
    const uint32_t v = tile < n_tiles ? tie_ws[cpad + tile] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    // the tile's count is replaced by the ties of the chunk's earlier tiles: a consumer wave then needs two broadcast
    // dwords per tile (chunk prefix + this) instead of the chunk's 64 counts and a masked wave sum
    if (tile < n_tiles) tie_ws[cpad + tile] = incl - v;
    if (lane == 63) tie_ws[c] = incl;
}

// exclusive prefix of the per-chunk sums, in place (one workgroup, n = tiles / 64 entries);
// total -> st->reserved[0]
__global__ void __launch_bounds__(1024) k_tie_scan(bfpq_select_state* st, uint32_t* coarse, int64_t n)
{
    __shared__ unsigned long long s_w[16];
    const bool ranked = st->k > 0 && st->need != 0 && st->need != st->ties;
    if (!ranked) { if (threadIdx.x == 0) st->reserved[0] = 0; return; }
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    unsigned long long sum = 0;
    for (int64_t i = lo; i < hi; i++) sum += coarse[i];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = (unsigned long long)__shfl_up((long long)incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    unsigned long long woff = 0, total = 0;
    for (int i = 0; i < 16; i++) { if (i < w) woff += s_w[i]; total += s_w[i]; }
    unsigned long long run = woff + incl - sum;
    for (int64_t i = lo; i < hi; i++) { const uint32_t c = coarse[i]; coarse[i] = (uint32_t)run; run += c; }
    if (threadIdx.x == 0) st->reserved[0] = (int64_t)total;
}

template <int DT, bool FAST>
__global__ void __launch_bounds__(kThreads) k_threshold_apply(const void* in, void* out, int64_t numel,
                                                              const bfpq_select_state* st, const uint32_t* tie_ws,
                                                              const int64_t* tie_base)
{
    using raw_t = typename Traits<DT>::raw_t;
    constexpr int VEC = Traits<DT>::VEC;
    const int64_t n_items = (numel + VEC - 1) / VEC;
    ThrCtx t; t.load(st, tie_ws, tie_base, n_items);
    const int64_t n_round = (n_items + 63) / 64 * 64;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    int64_t item = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    uint32_t cur[VEC], nxt[VEC];
    sweep_load<DT, FAST>(in, item, n_items, numel, cur);
    for (; item < n_round; item += stride) {
        sweep_load<DT, FAST>(in, item + stride, n_items, numel, nxt);
        const bool valid = item < n_items;
        const TileTies tt = tile_ties(item, t);
        const uint32_t prune = thr_prune_bits<DT>(cur, valid, item, tt, t);
        if (valid) {
            const int64_t e0 = item * VEC;
            if constexpr (FAST) {
                uint32_t r[VEC];
#pragma unroll
                for (int j = 0; j < VEC; j++) r[j] = ((prune >> j) & 1u) ? 0u : cur[j];
                uint4 o;
                if constexpr (VEC == 4) o = make_uint4(r[0], r[1], r[2], r[3]);
                else o = make_uint4(r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16));
                reinterpret_cast<uint4*>(out)[item] = o;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; j++)
                    if (e0 + j < numel) reinterpret_cast<raw_t*>(out)[e0 + j] = ((prune >> j) & 1u) ? (raw_t)0 : (raw_t)cur[j];
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; j++) cur[j] = nxt[j];
    }
}

// ---------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------
inline float h_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t h_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

float h_round_bf16(float f)
{
    uint32_t u = h_f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f;
    u += 0x7fffu + ((u >> 16) & 1u);
    return h_u2f(u & 0xffff0000u);
}

// fp32 -> nearest fp16 (ties to even) -> fp32, via exact double arithmetic on the fp16 grid
float h_round_f16(float f)
{
    if (f != f || f == 0.0f) return f;
    const double a = fabs((double)f);
    if (a >= 65520.0) return f < 0 ? -INFINITY : INFINITY;
    int ex;
    frexp(a, &ex);                               // a = m * 2^ex, m in [0.5, 1)
    int q = ex - 11;                             // 11 significant bits
    if (q < -24) q = -24;                        // subnormal grid
    const double r = nearbyint(ldexp(a, -q));    // default rounding mode: ties to even
    const double v = ldexp(r, q);
    return (float)(f < 0 ? -v : v);
}

float h_round(float f, int dtype) { return dtype == BFPQ_F32 ? f : (dtype == BFPQ_F16 ? h_round_f16(f) : h_round_bf16(f)); }

int dtype_vec(int dtype) { return dtype == BFPQ_F32 ? 4 : 8; }
int dtype_size(int dtype) { return dtype == BFPQ_F32 ? 4 : 2; }
bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// workgroups for `work_threads` grid-stride work items: at most kMaxGrid, and balanced -- every
// workgroup gets the same number of sweeps (22016 blocks of work -> 18 sweeps x 1224 workgroups, not
// 1280 workgroups of which 256 do one sweep more)
int grid_for(int64_t work_threads)
{
    int64_t g = (work_threads + kThreads - 1) / kThreads;
    if (g < 1) g = 1;
    if (g <= kMaxGrid) return (int)g;
    const int64_t sweeps = (g + kMaxGrid - 1) / kMaxGrid;
    return (int)((g + sweeps - 1) / sweeps);
}

template <int DT, int NM, bool SFIRST, bool STOCH, bool DEQ_ONLY>
int launch_fused_o(const FusedArgs& a, hipStream_t s)
{
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

// resident capacity of the device for a kernel (workgroups), cached per kernel: the one-pass unstructured mode spins
// on lower-numbered units, so its grid must not exceed what can run at once
template <typename K>
int resident_workgroups(K kernel)
{
    static int cus = 0;                                           // per process: one device model per node
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 64;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    return per_cu * cus;
}

template <int DT>
int launch_fused_threshold(const FusedArgs& a, hipStream_t s)
{
    dim3 grid(grid_for(a.n_items)), block(kThreads);
    const bool deq_only = a.out_deq && !a.out_codes && !a.out_exp;
    if (a.unit_status) {                                         // one-pass mode: units of 2048 items, co-resident workgroups only
        const int64_t n_units = (a.n_items + 2047) / 2048;
        int cap;
        if (a.seed) cap = resident_workgroups(k_fused_flat<DT, -1, true, true, -1, false>);
        else if (deq_only && a.lpb == 8) cap = resident_workgroups(k_fused_flat<DT, -1, true, false, 8, true>);
        else if (deq_only && a.lpb == 4) cap = resident_workgroups(k_fused_flat<DT, -1, true, false, 4, true>);
        else if (deq_only) cap = resident_workgroups(k_fused_flat<DT, -1, true, false, -1, true>);
        else cap = resident_workgroups(k_fused_flat<DT, -1, true, false, -1, false>);
        int64_t g = n_units < cap ? n_units : cap;                     // (equal units per workgroup, i.e. fewer workgroups, measured slower)
        if (g > kMaxGrid) g = kMaxGrid;
        grid = dim3((unsigned)(g < 1 ? 1 : g));
    }
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

int launch_quant_rows(const void* in, void* out_deq, void* out_codes, int8_t* out_exp, int64_t rows, int64_t cols, int dtype,
                      int block, int mant_bits, float eps_dt, int code_bits, uint64_t seed, const uint8_t* exp_win, hipStream_t s)
{
    const int64_t total = rows * ((cols + block - 1) / block);
    if (total == 0) return 0;
    const int vec = dtype_vec(dtype);
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out_deq) | reinterpret_cast<uintptr_t>(out_codes)) & 15u) == 0;
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

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int bfpq_version(void) { return BFPQ_VERSION; }

int bfpq_tune(int key, int value)
{
    if (key == BFPQ_TUNE_MAX_GRID && value >= 1 && value <= 65535) { g_max_grid = value; return 0; }
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
        a.sel = nullptr; a.tie_counts = nullptr; a.tie_base = nullptr; a.unit_status = nullptr; a.unit_error = nullptr;
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
            a.sel = nullptr; a.tie_counts = nullptr; a.tie_base = nullptr; a.unit_status = nullptr; a.unit_error = nullptr;
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

int bfpq_nm_sparsify(const void* in, void* out, int64_t rows, int64_t cols, int dtype, int N, int M,
                     const uint8_t* nm4_lut, void* stream)
{
    if (!in || !out || dtype < 0 || dtype > 2 || rows < 0 || cols < 0) return BFPQ_E_ARG;
    if (!(N > 0 && M > 0 && N <= M)) return BFPQ_E_ARG;
    if (M > 64) return BFPQ_E_UNSUPPORTED;
    return bfpq_quantize_nm(in, out, nullptr, nullptr, rows, cols, dtype, 0, 0, 0.0, N, M, 1, 0, 0, nullptr, nm4_lut, nullptr, stream);
}

int bfpq_select_passes(int dtype) { return dtype == BFPQ_F32 ? 3 : 1; }

static int select_hist_impl(const void* in, int64_t numel, int dtype, int pass, const void* state, uint32_t* hist, void* stream,
                            unsigned long long* zero_ptr, int64_t zero_n);

int bfpq_select_hist(const void* in, int64_t numel, int dtype, int pass, const void* state, uint32_t* hist, void* stream)
{
    return select_hist_impl(in, numel, dtype, pass, state, hist, stream, nullptr, 0);
}

int bfpq_select_hist_prepare(const void* in, int64_t numel, int dtype, int pass, const void* state, uint32_t* hist,
                             uint32_t* tie_ws, void* stream)
{
    if (!tie_ws || dtype < 0 || dtype > 2 || numel < 0) return BFPQ_E_ARG;
    const int64_t n_items = numel / dtype_vec(dtype);
    return select_hist_impl(in, numel, dtype, pass, state, hist, stream, reinterpret_cast<unsigned long long*>(tie_ws), (n_items + 2047) / 2048);
}

static int select_hist_impl(const void* in, int64_t numel, int dtype, int pass, const void* state, uint32_t* hist, void* stream,
                            unsigned long long* zero_ptr, int64_t zero_n)
{
    if (!in || !state || !hist || dtype < 0 || dtype > 2 || numel < 0 || pass < 0 || pass >= bfpq_select_passes(dtype)) return BFPQ_E_ARG;
    int shift, nbits;
    select_digit(dtype, pass, &shift, &nbits);
    const size_t lds = sizeof(uint32_t) << nbits;
    const int threads = 1024;
    int64_t g = (numel / dtype_vec(dtype) + threads - 1) / threads;
    const int grid = (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
    hipStream_t s = (hipStream_t)stream;
    const bfpq_select_state* st = (const bfpq_select_state*)state;
    hipError_t err = hipSuccess;
    if (dtype == BFPQ_F32) {
        hipLaunchKernelGGL((k_select_hist<BFPQ_F32>), dim3(grid), dim3(threads), lds, s, in, numel, shift, nbits, pass == 0, st, hist, zero_ptr, zero_n);
    } else if (dtype == BFPQ_F16) {
        err = hipFuncSetAttribute((const void*)k_select_hist<BFPQ_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return (int)err;
        hipLaunchKernelGGL((k_select_hist<BFPQ_F16>), dim3(grid), dim3(threads), lds, s, in, numel, shift, nbits, pass == 0, st, hist, zero_ptr, zero_n);
    } else {
        err = hipFuncSetAttribute((const void*)k_select_hist<BFPQ_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return (int)err;
        hipLaunchKernelGGL((k_select_hist<BFPQ_BF16>), dim3(grid), dim3(threads), lds, s, in, numel, shift, nbits, pass == 0, st, hist, zero_ptr, zero_n);
    }
    return (int)hipGetLastError();
}

int bfpq_select_scan(int dtype, int pass, void* state, uint32_t* hist, int64_t k, void* stream)
{
    if (!state || !hist || dtype < 0 || dtype > 2 || pass < 0 || pass >= bfpq_select_passes(dtype) || k < 0) return BFPQ_E_ARG;
    int shift, nbits;
    select_digit(dtype, pass, &shift, &nbits);
    const int last = pass == bfpq_select_passes(dtype) - 1, first = pass == 0;
    bfpq_select_state* st = (bfpq_select_state*)state;
    hipStream_t s = (hipStream_t)stream;
    const int per = (1 << nbits) / 16 / 64;              // bins per lane: 32 (15-bit digit), 2 (11-bit), 0 -> 1 (9-bit)
    if (per == 32) {
        const size_t lds = 16 * 64 * 36 * sizeof(uint32_t);             // 144 KiB of the CU's 160
        const hipError_t err = hipFuncSetAttribute((const void*)k_select_scan<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return (int)err;
        hipLaunchKernelGGL(k_select_scan<32>, dim3(1), dim3(1024), lds, s, st, hist, shift, nbits, last, first, k);
    }
    else if (per == 2) hipLaunchKernelGGL(k_select_scan<2>, dim3(1), dim3(1024), 0, s, st, hist, shift, nbits, last, first, k);
    else hipLaunchKernelGGL(k_select_scan<1>, dim3(1), dim3(1024), 0, s, st, hist, shift, nbits, last, first, k);
    return (int)hipGetLastError();
}

int64_t bfpq_tie_workspace_elems(int64_t numel, int dtype)
{
    if (dtype < 0 || dtype > 2 || numel < 0) return BFPQ_E_ARG;
    const int vec = dtype_vec(dtype);
    const TieLayout l = tie_layout((numel + vec - 1) / vec);
    return l.cpad + l.n_chunks * 64 + 64;      // + slack: the ragged last sweep of a consumer reads up to 3 tiles past the end
}

int bfpq_tie_count(const void* in, int64_t numel, int dtype, void* state, uint32_t* tie_ws, void* stream)
{
    if (!in || !state || !tie_ws || dtype < 0 || dtype > 2 || numel < 0) return BFPQ_E_ARG;
    if (numel == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    bfpq_select_state* st = (bfpq_select_state*)state;
    const int vec = dtype_vec(dtype);
    const int64_t n_items = (numel + vec - 1) / vec;
    const bool fast = (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && numel % vec == 0;
    int64_t g = (n_items + kThreads - 1) / kThreads;
    const dim3 grid((unsigned)(g > 2048 ? 2048 : g)), block(kThreads);
    const TieLayout lay = tie_layout(n_items);
#define BFPQ_TC(DT) do { if (fast) hipLaunchKernelGGL((k_tie_count<DT, true>), grid, block, 0, s, in, numel, st, tie_ws); \
                         else hipLaunchKernelGGL((k_tie_count<DT, false>), grid, block, 0, s, in, numel, st, tie_ws); } while (0)
    if (dtype == BFPQ_F32) BFPQ_TC(BFPQ_F32); else if (dtype == BFPQ_F16) BFPQ_TC(BFPQ_F16); else BFPQ_TC(BFPQ_BF16);
#undef BFPQ_TC
    hipLaunchKernelGGL(k_tie_chunk_sum, dim3((unsigned)((lay.n_chunks + 3) / 4)), dim3(kThreads), 0, s, st, tie_ws, lay.n_tiles, lay.n_chunks, lay.cpad);
    hipLaunchKernelGGL(k_tie_scan, dim3(1), dim3(1024), 0, s, st, tie_ws, lay.n_chunks);
    return (int)hipGetLastError();
}

int bfpq_threshold_apply(const void* in, void* out, int64_t numel, int dtype, const void* state,
                         const uint32_t* tie_ws, const int64_t* tie_base, void* stream)
{
    if (!in || !out || !state || !tie_ws || dtype < 0 || dtype > 2 || numel < 0) return BFPQ_E_ARG;
    if (numel == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bfpq_select_state* st = (const bfpq_select_state*)state;
    const int vec = dtype_vec(dtype);
    const bool fast = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0 && numel % vec == 0;
    const dim3 grid(grid_for((numel + vec - 1) / vec)), block(kThreads);
#define BFPQ_TA(DT) do { if (fast) hipLaunchKernelGGL((k_threshold_apply<DT, true>), grid, block, 0, s, in, out, numel, st, tie_ws, tie_base); \
                         else hipLaunchKernelGGL((k_threshold_apply<DT, false>), grid, block, 0, s, in, out, numel, st, tie_ws, tie_base); } while (0)
    if (dtype == BFPQ_F32) BFPQ_TA(BFPQ_F32); else if (dtype == BFPQ_F16) BFPQ_TA(BFPQ_F16); else BFPQ_TA(BFPQ_BF16);
#undef BFPQ_TA
    return (int)hipGetLastError();
}

static int quantize_threshold_impl(const void* in, void* out_deq, void* out_codes, int8_t* out_exp,
                                   int64_t rows, int64_t cols, int dtype, int block_size, int mant_bits, double epsilon,
                                   int code_bits, uint64_t stoch_seed, const uint8_t* exp_win,
                                   const void* state, const uint32_t* counts, const int64_t* tie_base,
                                   void* scratch, void* stream, bool onepass, bool status_prepared = false);

int bfpq_quantize_threshold(const void* in, void* out_deq, void* out_codes, int8_t* out_exp,
                            int64_t rows, int64_t cols, int dtype, int block_size, int mant_bits, double epsilon,
                            int code_bits, uint64_t stoch_seed, const uint8_t* exp_win,
                            const void* state, const uint32_t* counts, const int64_t* tie_base,
                            void* scratch, void* stream)
{
    return quantize_threshold_impl(in, out_deq, out_codes, out_exp, rows, cols, dtype, block_size, mant_bits, epsilon, code_bits,
                                   stoch_seed, exp_win, state, counts, tie_base, scratch, stream, false);
}

int bfpq_quantize_threshold_onepass(const void* in, void* out_deq, void* out_codes, int8_t* out_exp,
                                    int64_t rows, int64_t cols, int dtype, int block_size, int mant_bits, double epsilon,
                                    int code_bits, uint64_t stoch_seed, const uint8_t* exp_win,
                                    void* state, uint32_t* tie_ws, int status_prepared, void* stream)
{
    if (dtype < 0 || dtype > 2 || rows < 0 || cols < 0 || block_size <= 0) return BFPQ_E_ARG;
    if (rows * cols && !fused_shape_ok(rows, cols, dtype, block_size, 0, 0)) return BFPQ_E_UNSUPPORTED;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out_deq) | reinterpret_cast<uintptr_t>(out_codes) |
                           reinterpret_cast<uintptr_t>(tie_ws)) & 15u) == 0;
    if (!aligned) return BFPQ_E_UNSUPPORTED;
    return quantize_threshold_impl(in, out_deq, out_codes, out_exp, rows, cols, dtype, block_size, mant_bits, epsilon, code_bits,
                                   stoch_seed, exp_win, state, tie_ws, nullptr, nullptr, stream, true, status_prepared != 0);
}

static int quantize_threshold_impl(const void* in, void* out_deq, void* out_codes, int8_t* out_exp,
                                   int64_t rows, int64_t cols, int dtype, int block_size, int mant_bits, double epsilon,
                                   int code_bits, uint64_t stoch_seed, const uint8_t* exp_win,
                                   const void* state, const uint32_t* counts, const int64_t* tie_base,
                                   void* scratch, void* stream, bool onepass, bool status_prepared)
{
    hipStream_t s = (hipStream_t)stream;
    if (dtype < 0 || dtype > 2 || rows < 0 || cols < 0 || block_size <= 0) return BFPQ_E_ARG;
    if (rows * cols == 0) return 0;
    if (!in || !state || !counts || !exp_win || (!out_deq && !out_codes && !out_exp)) return BFPQ_E_ARG;
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
        a.sel = (const bfpq_select_state*)state; a.tie_counts = counts; a.tie_base = tie_base;
        a.unit_status = nullptr; a.unit_error = nullptr;
        if (onepass) {
            // tie_ws doubles as the unit status array (8 B per unit of 2048 items <= the 4 B per 64 items it was sized for)
            // and, in the unranked case, as a harmless target of the multi-pass kernel's tie reads
            const int64_t n_units = (a.n_items + 2047) / 2048;
            bfpq_select_state* st = (bfpq_select_state*)const_cast<void*>(state);
            a.unit_status = reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(counts));
            a.unit_error = reinterpret_cast<int*>(&st->reserved[1]);
            if (!status_prepared) {
                const hipError_t me = hipMemsetAsync(a.unit_status, 0, sizeof(unsigned long long) * (size_t)n_units, s);
                if (me != hipSuccess) return (int)me;
            }
        }
        if (dtype == BFPQ_F32) return launch_fused_threshold<BFPQ_F32>(a, s);
        if (dtype == BFPQ_F16) return launch_fused_threshold<BFPQ_F16>(a, s);
        return launch_fused_threshold<BFPQ_BF16>(a, s);
    }
    void* tmp = out_deq ? out_deq : scratch;
    if (!tmp) return BFPQ_E_ARG;
    if (out_codes && code_bits == 4 && (block_size & 1)) return BFPQ_E_UNSUPPORTED;
    int rc = bfpq_threshold_apply(in, tmp, rows * cols, dtype, state, counts, tie_base, stream);
    if (rc) return rc;
    return launch_quant_rows(tmp, out_deq, out_codes, out_exp, rows, cols, dtype, block_size, mant_bits, eps_dt, code_bits, stoch_seed, exp_win, s);
}

}  // extern "C"
