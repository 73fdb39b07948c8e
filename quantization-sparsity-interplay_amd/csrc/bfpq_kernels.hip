// bfpq_kernels.hip -- gfx950 kernels + C-ABI launchers of libbfpq.so (see include/bfpq.h).
//
// Reference path replaced: src/transformers/bfp/bfp_ops.py:16-149.
//
// Kernels
//   (bfpq_fused.h, compiled per dtype by bfpq_fused_dt.hip:)
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
extern "C" __attribute__((visibility("hidden"))) int bfpq_g_mx8_variant;   // bfpq_mxgemm.hip

#include "bfpq_device.h"
#include "bfpq_quant_math.h"

extern "C" { __attribute__((visibility("hidden"))) int bfpq_g_max_grid = BFPQ_MAXGRID; }
static int bfpq_g_list_own_mb = 24;     // bfpq_tune(BFPQ_TUNE_LIST_OWN_MB): tensors of a list call from this many MB on get launches of their own


using namespace bfpq_dev;

namespace {
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
    if (key == BFPQ_TUNE_MX8_VARIANT && value >= -1 && value <= 6) { bfpq_g_mx8_variant = value; return 0; }
    if (key == BFPQ_TUNE_LIST_OWN_MB && value >= 0) { bfpq_g_list_own_mb = value; return 0; }
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
        return fused_launch(dtype, a, M, sparsify_first != 0, s);
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
            return fused_launch(dtype, a, 0, true, s);
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

int bfpq_quantize_mx8(const void* in, void* out8, void* out_scale, int64_t rows, int64_t cols, int dtype, int mant_bits, double epsilon,
                      const uint8_t* exp_win, void* stream)
{
    if (dtype < 0 || dtype > 2 || rows < 0 || cols < 0 || mant_bits < 1 || mant_bits > 4) return BFPQ_E_ARG;
    if (rows * cols == 0) return 0;
    if (!in || !out8 || !out_scale || !exp_win) return BFPQ_E_ARG;
    if (cols % 64 != 0) return BFPQ_E_UNSUPPORTED;                                     // (callers take bfpq_quantize_nm + bfpq_mx8_from_hbfp)
    if ((reinterpret_cast<uintptr_t>(in) & 15u) || (reinterpret_cast<uintptr_t>(out8) & 7u) || (reinterpret_cast<uintptr_t>(out_scale) & 7u)) return BFPQ_E_UNSUPPORTED;
    const float eps_dt = h_round((float)epsilon, dtype);
    FusedArgs a;
    a.in = in; a.out_deq = nullptr; a.out_codes = out8; a.out_exp = reinterpret_cast<int8_t*>(out_scale);
    a.n_items = rows * cols / dtype_vec(dtype);
    a.exp_win = exp_win; a.nm_lut = nullptr; a.seed = 0; a.eps_dt = eps_dt;
    a.lpb = 64 / dtype_vec(dtype);
    a.mant_bits = mant_bits; a.N = 0; a.code_bits = 8;
    a.force_slow = 0;
    set_hot16(a, dtype, mant_bits, eps_dt);
    a.selws = nullptr;
    if (dtype == BFPQ_F32) {                                                            // fp32: the general arithmetic of the fused kernel, image bytes in its store epilogue
        a.code_bits = kCodeBitsE4M3;
        return fused_launch(dtype, a, 0, true, (hipStream_t)stream);
    }
    return fused_mx8(dtype, a, (hipStream_t)stream);
}

int bfpq_fake_quantize(const bfpq_plan* p, const void* in, void* out, int64_t rows, int64_t cols, void* stream)
{
    if (!p) return BFPQ_E_ARG;
    return bfpq_quantize_nm(in, out, nullptr, nullptr, rows, cols, p->dtype, p->block_size, p->mant_bits, p->epsilon, p->N, p->M,
                            p->sparsify_first, 0, 0, p->exp_win_dev, p->nm_lut_dev, nullptr, stream);
}

// A tensor of at least this many input bytes gets a launch of its own inside a list call (k_fused_flat), on one of the call's lanes
// (the caller's stream and its aux streams); smaller ones share list launches (k_fused_batched).  Measured on [4096,11008] / [4096,4096] /
// [1024,4096] bf16 2:4 -> HBFP4, 64 tensors per pass (tools_dev/ab_footprint.py, us per tensor): list kernel 35.7-37.0 / 11.4-12.9 /
// 2.6-3.1; one launch per tensor on one stream 31.9-33.4 / 13.7 / 5.6-5.9; one launch per tensor over two streams 29.5-30.2 / 11.6 /
// 5.4-5.5.  The list kernel's workgroups drift apart over a long list (no launch boundary pulls the sweep's front together again)
// and it wins only where a launch of its own would be mostly ramp and tail; two streams let the tail of one tensor's launch run
// beside the ramp of the next one.
#define kListOwnLaunchBytes ((int64_t)bfpq_g_list_own_mb << 20)

// Lanes of a list call: the caller's stream (lane 0) and up to kMaxLanes - 1 aux streams.  Every tensor that gets launches of its
// own goes, whole, to the lane that has been given the fewest bytes so far; the lanes run side by side without any event between
// them -- one fork (the aux lanes wait for what the caller's stream held at the call; the event is recorded in front of the call's first
// launch of this kind) and one join (the caller's stream waits for every aux lane that was used).  hipGraph-capturable from the caller's stream.
constexpr int kMaxLanes = 8;
struct Lanes {
    hipStream_t s[kMaxLanes];
    int64_t bytes[kMaxLanes];
    bool used[kMaxLanes];
    int n;
    hipEvent_t fork;
    Lanes(void* stream, void* const* aux, int n_aux) : n(1), fork(nullptr)
    {
        s[0] = (hipStream_t)stream;
        for (int i = 0; aux && i < n_aux && n < kMaxLanes; i++) {
            hipStream_t a = (hipStream_t)aux[i];
            bool dup = !a;
            for (int j = 0; j < n; j++) dup = dup || s[j] == a;
            if (!dup) s[n++] = a;
        }
        for (int i = 0; i < kMaxLanes; i++) { bytes[i] = 0; used[i] = false; }
    }
    int pick(int64_t b)                                                    // lane for a tensor of b bytes
    {
        // the fork is recorded in front of the call's FIRST own launch, whichever lane takes it: recorded later, it would also make
        // the aux lanes wait for the tensors lane 0 has been given in the meantime
        if (n > 1 && !fork && (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess || hipEventRecord(fork, s[0]) != hipSuccess)) {
            if (fork) { (void)hipEventDestroy(fork); fork = nullptr; }
            n = 1;                                                         // (no event: everything on the caller's stream)
        }
        int l = 0;
        for (int i = 1; i < n; i++) if (bytes[i] < bytes[l]) l = i;
        if (l > 0 && !used[l]) {
            if (hipStreamWaitEvent(s[l], fork, 0) != hipSuccess) l = 0;
            else used[l] = true;
        }
        bytes[l] += b;
        return l;
    }
    int join(int rc)                                                       // also on an error path: a capture must not be left forked
    {
        for (int l = 1; l < n; l++) {
            if (!used[l]) continue;
            hipEvent_t e = nullptr;
            hipError_t h = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (h == hipSuccess) h = hipEventRecord(e, s[l]);
            if (h == hipSuccess) h = hipStreamWaitEvent(s[0], e, 0);
            if (e) (void)hipEventDestroy(e);
            if (h != hipSuccess && !rc) rc = (int)h;
        }
        if (fork) (void)hipEventDestroy(fork);
        fork = nullptr;
        return rc;
    }
};

int bfpq_fake_quantize_list(const bfpq_plan* p, const bfpq_tensor_desc* descs, int n, void* stream, void* const* aux_streams, int n_aux)
{
    if (!p || (n > 0 && !descs) || n < 0 || n_aux < 0 || (n_aux > 0 && !aux_streams)) return BFPQ_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    Lanes lanes(stream, aux_streams, n_aux);
    auto join = [&](int rc) { return lanes.join(rc); };
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
    // (the small tensors' list launches stay on the caller's stream: spread over the lanes in byte-budgeted batches they were SLOWER --
    // OPT-125m's 72 weights 147 us against 122-131, ViT-L's 144 weights 462 against 417-439: the fork and the join cost more than two
    // concurrent list launches gain)
    // lanes only pay when at least two tensors get launches of their own: for ONE large tensor next to small ones (a Linear's weight
    // next to its activation) the fork and the join cost more than running the two side by side gains (bench_linear.py, un-cached
    // BFPLinear forward, 16 tokens x down_proj: 75.2 us with the weight on an aux lane against 72.6 on one stream)
    if (lanes.n > 1) {
        int n_own = 0;
        for (int i = 0; i < n; i++)
            if (descs[i].rows > 0 && descs[i].cols > 0 && descs[i].rows * descs[i].cols * dtype_size(dtype) >= kListOwnLaunchBytes) n_own++;
        if (n_own < 2) lanes.n = 1;
    }
    auto flush = [&]() -> int {
        if (b.n == 0) return 0;
        const int M = batch_has_nm ? 4 : 0;
        int rc;
        rc = fused_batched(dtype, a, b, M, p->sparsify_first != 0, s);
        b.n = 0; b.total_chunks = 0; batch_has_nm = false;
        return rc;
    };
    for (int i = 0; i < n; i++) {
        const bfpq_tensor_desc& d = descs[i];
        if (d.rows < 0 || d.cols < 0) return join(BFPQ_E_ARG);
        if (d.rows * d.cols == 0) continue;
        if (!d.in_dev || !d.out_dev) return join(BFPQ_E_ARG);
        const bool nm = any_nm_cfg && d.apply_nm != 0;
        const int N = nm ? p->N : 0, M = nm ? p->M : 0;
        if (p->block_size == 0 && M == 0) return join(BFPQ_E_ARG);         // identity: the caller keeps its tensor
        const bool aligned = ((reinterpret_cast<uintptr_t>(d.in_dev) | reinterpret_cast<uintptr_t>(d.out_dev)) & 15u) == 0;
        const bool ok = aligned && (M == 0 || (M == 4 && p->nm_lut_dev)) && p->block_size > 0 &&
                        fused_shape_ok(d.rows, d.cols, dtype, p->block_size, N, M) &&
                        (d.rows * d.cols / dtype_vec(dtype)) % kThreads == 0 &&          // whole chunks of 256 lane items only
                        (d.rows * d.cols / dtype_vec(dtype) + kThreads - 1) / kThreads < ((int64_t)1 << 31);
        const bool own = d.rows * d.cols * dtype_size(dtype) >= kListOwnLaunchBytes;
        if (!ok || own) {
            const int rc = bfpq_quantize_nm(d.in_dev, d.out_dev, nullptr, nullptr, d.rows, d.cols, dtype, p->block_size, p->mant_bits, p->epsilon,
                                            N, M, p->sparsify_first, 0, 0, p->exp_win_dev, p->nm_lut_dev, nullptr,
                                            own ? (void*)lanes.s[lanes.pick(d.rows * d.cols * dtype_size(dtype))] : stream);
            if (rc) return join(rc);
            continue;
        }
        const int64_t items = d.rows * d.cols / dtype_vec(dtype);
        const int64_t chunks = (items + kThreads - 1) / kThreads;
        if (b.n == kMaxBatch || (int64_t)b.total_chunks + chunks >= ((int64_t)1 << 32)) { const int rc = flush(); if (rc) return join(rc); }
        BatchDesc& o = b.d[b.n++];
        o.in = d.in_dev; o.out = d.out_dev; o.n_items = items; o.chunk0 = b.total_chunks; o.flags = nm ? 1u : 0u;
        b.total_chunks += (uint32_t)chunks;
        batch_has_nm = batch_has_nm || nm;
    }
    return join(flush());
}

int bfpq_fake_quantize_batched(const bfpq_plan* p, const bfpq_tensor_desc* descs, int n, void* stream)
{
    return bfpq_fake_quantize_list(p, descs, n, stream, nullptr, 0);
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
        return fused_threshold(dtype, a, s);
    }
    void* tmp = out_deq ? out_deq : scratch;
    if (!tmp) return BFPQ_E_ARG;
    if (out_codes && code_bits == 4 && (block_size & 1)) return BFPQ_E_UNSUPPORTED;
    int rc = bfpq_threshold_apply(in, tmp, rows * cols, dtype, ws, stream);
    if (rc) return rc;
    return launch_quant_rows(tmp, out_deq, out_codes, out_exp, rows, cols, dtype, block_size, mant_bits, eps_dt, code_bits, stoch_seed, exp_win, s);
}

int bfpq_prune_quantize(const void* in, void* out, int64_t rows, int64_t cols, int dtype, int block_size, int mant_bits,
                        double epsilon, int64_t k, const uint8_t* exp_win, void* ws, void* stream)
{
    if (dtype < 0 || dtype > 2 || rows < 0 || cols < 0 || block_size <= 0 || k < 0 || k > rows * cols) return BFPQ_E_ARG;
    if (rows * cols == 0) return 0;
    if (!in || !out || !ws || !exp_win || in == out || mant_bits < 0 || mant_bits > 23) return BFPQ_E_ARG;
    if (rows * cols >= ((int64_t)1 << 32)) return BFPQ_E_UNSUPPORTED;
    const int rc = bfpq_select(in, rows * cols, dtype, k, ws, stream);
    if (rc) return rc;
    return bfpq_quantize_threshold(in, out, nullptr, nullptr, rows, cols, dtype, block_size, mant_bits, epsilon, 0, 0, exp_win, ws, nullptr, stream);
}

// Whole tensors over independent lanes: a tensor's two launches (selection, prune + quantize) go back to back to the lane that has
// been given the fewest bytes so far, every lane has its own workspace, and nothing orders the lanes against one another.  LLaMA-13B,
// all 280 weights, bf16 (tools_dev/ab_prune_list.py, hipGraph): one lane 16.8 ms, two 14.2, three 12.7, four 12.5 -- and 15.5 for the
// round-3 scheme this replaces (selection of tensor i + 1 on the caller's stream beside the apply launch of tensor i on ONE aux
// stream, ordered by two events per tensor): a lane's serial tail (the last workgroup's resolve step, the kernel boundary) is
// covered by the other lanes' streaming, and three or four lanes cover it better than one partner does.
int bfpq_prune_quantize_list(const bfpq_prune_desc* descs, int n, int dtype, int block_size, int mant_bits, double epsilon,
                             const uint8_t* exp_win, void* const* wss, int n_ws, void* stream, void* const* aux_streams, int n_aux)
{
    if (n < 0 || (n > 0 && !descs) || !wss || n_ws < 1 || n_aux < 0 || (n_aux > 0 && !aux_streams)) return BFPQ_E_ARG;
    for (int i = 0; i < n_ws; i++) if (!wss[i]) return BFPQ_E_ARG;
    for (int i = 0; i < n_ws; i++) for (int j = 0; j < i; j++) if (wss[i] == wss[j]) return BFPQ_E_ARG;   // (a workspace per lane)
    Lanes lanes(stream, aux_streams, n_aux < n_ws - 1 ? n_aux : n_ws - 1);
    int rc = 0;
    for (int i = 0; i < n && !rc; i++) {
        const bfpq_prune_desc& d = descs[i];
        if (d.rows < 0 || d.cols < 0 || d.k < 0 || d.k > d.rows * d.cols) { rc = BFPQ_E_ARG; break; }
        if (d.rows * d.cols == 0) continue;
        if (!d.in_dev || !d.out_dev || d.in_dev == d.out_dev) { rc = BFPQ_E_ARG; break; }
        const int l = n > 1 ? lanes.pick(d.rows * d.cols * dtype_size(dtype < 0 || dtype > 2 ? 0 : dtype)) : 0;
        rc = bfpq_prune_quantize(d.in_dev, d.out_dev, d.rows, d.cols, dtype, block_size, mant_bits, epsilon, d.k, exp_win, wss[l], lanes.s[l]);
    }
    return lanes.join(rc);
}

int bfpq_prune_quantize_batched(const bfpq_prune_desc* descs, int n, int dtype, int block_size, int mant_bits, double epsilon,
                                const uint8_t* exp_win, void* const* wss, int n_ws, void* stream, void* aux_stream)
{
    void* aux[1] = {aux_stream};
    return bfpq_prune_quantize_list(descs, n, dtype, block_size, mant_bits, epsilon, exp_win, wss, n_ws, stream, aux, aux_stream ? 1 : 0);
}

}  // extern "C"
