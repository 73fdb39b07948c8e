// bfpq_int.hip -- the 'int' per-channel format of the reference's _quantize (bfp_ops.py:111-120):
// int_ops.Quantizer with configure() defaults (perchannel, symmetric; int_ops.py:18-31),
// find_params (:33-115) and quantize (:6-8).  Result is fp32 whatever the input dtype (the reference
// takes min/max against an fp32 zero tensor, which promotes everything downstream).
//
// The tensor is viewed as [outer, C, inner], the channel in the middle:
//   weight:            outer = 1,    C = shape[0],  inner = rest           -> k_int_rows  (one pass kernel:
//                                                                             a wave or a workgroup owns a channel,
//                                                                             reduces min/max, then quantizes it;
//                                                                             the second read of the row is L2-served)
//   2-D/3-D activation outer = rows, C = last dim,  inner = 1              -> k_int_cols_minmax_wide (+ atomics on
//                                                                             order-preserving integer keys:
//                                                                             deterministic) / k_int_cols_quant_flat
//                                                                             (C % 4 != 0 or unaligned: k_int_cols_minmax /
//                                                                             k_int_cols_quant; no flat grid: k_int_cols_quant_vec)
//   4-D activation     outer = N,    C = shape[1],  inner = H*W            -> k_int_seg_minmax / k_int_seg_quant
// All fp32 arithmetic is single IEEE operations in the reference's order (x / scale is a true division),
// compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "bfpq.h"
#include "bfpq_common.h"

using namespace bfpq;

namespace {

constexpr int kT = 256;

// order-preserving map float bits -> uint32 (so integer atomicMin/Max order floats; NaNs sort outermost)
__device__ __forceinline__ uint32_t f_key(float f) { const uint32_t u = f2u(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key_f(uint32_t k) { return u2f((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// find_params for one channel (int_ops.py:54-67): returns scale; min/max are the raw channel extrema
__device__ __forceinline__ float int_scale(float xmin, float xmax, float maxq)
{
    xmin = t_min(xmin, 0.0f);                              // torch.minimum(x.min(1)[0], zeros)   :55
    xmax = t_max(xmax, 0.0f);                              // torch.maximum(x.max(1)[0], zeros)   :56
    xmax = t_max(fabsf(xmin), xmax);                       // sym                                  :59
    if (xmin < 0.0f) xmin = -xmax;                         //                                      :60-62
    if (xmin == 0.0f && xmax == 0.0f) { xmin = -1.0f; xmax = 1.0f; }   //                           :63-65
    return (xmax - xmin) / maxq;                           //                                      :67
}

__device__ __forceinline__ float int_q(float x, float scale, float zero, float maxq)
{
    float q = rintf(x / scale) + zero;                     // round(x / scale) + zero              :7
    q = t_min(t_max(q, 0.0f), maxq);                       // clamp(., 0, maxq)
    return scale * (q - zero);                             //                                      :8
}

// Four elements at once without the division where that is provably the same: t = x * (1 / scale) lies within 3 ulp-halves
// (< 2^-22 |t|) of the correctly rounded quotient x / scale, so rint(t) == rint(x / scale) unless t is that close to a
// half-integer; a lane whose element is (within 2^-21 |t|, or not finite: NaN / inf land there too) makes its WAVE redo the
// unit with the true division.  The channel's extreme element sits exactly on +-(maxq / 2) and always does.  Everything behind
// the rint is the reference's own sequence (clamp as one v_med3: the operands are finite here).  Used by the per-row kernel (75 % busy
// issuing vector instructions with the division, profiles/r03d_pmc_int.md): [4096,11008] bf16 53.5 -> 52.4 us, [4096,4096] f32 26.8 -> 25.2 us
// interleaved; the per-column quantize launch measured SLOWER with it and keeps the division.
__device__ __forceinline__ void int_q4(const float* v, const float* scale, const float* rcp, float zero, float maxq, float* r)
{
    bool risky = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float t = v[j] * rcp[j];
        const float n = rintf(t);
        const float away = 0.5f - fabsf(t - n);                // distance to the nearest half-integer (t - n is exact)
        risky |= !(away > fabsf(t) * 4.76837158e-7f);           // 2^-21 |t|
        r[j] = scale[j] * (__builtin_amdgcn_fmed3f(n + zero, 0.0f, maxq) - zero);
    }
    if (__any(risky)) {
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = int_q(v[j], scale[j], zero, maxq);
    }
}

template <int DT> __device__ __forceinline__ float ldf(const void* in, int64_t i)
{
    using raw_t = typename Traits<DT>::raw_t;
    return raw_to_f32<DT>((uint32_t)reinterpret_cast<const raw_t*>(in)[i]);
}

__device__ __forceinline__ void wave_minmax(float& mn, float& mx, bool& nan)
{
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    nan = __any(nan);
}

// weight path: G threads (64 = one wave, or the whole 256-thread workgroup) own one channel of `inner`
// contiguous elements
template <int DT, int G>
__global__ void __launch_bounds__(kT) k_int_rows(const void* in, float* out, int64_t C, int64_t inner, float maxq, float zero)
{
    __shared__ float s_mn[kT / 64], s_mx[kT / 64];
    __shared__ int s_nan[kT / 64];
    constexpr int PER_WG = kT / G;
    const int sub = threadIdx.x / G, lig = threadIdx.x % G;
    for (int64_t c0 = (int64_t)blockIdx.x * PER_WG; c0 < C; c0 += (int64_t)gridDim.x * PER_WG) {
        const int64_t c = c0 + sub;
        const bool live = c < C;
        const int64_t base = (live ? c : 0) * inner;
        float mn = 0.0f, mx = 0.0f;
        bool nan = false;
        if (live)
            for (int64_t i = lig; i < inner; i += G) {
                const float v = ldf<DT>(in, base + i);
                nan |= v != v;
                mn = fminf(mn, v); mx = fmaxf(mx, v);          // fminf/fmaxf skip NaN; tracked separately
            }
        wave_minmax(mn, mx, nan);
        if constexpr (G > 64) {
            const int w = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) { s_mn[w] = mn; s_mx[w] = mx; s_nan[w] = nan; }
            __syncthreads();
            for (int i = 0; i < kT / 64; i++) { mn = fminf(mn, s_mn[i]); mx = fmaxf(mx, s_mx[i]); nan |= s_nan[i] != 0; }
            __syncthreads();
        }
        if (nan) { mn = u2f(0x7fc00000u); mx = mn; }           // torch min/max propagate NaN
        const float scale = int_scale(mn, mx, maxq);
        if (live)
            for (int64_t i = lig; i < inner; i += G) out[base + i] = int_q(ldf<DT>(in, base + i), scale, zero, maxq);
    }
}

__global__ void __launch_bounds__(kT) k_int_fill(uint32_t* ws, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) ws[i] = 0xffffffffu;
}

// activation path (inner == 1): thread owns one column of a chunk of rows
template <int DT>
__global__ void __launch_bounds__(kT) k_int_cols_minmax(const void* in, int64_t outer, int64_t C, int64_t rows_per_chunk, uint32_t* ws)
{
    const int64_t col = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (col >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < outer ? r0 + rows_per_chunk : outer;
    float mn = 0.0f, mx = 0.0f;
    bool nan = false;
    for (int64_t r = r0; r < r1; r++) {
        const float v = ldf<DT>(in, r * C + col);
        nan |= v != v;
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    if (nan) { mn = u2f(0xffc00000u); mx = u2f(0x7fc00000u); }     // keys: -NaN smallest, +NaN largest
    atomicMin(&ws[col], f_key(mn));
    atomicMin(&ws[C + col], ~f_key(mx));
}

template <int DT>
__global__ void __launch_bounds__(kT) k_int_cols_quant(const void* in, float* out, int64_t outer, int64_t C, int64_t rows_per_chunk,
                                                       const uint32_t* ws, float maxq, float zero)
{
    const int64_t col = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (col >= C) return;
    const float scale = int_scale(key_f(ws[col]), key_f(~ws[C + col]), maxq);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < outer ? r0 + rows_per_chunk : outer;
    for (int64_t r = r0; r < r1; r++) out[r * C + col] = int_q(ldf<DT>(in, r * C + col), scale, zero, maxq);
}

// activation path, vector flavour (C % VEC == 0, 16-B aligned): thread owns VEC adjacent columns of a row chunk
template <int DT>
__device__ __forceinline__ void load_vec_f(const void* in, int64_t item, float* v)
{
    constexpr int VEC = Traits<DT>::VEC;
    const uint4 q = reinterpret_cast<const uint4*>(in)[item];
    if constexpr (VEC == 4) { v[0] = u2f(q.x); v[1] = u2f(q.y); v[2] = u2f(q.z); v[3] = u2f(q.w); }
    else {
        const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; j++) { v[2 * j] = raw_to_f32<DT>(d[j] & 0xffffu); v[2 * j + 1] = raw_to_f32<DT>(d[j] >> 16); }
    }
}

template <int DT>
__global__ void __launch_bounds__(kT) k_int_cols_quant_vec(const void* in, float* out, int64_t outer, int64_t C, int64_t rows_per_chunk,
                                                           const uint32_t* ws, float maxq, float zero)
{
    constexpr int VEC = Traits<DT>::VEC;
    constexpr int R = 16;                                  // rows per batch: all their loads are issued before anything waits
    const int64_t ipr = C / VEC;
    const int64_t cg = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (cg >= ipr) return;
    const int64_t r0 = blockIdx.y, rstep = gridDim.y;      // rows r0, r0 + rstep, ... (see the min/max launch); rstep * R >= outer
    const uint4* src = reinterpret_cast<const uint4*>(in);
    uint4 q[R];
#pragma unroll
    for (int k = 0; k < R; k++) { const int64_t r = r0 + k * rstep; q[k] = src[(r < outer ? r : r0) * ipr + cg]; }
    // the scales (two dependent L2 round trips and VEC divisions) are worked out while the first batch is in flight: a
    // workgroup lives for one or two batches, so a set-up in front of the loads left the memory pipe idle a third of the time
    float scale[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) scale[j] = int_scale(key_f(ws[cg * VEC + j]), key_f(~ws[C + cg * VEC + j]), maxq);
#pragma unroll
    for (int k = 0; k < R; k++) {                          // (rows_per_chunk <= R: one batch per thread)
        float v[VEC];
        if constexpr (VEC == 4) { v[0] = u2f(q[k].x); v[1] = u2f(q[k].y); v[2] = u2f(q[k].z); v[3] = u2f(q[k].w); }
        else {
            const uint32_t d[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
            for (int j = 0; j < 4; j++) { v[2 * j] = raw_to_f32<DT>(d[j] & 0xffffu); v[2 * j + 1] = raw_to_f32<DT>(d[j] >> 16); }
        }
#pragma unroll
        for (int j = 0; j < VEC; j++) v[j] = int_q(v[j], scale[j], zero, maxq);
        if (r0 + k * rstep < outer) {
            float4* o = reinterpret_cast<float4*>(out) + ((r0 + k * rstep) * ipr + cg) * (VEC / 4);
            o[0] = make_float4(v[0], v[1], v[2], v[3]);
            if constexpr (VEC == 8) o[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// Activation path, r03 form ([4096,4096] bf16: 39.7 -> 29.4 us through the C ABI, [16384,4096]: 142 -> 89 us = 56 % of 8 TB/s on 6 B/elem;
// tools_dev/ab_int.py).  What the pair above cost: the quantize launch was ONE generation of workgroups that all loaded, then all
// computed (a true division per element), then all stored -- three phases in a row -- and, the larger part, every store instruction
// of a wave wrote 16 bytes of every 32 (a lane's eight fp32 results leave in two instructions): launches of 70-120 us at random on
// [16384,4096] where the hole-free form below takes 60.
//   k_int_cols_minmax_wide: 1024-thread workgroups, each 64 column groups wide, over row tiles of 16 R rows: wave w reads rows
//     tile 16 R + k 16 + w, a workgroup walks tiles blockIdx.y, + gridDim.y, ... (about one workgroup per CU, the next tile's loads in
//     flight while the current one is reduced), the 16 waves meet in LDS once at the end (row stride 72 words: lanes write
//     consecutive words, the reducing threads read consecutive COLUMNS with a two-way conflict at most) and leave 2 x 64 VEC atomics
//     per workgroup (one wave-instruction = 64 consecutive columns).
//   k_int_cols_quant_flat: below.
template <int DT, int R>
__global__ void __launch_bounds__(1024) k_int_cols_minmax_wide(const void* in, int64_t outer, int64_t C, uint32_t* ws)
{
    constexpr int VEC = Traits<DT>::VEC;
    constexpr int W = 16, LS = 72;                           // waves per workgroup; LDS row stride in words
    const int64_t ipr = C / VEC;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t cg0 = (int64_t)blockIdx.x * 64 + lane;
    const int64_t cg = cg0 < ipr ? cg0 : ipr - 1;            // idle lanes repeat the last column group (their results are not used)
    const uint4* src = reinterpret_cast<const uint4*>(in);
    // row tiles of 16 R rows: this workgroup takes tiles blockIdx.y, blockIdx.y + gridDim.y, ...; the next tile's R loads are issued
    // before the current tile is reduced.  Rows past the end repeat the last row: no extremum changes.
    auto ld = [&](int64_t tile, int k) __attribute__((always_inline)) {
        const int64_t r = tile * (W * R) + (int64_t)k * W + w;
        return src[(r < outer ? r : outer - 1) * ipr + cg];
    };
    const int64_t n_tiles = (outer + W * R - 1) / (W * R);
    uint4 q[R];
#pragma unroll
    for (int k = 0; k < R; k++) q[k] = ld(blockIdx.y, k);
    float mn[VEC], mx[VEC];
    bool nan[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) { mn[j] = 0.0f; mx[j] = 0.0f; nan[j] = false; }
    for (int64_t t = blockIdx.y; t < n_tiles; t += gridDim.y) {
        uint4 nx[R];
#pragma unroll
        for (int k = 0; k < R; k++) nx[k] = ld(t + gridDim.y, k);
#pragma unroll
        for (int k = 0; k < R; k++) {
            float v[VEC];
            if constexpr (VEC == 4) { v[0] = u2f(q[k].x); v[1] = u2f(q[k].y); v[2] = u2f(q[k].z); v[3] = u2f(q[k].w); }
            else {
                const uint32_t d[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
                for (int j = 0; j < 4; j++) { v[2 * j] = raw_to_f32<DT>(d[j] & 0xffffu); v[2 * j + 1] = raw_to_f32<DT>(d[j] >> 16); }
            }
#pragma unroll
            for (int j = 0; j < VEC; j++) { nan[j] |= v[j] != v[j]; mn[j] = fminf(mn[j], v[j]); mx[j] = fmaxf(mx[j], v[j]); }
            q[k] = nx[k];
        }
    }
    __shared__ uint32_t s_k[W][VEC * LS];                    // [wave][j * LS + lane]; minima first, then the (inverted) maxima: 36 KB
    const int64_t col0 = (int64_t)blockIdx.x * 64 * VEC;
#pragma unroll
    for (int which = 0; which < 2; which++) {
        if (which) __syncthreads();
#pragma unroll
        for (int j = 0; j < VEC; j++) {
            if (nan[j]) { mn[j] = u2f(0xffc00000u); mx[j] = u2f(0x7fc00000u); }
            s_k[w][j * LS + lane] = which ? ~f_key(mx[j]) : f_key(mn[j]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 64 * VEC; c += 1024) {
            const int at = (c % VEC) * LS + c / VEC;
            uint32_t k = s_k[0][at];
#pragma unroll
            for (int p = 1; p < W; p++) { const uint32_t o = s_k[p][at]; k = k < o ? k : o; }
            if (col0 + c < C) atomicMin(&ws[(which ? C : 0) + col0 + c], k);
        }
    }
}

// The quantize launch as a FLAT stream: unit i = four adjacent elements, a workgroup's units are contiguous, the grid moves through
// input and output as one compact front.  The host makes the grid stride a multiple of the units per row, so a thread meets ONE
// column unit on every trip and keeps its four scales in registers; UNR units per trip, the next trip's loads issued before the current
// units are quantized and stored.  Measured and dropped: q = rint(x * (1/scale)) with an exact fallback (int_q4, which the
// per-row kernel uses) -- bit-identical and SLOWER here, before and after the stores lost their holes ([4096,4096] bf16 32.4 against 29.3 us,
// [16384,4096] 96.2 against 88.8), although the counters show the launch 64 % busy issuing vector instructions; write-through (sc1)
// 16-byte stores, to spare the launch the write-back of the dirty L2 lines it leaves (~4 us at its end, seen as the duration of
// the NEXT launch in a kernel trace) -- 47-55 us instead of 25: bulk write-through runs at the fabric's ~1.3 TB/s.
// A thread's unit is FOUR elements -- one 16-byte fp32 store, so that every store instruction of a wave writes 1 KB without holes (with
// eight elements per lane the two halves of every 32 bytes left in different instructions) -- read by one 8-byte (16-bit dtypes) or
// 16-byte (fp32) load.
template <int DT, int UNR>
__global__ void __launch_bounds__(kT) k_int_cols_quant_flat(const void* in, float* out, int64_t n_units, int64_t upr, int64_t C,
                                                            const uint32_t* ws, float maxq, float zero)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    using in_t = typename std::conditional<DT == BFPQ_F32, u4v, u2v>::type;
    const int64_t stride = (int64_t)gridDim.x * kT;            // host: stride % upr == 0
    const int64_t i0 = (int64_t)blockIdx.x * kT + threadIdx.x;
    const int64_t cu = i0 % upr;                               // this thread's column unit on every trip
    const in_t* src = reinterpret_cast<const in_t*>(in);
    auto ld = [&](int64_t i) __attribute__((always_inline)) {  // unconditional, clamped: a load inside a branch makes hipcc wait for it at once
        return __builtin_nontemporal_load(src + (i < n_units ? i : n_units - 1));
    };
    in_t a[UNR];
#pragma unroll
    for (int k = 0; k < UNR; k++) a[k] = ld(i0 + k * stride);
    float scale[4];
    {
        const uint4 x = *reinterpret_cast<const uint4*>(ws + cu * 4), y = *reinterpret_cast<const uint4*>(ws + C + cu * 4);   // (C % 4 == 0, ws 16-byte aligned: host)
        scale[0] = int_scale(key_f(x.x), key_f(~y.x), maxq); scale[1] = int_scale(key_f(x.y), key_f(~y.y), maxq);
        scale[2] = int_scale(key_f(x.z), key_f(~y.z), maxq); scale[3] = int_scale(key_f(x.w), key_f(~y.w), maxq);
    }
    f4v* o = reinterpret_cast<f4v*>(out);
    for (int64_t i = i0; i < n_units; i += UNR * stride) {
        in_t n[UNR];
#pragma unroll
        for (int k = 0; k < UNR; k++) n[k] = ld(i + (UNR + k) * stride);
#pragma unroll
        for (int k = 0; k < UNR; k++) {
            float v[4];
            if constexpr (DT == BFPQ_F32) { v[0] = u2f(a[k].x); v[1] = u2f(a[k].y); v[2] = u2f(a[k].z); v[3] = u2f(a[k].w); }
            else {
                v[0] = raw_to_f32<DT>(a[k].x & 0xffffu); v[1] = raw_to_f32<DT>(a[k].x >> 16);
                v[2] = raw_to_f32<DT>(a[k].y & 0xffffu); v[3] = raw_to_f32<DT>(a[k].y >> 16);
            }
            const f4v w = {int_q(v[0], scale[0], zero, maxq), int_q(v[1], scale[1], zero, maxq), int_q(v[2], scale[2], zero, maxq), int_q(v[3], scale[3], zero, maxq)};
            if (i + k * stride < n_units) __builtin_nontemporal_store(w, o + i + k * stride);
            a[k] = n[k];
        }
    }
}

// 4-D activation path: one wave per (outer, channel) segment of `inner` contiguous elements
template <int DT>
__global__ void __launch_bounds__(kT) k_int_seg_minmax(const void* in, int64_t outer, int64_t C, int64_t inner, uint32_t* ws)
{
    const int lane = threadIdx.x & 63;
    const int64_t nseg = outer * C;
    for (int64_t seg = ((int64_t)blockIdx.x * kT + threadIdx.x) >> 6; seg < nseg; seg += (int64_t)gridDim.x * (kT / 64)) {
        float mn = 0.0f, mx = 0.0f;
        bool nan = false;
        for (int64_t i = lane; i < inner; i += 64) {
            const float v = ldf<DT>(in, seg * inner + i);
            nan |= v != v;
            mn = fminf(mn, v); mx = fmaxf(mx, v);
        }
        wave_minmax(mn, mx, nan);
        if (nan) { mn = u2f(0xffc00000u); mx = u2f(0x7fc00000u); }
        if (lane == 0) { atomicMin(&ws[seg % C], f_key(mn)); atomicMin(&ws[C + seg % C], ~f_key(mx)); }
    }
}

template <int DT>
__global__ void __launch_bounds__(kT) k_int_seg_quant(const void* in, float* out, int64_t outer, int64_t C, int64_t inner,
                                                      const uint32_t* ws, float maxq, float zero)
{
    const int lane = threadIdx.x & 63;
    const int64_t nseg = outer * C;
    for (int64_t seg = ((int64_t)blockIdx.x * kT + threadIdx.x) >> 6; seg < nseg; seg += (int64_t)gridDim.x * (kT / 64)) {
        const int64_t c = seg % C;
        const float scale = int_scale(key_f(ws[c]), key_f(~ws[C + c]), maxq);
        for (int64_t i = lane; i < inner; i += 64) out[seg * inner + i] = int_q(ldf<DT>(in, seg * inner + i), scale, zero, maxq);
    }
}

// weight path, vector flavour: inner % VEC == 0 and 16-B aligned rows -> 16-B loads, 16-B fp32 stores
template <int DT, int G>
__global__ void __launch_bounds__(kT) k_int_rows_vec(const void* in, float* out, int64_t C, int64_t inner, float maxq, float zero)
{
    constexpr int VEC = Traits<DT>::VEC;
    __shared__ float s_mn[kT / 64], s_mx[kT / 64];
    __shared__ int s_nan[kT / 64];
    constexpr int PER_WG = kT / G;
    const int sub = threadIdx.x / G, lig = threadIdx.x % G;
    const int64_t ipr = inner / VEC;                               // lane items per row
    auto load8 = [&](int64_t item, float* v) __attribute__((always_inline)) {
        const uint4 q = reinterpret_cast<const uint4*>(in)[item];
        if constexpr (VEC == 4) { v[0] = u2f(q.x); v[1] = u2f(q.y); v[2] = u2f(q.z); v[3] = u2f(q.w); }
        else {
            const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; j++) { v[2 * j] = raw_to_f32<DT>(d[j] & 0xffffu); v[2 * j + 1] = raw_to_f32<DT>(d[j] >> 16); }
        }
    };
    for (int64_t c0 = (int64_t)blockIdx.x * PER_WG; c0 < C; c0 += (int64_t)gridDim.x * PER_WG) {
        const int64_t c = c0 + sub;
        const bool live = c < C;
        const int64_t base = (live ? c : 0) * ipr;
        float mn = 0.0f, mx = 0.0f;
        bool nan = false;
        if (live)
            for (int64_t i = lig; i < ipr; i += G) {
                float v[VEC];
                load8(base + i, v);
#pragma unroll
                for (int j = 0; j < VEC; j++) { nan |= v[j] != v[j]; mn = fminf(mn, v[j]); mx = fmaxf(mx, v[j]); }
            }
        wave_minmax(mn, mx, nan);
        if constexpr (G > 64) {
            const int w = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) { s_mn[w] = mn; s_mx[w] = mx; s_nan[w] = nan; }
            __syncthreads();
            for (int i = 0; i < kT / 64; i++) { mn = fminf(mn, s_mn[i]); mx = fmaxf(mx, s_mx[i]); nan |= s_nan[i] != 0; }
            __syncthreads();
        }
        if (nan) { mn = u2f(0x7fc00000u); mx = mn; }
        const float scale = int_scale(mn, mx, maxq);
        if (live)
            for (int64_t i = lig; i < ipr; i += G) {
                float v[VEC];
                load8(base + i, v);
#pragma unroll
                for (int j = 0; j < VEC; j++) v[j] = int_q(v[j], scale, zero, maxq);
                float4* o = reinterpret_cast<float4*>(out) + (base + i) * (VEC / 4);
                o[0] = make_float4(v[0], v[1], v[2], v[3]);
                if constexpr (VEC == 8) o[1] = make_float4(v[4], v[5], v[6], v[7]);
            }
    }
}

// weight path, register flavour: one workgroup per row, the whole row (<= ITEMS 16-byte items per thread) is loaded at
// once and stays in registers between the min/max reduction and the quantize+store -- one read of the input, every load
// of a row in flight together (the looped flavour above is one dependent load per iteration: latency-bound, 87 us on
// [4096,11008] bf16 against 49 us of traffic)
template <int DT, int ITEMS>
__global__ void __launch_bounds__(kT) k_int_rows_reg(const void* in, float* out, int64_t C, int64_t inner, float maxq, float zero)
{
    // r03: a thread's unit is FOUR elements (an 8-byte load for the 16-bit dtypes, one 16-byte fp32 store), ITEMS2 = 2 ITEMS of them per
    // row: every store instruction of a wave writes 1 KB without holes (eight elements per lane left the two halves of every 32 bytes
    // in different instructions: [4096,11008] bf16 54.3 us; this form: see profiles/r03d_suite.md)
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    using in_t = typename std::conditional<DT == BFPQ_F32, u4v, u2v>::type;
    constexpr int N = DT == BFPQ_F32 ? ITEMS : 2 * ITEMS;          // units per thread
    __shared__ float s_mn[2][kT / 64], s_mx[2][kT / 64];
    __shared__ int s_nan[2][kT / 64];
    const int64_t upr = inner / 4;                                 // units per row, <= N * kT
    const int w = threadIdx.x >> 6;
    int ph = 0;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x, ph ^= 1) {
        const in_t* row = reinterpret_cast<const in_t*>(in) + c * upr;
        in_t q[N];
#pragma unroll
        for (int k = 0; k < N; k++) {                              // clamped, unconditional: all N loads in flight
            const int64_t i = (int64_t)k * kT + threadIdx.x;
            q[k] = row[i < upr ? i : upr - 1];
        }
        float mn = 0.0f, mx = 0.0f;
        bool nan = false;
        float v[N][4];
#pragma unroll
        for (int k = 0; k < N; k++) {
            if constexpr (DT == BFPQ_F32) { v[k][0] = u2f(q[k].x); v[k][1] = u2f(q[k].y); v[k][2] = u2f(q[k].z); v[k][3] = u2f(q[k].w); }
            else {
                v[k][0] = raw_to_f32<DT>(q[k].x & 0xffffu); v[k][1] = raw_to_f32<DT>(q[k].x >> 16);
                v[k][2] = raw_to_f32<DT>(q[k].y & 0xffffu); v[k][3] = raw_to_f32<DT>(q[k].y >> 16);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { nan |= v[k][j] != v[k][j]; mn = fminf(mn, v[k][j]); mx = fmaxf(mx, v[k][j]); }
        }                                                          // (clamped duplicates of the last unit change no extremum)
        wave_minmax(mn, mx, nan);
        if ((threadIdx.x & 63) == 0) { s_mn[ph][w] = mn; s_mx[ph][w] = mx; s_nan[ph][w] = nan; }
        __syncthreads();                                           // double-buffered by row parity: one barrier per row
#pragma unroll
        for (int i = 0; i < kT / 64; i++) { mn = fminf(mn, s_mn[ph][i]); mx = fmaxf(mx, s_mx[ph][i]); nan |= s_nan[ph][i] != 0; }
        if (nan) { mn = u2f(0x7fc00000u); mx = mn; }
        const float scale = int_scale(mn, mx, maxq), rcp = 1.0f / scale;
        const float sc4[4] = {scale, scale, scale, scale}, rc4[4] = {rcp, rcp, rcp, rcp};
        f4v* dst = reinterpret_cast<f4v*>(out) + c * upr;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const int64_t i = (int64_t)k * kT + threadIdx.x;
            float r[4];
            int_q4(v[k], sc4, rc4, zero, maxq, r);             // (outside the branch: the wave votes)
            if (i < upr) {
                dst[i] = (f4v){r[0], r[1], r[2], r[3]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 2:4 compaction of 4-bit codes (optional part of the packed format): a group of 4 codes with at
// most 2 non-zeros becomes 2 nibbles + 2 x 2-bit positions = 12 bits instead of 16.  One thread per 4 code bytes
// (8 elements = 2 groups): 2 value bytes + 1 index byte, kept in two arrays so that both stay naturally aligned:
//   vals [n/4 * 2] bytes: group g -> byte g = v0 | v1 << 4   (codes at positions p0 < p1)
//   idx  [n/4]     bytes: groups 2t, 2t+1 -> p0a | p1a << 2 | p0b << 4 | p1b << 6
// A group with more than 2 non-zero codes cannot be represented: status[0] is set to 1 (checked by the caller).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t compact_group(uint32_t g16, uint32_t* bad)
{
    // g16: 4 nibbles, element i at bits 4i.  Returns v0 | v1 << 4 | p0 << 8 | p1 << 10.
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) nz |= (((g16 >> (4 * i)) & 0xfu) != 0u) << i;
    const int cnt = __builtin_popcount(nz);
    if (cnt > 2) *bad = 1u;
    uint32_t p0, p1;
    if (cnt >= 2) { p0 = (uint32_t)__builtin_ctz(nz); p1 = (uint32_t)__builtin_ctz(nz & (nz - 1u)); }
    else if (cnt == 1) { const uint32_t p = (uint32_t)__builtin_ctz(nz); const uint32_t o = p == 0u ? 1u : 0u; p0 = p < o ? p : o; p1 = p < o ? o : p; }
    else { p0 = 0u; p1 = 1u; }
    return ((g16 >> (4 * p0)) & 0xfu) | (((g16 >> (4 * p1)) & 0xfu) << 4) | (p0 << 8) | (p1 << 10);
}

__global__ void __launch_bounds__(kT) k_compact24(const uint32_t* __restrict__ codes, uint16_t* __restrict__ vals, uint8_t* __restrict__ idx,
                                                  int64_t n, int32_t* status)
{
    uint32_t bad = 0;
    for (int64_t t = (int64_t)blockIdx.x * kT + threadIdx.x; t < n; t += (int64_t)gridDim.x * kT) {
        const uint32_t w = codes[t];
        const uint32_t a = compact_group(w & 0xffffu, &bad), b = compact_group(w >> 16, &bad);
        vals[t] = (uint16_t)((a & 0xffu) | ((b & 0xffu) << 8));
        idx[t] = (uint8_t)((a >> 8) | ((b >> 8) << 4));
    }
    if (bad) atomicOr(reinterpret_cast<unsigned int*>(status), 1u);
}

__global__ void __launch_bounds__(kT) k_expand24(const uint16_t* __restrict__ vals, const uint8_t* __restrict__ idx, uint32_t* __restrict__ codes, int64_t n)
{
    for (int64_t t = (int64_t)blockIdx.x * kT + threadIdx.x; t < n; t += (int64_t)gridDim.x * kT) {
        const uint32_t v = vals[t], p = idx[t];
        uint32_t w = 0;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const uint32_t vb = (v >> (8 * g)) & 0xffu, pb = (p >> (4 * g)) & 0xfu;
            w |= (((vb & 0xfu) << (4 * (pb & 3u))) | ((vb >> 4) << (4 * (pb >> 2)))) << (16 * g);
        }
        codes[t] = w;
    }
}

template <int DT>
int run_int(const void* in, float* out, int64_t outer, int64_t C, int64_t inner, float maxq, float zero, uint32_t* ws, hipStream_t s)
{
    if (outer == 1 && inner % Traits<DT>::VEC == 0 && inner / Traits<DT>::VEC <= 8 * kT && inner / Traits<DT>::VEC >= kT / 2 &&
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0) {
        const int64_t ipr = inner / Traits<DT>::VEC;
        const dim3 grid((unsigned)(C > 16384 ? 16384 : C));
        if (ipr <= 2 * kT) hipLaunchKernelGGL((k_int_rows_reg<DT, 2>), grid, dim3(kT), 0, s, in, out, C, inner, maxq, zero);
        else if (ipr <= 4 * kT) hipLaunchKernelGGL((k_int_rows_reg<DT, 4>), grid, dim3(kT), 0, s, in, out, C, inner, maxq, zero);
        else if (ipr <= 6 * kT) hipLaunchKernelGGL((k_int_rows_reg<DT, 6>), grid, dim3(kT), 0, s, in, out, C, inner, maxq, zero);
        else hipLaunchKernelGGL((k_int_rows_reg<DT, 8>), grid, dim3(kT), 0, s, in, out, C, inner, maxq, zero);
        return (int)hipGetLastError();
    }
    if (outer == 1 && inner % Traits<DT>::VEC == 0 &&
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0) {
        if (inner <= 16384) {
            int64_t g = (C + 3) / 4;
            hipLaunchKernelGGL((k_int_rows_vec<DT, 64>), dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(kT), 0, s, in, out, C, inner, maxq, zero);
        } else {
            hipLaunchKernelGGL((k_int_rows_vec<DT, 256>), dim3((unsigned)(C > 8192 ? 8192 : C)), dim3(kT), 0, s, in, out, C, inner, maxq, zero);
        }
        return (int)hipGetLastError();
    }
    if (outer == 1) {
        if (inner <= 16384) {
            int64_t g = (C + 3) / 4;
            hipLaunchKernelGGL((k_int_rows<DT, 64>), dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(kT), 0, s, in, out, C, inner, maxq, zero);
        } else {
            hipLaunchKernelGGL((k_int_rows<DT, 256>), dim3((unsigned)(C > 4096 ? 4096 : C)), dim3(kT), 0, s, in, out, C, inner, maxq, zero);
        }
        return (int)hipGetLastError();
    }
    if (!ws) return BFPQ_E_ARG;
    // min keys start at the top; the max keys are kept INVERTED (atomicMin on ~key) so that one fill serves both arrays.
    // (A workspace the caller keeps all ones, refilled by the quantize launch's last workgroup -- a ticket per workgroup -- instead of
    // the fill launch was built and measured: no faster on large tensors, 14.6 against 10.8 us on [1576,1024] f32: a thousand
    // tickets on one address serialise at ~9 ns each, and the fill launch hides behind its predecessor's write-back.)
    hipLaunchKernelGGL(k_int_fill, dim3((unsigned)((2 * C + 4 * kT - 1) / (4 * kT) > 64 ? 64 : (2 * C + 4 * kT - 1) / (4 * kT))), dim3(kT), 0, s, ws, 2 * C);   // (hipMemsetAsync: a 4.9 us node)
    if (inner == 1 && C % Traits<DT>::VEC == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0) {
        const int64_t ipr = C / Traits<DT>::VEC;
        const int64_t gxw = (ipr + 63) / 64;
        {
            // row tiles of 16 R rows; ~TG workgroups (two of 1024 threads fit a CU), every workgroup the same number of tiles where that is possible
            const int64_t tg = 256;
            const bool big = gxw * ((outer + 63) / 64) >= 128;
            const int64_t n_tiles = big ? (outer + 63) / 64 : (outer + 31) / 32;
            int64_t gy = tg / gxw;
            if (gy < 1) gy = 1;
            if (gy > n_tiles) gy = n_tiles;
            gy = (n_tiles + (n_tiles + gy - 1) / gy - 1) / ((n_tiles + gy - 1) / gy);       // the fewest workgroups with that many trips
            if (gy > 65535) gy = 65535;
            if (big) hipLaunchKernelGGL((k_int_cols_minmax_wide<DT, 4>), dim3((unsigned)gxw, (unsigned)gy), dim3(1024), 0, s, in, outer, C, ws);
            else hipLaunchKernelGGL((k_int_cols_minmax_wide<DT, 2>), dim3((unsigned)gxw, (unsigned)gy), dim3(1024), 0, s, in, outer, C, ws);
        }
        const int64_t upr = C / 4;                                     // the flat launch counts in units of four elements
        int64_t a = upr, b = kT;
        while (b) { const int64_t t = a % b; a = b; b = t; }
        const int64_t m = upr / a;                                     // its grid: a multiple of upr / gcd(upr, 256)
        if (m <= 2048 && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0) {
            const int64_t n_units = outer * upr;
            // workgroups (tools_dev/ab_int.py): a column count that divides the grid stride as it stands ran fastest with two per CU ([4096,4096]
            // bf16: 29.4 us at 512, 31-32 at 768 / 1024); the others with many ([2048,11008]: 42.8 at 516, 38.2 at 2021; [2048,13824]: 51.2 / 45.8)
            const int64_t tgt = m <= 4 ? 512 : 2048;
            constexpr int UNR = DT == BFPQ_F32 ? 2 : 4;
            int64_t G = (tgt + m - 1) / m * m;
            const int64_t need = ((n_units + kT - 1) / kT + m - 1) / m * m;
            if (G > need) G = need;
            hipLaunchKernelGGL((k_int_cols_quant_flat<DT, UNR>), dim3((unsigned)G), dim3(kT), 0, s, in, out, n_units, upr, C, (const uint32_t*)ws, maxq, zero);
        } else {
            const int64_t gx = (ipr + kT - 1) / kT;
            const int64_t rpc_q = 16;                                  // one batch of 16 rows per thread
            const int64_t rows_per_launch = rpc_q * 65535;             // (grid.y limit: more than a million rows go in several launches)
            for (int64_t row0 = 0; row0 < outer; row0 += rows_per_launch) {
                const int64_t n = outer - row0 < rows_per_launch ? outer - row0 : rows_per_launch;
                hipLaunchKernelGGL((k_int_cols_quant_vec<DT>), dim3((unsigned)gx, (unsigned)((n + rpc_q - 1) / rpc_q)), dim3(kT), 0, s,
                                   (const void*)(reinterpret_cast<const char*>(in) + row0 * C * (int64_t)sizeof(typename Traits<DT>::raw_t)), out + row0 * C,
                                   n, C, rpc_q, (const uint32_t*)ws, maxq, zero);
            }
        }
    } else if (inner == 1) {
        int64_t chunks = (outer + 63) / 64;
        if (chunks > 1024) chunks = 1024;
        const int64_t rpc = (outer + chunks - 1) / chunks;
        const dim3 grid((unsigned)((C + kT - 1) / kT), (unsigned)((outer + rpc - 1) / rpc));
        hipLaunchKernelGGL((k_int_cols_minmax<DT>), grid, dim3(kT), 0, s, in, outer, C, rpc, ws);
        hipLaunchKernelGGL((k_int_cols_quant<DT>), grid, dim3(kT), 0, s, in, out, outer, C, rpc, (const uint32_t*)ws, maxq, zero);
    } else {
        int64_t g = (outer * C + 3) / 4;
        const dim3 grid((unsigned)(g > 4096 ? 4096 : g));
        hipLaunchKernelGGL((k_int_seg_minmax<DT>), grid, dim3(kT), 0, s, in, outer, C, inner, ws);
        hipLaunchKernelGGL((k_int_seg_quant<DT>), grid, dim3(kT), 0, s, in, out, outer, C, inner, (const uint32_t*)ws, maxq, zero);
    }
    return (int)hipGetLastError();
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// packed HBFP -> tensor: out = code * 2^(e - mant_bits) (exact in the dtype by construction; exponent -128
// marks a block the quantizer turned into NaN).  Lane item = VEC elements, as everywhere else.
// ---------------------------------------------------------------------------------------------
// 4-bit codes, regular shape (cols % block == 0, block % 8 == 0, aligned): one lane item = 8 elements = one dword of
// codes, one exponent per block/8 adjacent items, one 16-byte (16-bit dtypes) or two 16-byte (fp32) stores -- a streaming
// kernel (0.516 B in, sizeof out per element) instead of the element-at-a-time general decoder above
template <int DT>
__global__ void __launch_bounds__(kT) k_dequant4_vec(const uint32_t* __restrict__ codes, const int8_t* __restrict__ exps, void* __restrict__ out,
                                                     int64_t n_items, int ipb, int ipb_shift, int mant_bits)
{
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * kT) {
        const uint32_t w = __builtin_nontemporal_load(codes + i);
        const int e = exps[ipb_shift >= 0 ? (i >> ipb_shift) : (i / ipb)];      // (a 64-bit division per item otherwise)
        const float scale = ldexpf(1.0f, e - mant_bits);
        uint32_t o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int code = (int)(((w >> (4 * j)) & 0xfu) ^ 8u) - 8;
            const float val = e == -128 ? u2f(0x7fc00000u) : (float)code * scale;      // exact: |code| <= 7, scale a power of two
            o[j] = DT == BFPQ_F32 ? f2u(val) : f32_to_raw<DT>(val);
        }
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        if constexpr (DT == BFPQ_F32) {
            u4v* dst = reinterpret_cast<u4v*>(out) + 2 * i;
            __builtin_nontemporal_store((u4v){o[0], o[1], o[2], o[3]}, dst);
            __builtin_nontemporal_store((u4v){o[4], o[5], o[6], o[7]}, dst + 1);
        } else {
            __builtin_nontemporal_store((u4v){o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16)},
                                        reinterpret_cast<u4v*>(out) + i);
        }
    }
}

template <int DT>
__global__ void __launch_bounds__(kT) k_dequant(const void* codes, const int8_t* exps, void* out, int64_t rows, int64_t cols,
                                                int block, int mant_bits, int code_bits)
{
    using raw_t = typename Traits<DT>::raw_t;
    constexpr int VEC = Traits<DT>::VEC;
    const int64_t nblk = (cols + block - 1) / block;
    const int64_t crow = code_bits == 4 ? (cols + 1) / 2 : cols;          // code bytes (4-bit) or entries per row
    const int64_t ipr = (cols + VEC - 1) / VEC;                           // lane items per row (last one may be ragged)
    const int64_t total = rows * ipr;
    const bool vec_ok = cols % VEC == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    for (int64_t v = (int64_t)blockIdx.x * kT + threadIdx.x; v < total; v += (int64_t)gridDim.x * kT) {
        const int64_t row = v / ipr, it = v - row * ipr;
        uint32_t o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) {
            const int64_t c = it * VEC + j;
            int code = 0, e = 0;
            if (c < cols) {
                e = exps[row * nblk + c / block];
                if (code_bits == 4) {
                    const uint32_t b = reinterpret_cast<const uint8_t*>(codes)[row * crow + c / 2];
                    code = (int)((b >> (4 * (c & 1))) & 0xfu);
                    if (code > 7) code -= 16;
                } else if (code_bits == 8) code = reinterpret_cast<const int8_t*>(codes)[row * crow + c];
                else code = reinterpret_cast<const int16_t*>(codes)[row * crow + c];
            }
            const float val = e == -128 ? u2f(0x7fc00000u) : ldexpf((float)code, e - mant_bits);
            o[j] = f32_to_raw<DT>(val);
        }
        if (vec_ok) {
            uint4 w;
            if constexpr (VEC == 4) w = make_uint4(o[0], o[1], o[2], o[3]);
            else w = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
            reinterpret_cast<uint4*>(out)[row * ipr + it] = w;
        } else {
#pragma unroll
            for (int j = 0; j < VEC; j++)
                if (it * VEC + j < cols) reinterpret_cast<raw_t*>(out)[row * cols + it * VEC + j] = (raw_t)o[j];
        }
    }
}

extern "C" {

int bfpq_dequantize(const void* codes, const int8_t* exps, void* out, int64_t rows, int64_t cols, int dtype,
                    int block_size, int mant_bits, int code_bits, void* stream)
{
    if (rows < 0 || cols < 0 || dtype < 0 || dtype > 2 || block_size <= 0 || mant_bits < 0 || mant_bits > 15) return BFPQ_E_ARG;
    if (!(code_bits == 4 || code_bits == 8 || code_bits == 16)) return BFPQ_E_ARG;
    if (rows * cols == 0) return 0;
    if (!codes || !exps || !out) return BFPQ_E_ARG;
    if (code_bits == 4 && cols % block_size == 0 && block_size % 8 == 0 &&
        ((reinterpret_cast<uintptr_t>(codes) & 3u) | (reinterpret_cast<uintptr_t>(out) & 15u)) == 0) {
        const int64_t n_items = rows * cols / 8;
        const int ipb4 = block_size / 8;
        int ipb4_shift = -1;
        if ((ipb4 & (ipb4 - 1)) == 0) { ipb4_shift = 0; while ((1 << ipb4_shift) < ipb4) ipb4_shift++; }
        int64_t g4 = (n_items + kT - 1) / kT;
        const dim3 grid4((unsigned)(g4 > 2048 ? 2048 : g4));
        hipStream_t s4 = (hipStream_t)stream;
        if (dtype == BFPQ_F32) hipLaunchKernelGGL((k_dequant4_vec<BFPQ_F32>), grid4, dim3(kT), 0, s4, (const uint32_t*)codes, exps, out, n_items, ipb4, ipb4_shift, mant_bits);
        else if (dtype == BFPQ_F16) hipLaunchKernelGGL((k_dequant4_vec<BFPQ_F16>), grid4, dim3(kT), 0, s4, (const uint32_t*)codes, exps, out, n_items, ipb4, ipb4_shift, mant_bits);
        else hipLaunchKernelGGL((k_dequant4_vec<BFPQ_BF16>), grid4, dim3(kT), 0, s4, (const uint32_t*)codes, exps, out, n_items, ipb4, ipb4_shift, mant_bits);
        return (int)hipGetLastError();
    }
    const int vec = dtype == BFPQ_F32 ? 4 : 8;
    int64_t g = (rows * ((cols + vec - 1) / vec) + kT - 1) / kT;
    const dim3 grid((unsigned)(g > 2048 ? 2048 : g));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == BFPQ_F32) hipLaunchKernelGGL((k_dequant<BFPQ_F32>), grid, dim3(kT), 0, s, codes, exps, out, rows, cols, block_size, mant_bits, code_bits);
    else if (dtype == BFPQ_F16) hipLaunchKernelGGL((k_dequant<BFPQ_F16>), grid, dim3(kT), 0, s, codes, exps, out, rows, cols, block_size, mant_bits, code_bits);
    else hipLaunchKernelGGL((k_dequant<BFPQ_BF16>), grid, dim3(kT), 0, s, codes, exps, out, rows, cols, block_size, mant_bits, code_bits);
    return (int)hipGetLastError();
}

int bfpq_compact24(const void* codes, void* vals, void* idx, int64_t n_bytes, int32_t* status, void* stream)
{
    if (n_bytes < 0 || n_bytes % 4 != 0) return BFPQ_E_ARG;
    if (n_bytes == 0) return 0;
    if (!codes || !vals || !idx || !status) return BFPQ_E_ARG;
    const int64_t n = n_bytes / 4;
    int64_t g = (n + kT - 1) / kT;
    hipLaunchKernelGGL(k_compact24, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(kT), 0, (hipStream_t)stream,
                       (const uint32_t*)codes, (uint16_t*)vals, (uint8_t*)idx, n, status);
    return (int)hipGetLastError();
}

int bfpq_expand24(const void* vals, const void* idx, void* codes, int64_t n_bytes, void* stream)
{
    if (n_bytes < 0 || n_bytes % 4 != 0) return BFPQ_E_ARG;
    if (n_bytes == 0) return 0;
    if (!codes || !vals || !idx) return BFPQ_E_ARG;
    const int64_t n = n_bytes / 4;
    int64_t g = (n + kT - 1) / kT;
    hipLaunchKernelGGL(k_expand24, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(kT), 0, (hipStream_t)stream,
                       (const uint16_t*)vals, (const uint8_t*)idx, (uint32_t*)codes, n);
    return (int)hipGetLastError();
}

int64_t bfpq_int_workspace_elems(int64_t C) { return C < 0 ? BFPQ_E_ARG : 2 * C; }

int bfpq_int_quantize(const void* in, float* out, int64_t outer, int64_t C, int64_t inner, int dtype, int bits,
                      uint32_t* ws, void* stream)
{
    if (outer < 0 || C < 0 || inner < 0 || dtype < 0 || dtype > 2 || bits < 0 || bits > 30) return BFPQ_E_ARG;
    if (outer * C * inner == 0) return 0;
    if (!in || !out) return BFPQ_E_ARG;
    const float maxq = (float)((1u << bits) - 1u);
    const float zero = (maxq + 1.0f) / 2.0f;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == BFPQ_F32) return run_int<BFPQ_F32>(in, out, outer, C, inner, maxq, zero, ws, s);
    if (dtype == BFPQ_F16) return run_int<BFPQ_F16>(in, out, outer, C, inner, maxq, zero, ws, s);
    return run_int<BFPQ_BF16>(in, out, outer, C, inner, maxq, zero, ws, s);
}

}  // extern "C"
