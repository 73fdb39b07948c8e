// bfpq_select.h -- device code shared by the launches that select the global magnitude threshold (bfpq_unstructured.hip: the
// histogram launch): publishing a segment's counts, the ticket, and the resolve step that the last workgroup runs.  Reference: src/transformers/bfp/bfp_ops.py:61-71.
#pragma once
#include <hip/hip_runtime.h>
#include "bfpq.h"
#include "bfpq_common.h"
#include "bfpq_device.h"

namespace bfpq_dev {

// Build with EXTRA=-DBFPQ_STAMPS for phase timing (tools_dev/stamps.py): thread 0 of every workgroup records the
// constant 100 MHz clock at named points; never in the product build.
#ifdef BFPQ_STAMPS
static __device__ unsigned long long bfpq_g_stamps[3][512][8];        // (one copy per translation unit, read back by that unit's bfpq_debug_stamps*)
#define STAMP(kern, idx) do { if (threadIdx.x == 0 && blockIdx.x < 512) { unsigned long long t_; \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); bfpq_g_stamps[kern][blockIdx.x][idx] = t_; } } while (0)
#else
#define STAMP(kern, idx) do { } while (0)
#endif

// words that one workgroup hands to another INSIDE a launch (segment windows -> the resolving workgroup) travel as
// agent-scope relaxed atomics: write-through stores, cache-bypassing loads, no fences (cdna_hip_programming.md, guideline 16 R1)
__device__ __forceinline__ void pub_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t pub_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// value of `v` in the lowest lane whose `cond` holds (0 when none does), for every lane
__device__ __forceinline__ uint32_t pick_lane(bool cond, uint32_t v)
{
    const unsigned long long m = __ballot(cond);
    if (!m) return 0u;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)__ffsll((long long)m) - 1));
}

// fine bin b of the segment's LDS histogram: 32-bit counters, or (U16) two 16-bit counters per word (bin 2i low half, 2i + 1 high)
template <bool U16> __device__ __forceinline__ uint32_t lds_bin(const uint32_t* s_hist, uint32_t b)
{
    if constexpr (U16) { const uint32_t w = s_hist[b >> 1]; return (b & 1u) ? w >> 16 : w & 0xffffu; }
    else return s_hist[b];
}

// coarse sums (NCB coarse bins of 128 fine bins) of the finished LDS histogram -> s_coarse; every thread sums 32 words with
// 16-byte LDS reads in a rotated order, then a reduction over the 1024 / NCB threads of a coarse bin.  1024 threads.
template <int NCB, bool U16>
__device__ __forceinline__ void coarse_from_lds(const uint32_t* s_hist, uint32_t* s_coarse)
{
    constexpr int TPC = kSelThreads / NCB;                    // threads per coarse bin: 4 (256 bins of 128 words) or 2 (512 bins of 64 words)
    constexpr int WPC = U16 ? 64 : 128;                       // words per coarse bin
    static_assert(TPC * 32 == WPC, "coarse_from_lds: 32 words per thread");
    const int t = threadIdx.x, q = t / TPC, r = t % TPC;
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint4 c4 = *reinterpret_cast<const uint4*>(&s_hist[q * WPC + ((j + q) & 7) * (TPC * 4) + r * 4]);
        if constexpr (U16) sum += (c4.x & 0xffffu) + (c4.x >> 16) + (c4.y & 0xffffu) + (c4.y >> 16) + (c4.z & 0xffffu) + (c4.z >> 16) + (c4.w & 0xffffu) + (c4.w >> 16);
        else sum += c4.x + c4.y + c4.z + c4.w;
    }
    sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0xB1, 0xf, 0xf, false);          // quad_perm [1,0,3,2]
    if constexpr (TPC == 4) sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    if (r == 0) s_coarse[q] = sum;
}

// After the streaming loop, a barrier and the coarse sums (s_coarse[NCB], 16-byte aligned, complete): publish the segment -- its
// coarse sums into one of the workspace's coarse-histogram copies, a 2048-bin window of the fine histogram around the segment's
// own quantile (target_frac of its elements), and the window's position -- then draw the ticket.  True for the LAST workgroup
// of the grid: everything every segment published is in memory when it returns.  fine(b): the segment's count of fine bin b.
// A SECOND window per segment (windows2 / seg_win2):
//   clo2 >= 0: at coarse bin clo2, a position all workgroups share (fp32's low digit);
//   clo2 <  0: where the segment's quantile would lie if its rank were off by what a segment of this size can be off by
//     (+- 4 sqrt(elements)), when that is outside the first window.  A distribution with a heavy ATOM next to the threshold -- a
//     tensor that is already half zeros pruned by half again -- has its quantile in the atom's bin in one segment and thousands
//     of bins higher in the next; with one window per segment half of them could not answer for the global threshold's bin and
//     were recounted by the last workgroup, one after the other (5.3 ms instead of 41 us on [5120,5120]).
// s_res: 4 words.  1024 threads; nobody waits for anybody here.
template <int NCB, class F>
__device__ __forceinline__ bool seg_publish_and_ticket(const uint32_t* s_coarse, uint32_t* s_res, SelWs* ws, double target_frac, F&& fine, int clo2 = -1)
{
    const int t = threadIdx.x;
    const int copy = blockIdx.x % BFPQ_SELECT_HIST_COPIES;
    constexpr int WC = kWinBins / 128;                           // coarse bins per window
    if (t < NCB) { const uint32_t c = s_coarse[t]; if (c) atomicAdd(&ws->coarse[copy][t], c); }
    STAMP(0, 3);
    // window: the 16 coarse bins (2048 bins) around the one that holds the segment's own quantile (first wave: NCB / 64 coarse
    // sums per lane, one wave scan)
    if (t < 64) {
        constexpr int PER = NCB / 64;
        uint32_t c[PER], mine = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { c[j] = s_coarse[t * PER + j]; mine += c[j]; }
        const uint32_t incl = wave_incl_scan(mine);
        const uint32_t seg_elems = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t excl = incl - mine;
        auto coarse_of = [&](uint64_t rank) __attribute__((always_inline)) {   // the coarse bin that holds the element of this rank (wave-uniform result)
            const bool hit = mine && excl <= rank && rank < (uint64_t)excl + mine;
            uint32_t e = excl;
            int A = t * PER;
#pragma unroll
            for (int j = 0; j < PER - 1; j++) { if (rank >= (uint64_t)e + c[j]) { e += c[j]; A = t * PER + j + 1; } else break; }
            return (int)pick_lane(hit, (uint32_t)A);
        };
        auto place = [](int A) { int lo = A - WC / 2; return lo < 0 ? 0 : (lo > NCB - WC ? NCB - WC : lo); };
        uint64_t target = (uint64_t)((double)seg_elems * target_frac);                          // (an anchor, not a count)
        if (seg_elems && target >= seg_elems) target = seg_elems - 1;
        const int clo = place(coarse_of(target));
        int clob = clo2;                                                                        // second window: given, or from the rank band
        if (clo2 < 0 && seg_elems) {
            const uint64_t eps = 4ull * (uint64_t)__builtin_sqrtf((float)seg_elems) + 4ull;
            const int A_lo = coarse_of(target > eps ? target - eps : 0ull), A_hi = coarse_of(target + eps < seg_elems ? target + eps : seg_elems - 1);
            if (A_lo < clo) clob = place(A_lo);
            else if (A_hi >= clo + WC) clob = place(A_hi);
        }
        // elements inside the first window, and inside the union of the two
        const uint32_t in_a = wave_sum(t < WC ? s_coarse[clo + t] : 0u);
        uint32_t in_b = 0;
        if (clob >= 0) in_b = wave_sum((t < WC && (uint32_t)(clob + t - clo) >= (uint32_t)WC) ? s_coarse[clob + t] : 0u);   // (bins of the second window outside the first)
        if (t == 0) {
            s_res[0] = seg_elems ? (uint32_t)clo * 128u : 0u;
            s_res[2] = (seg_elems && clob >= 0) ? (uint32_t)clob * 128u + 1u : 0u;                                         // (+ 1: "there is one")
            pub_store(&ws->seg_win[blockIdx.x], seg_elems ? (((uint32_t)clo * 128u) | (in_a != seg_elems ? 0x80000000u : 0u)) : 0u);   // (empty segment: no window, nothing outside it)
            pub_store(&ws->seg_win2[blockIdx.x], (seg_elems && clob >= 0) ? (((uint32_t)clob * 128u) | 0x40000000u | (in_a + in_b != seg_elems ? 0x80000000u : 0u)) : 0u);
        }
    }
    __syncthreads();
    STAMP(0, 4);
    const uint32_t lo = s_res[0], lo2p = s_res[2];
    pub_store(&ws->windows[blockIdx.x][t], fine(lo + (uint32_t)t));
    pub_store(&ws->windows[blockIdx.x][kSelThreads + t], fine(lo + (uint32_t)(kSelThreads + t)));
    if (lo2p) {                                              // (block-uniform) the second window
        const uint32_t lo2 = lo2p - 1u;
        pub_store(&ws->windows2[blockIdx.x][t], fine(lo2 + (uint32_t)t));
        pub_store(&ws->windows2[blockIdx.x][kSelThreads + t], fine(lo2 + (uint32_t)(kSelThreads + t)));
    }
    STAMP(0, 5);
    // publish: every storing wave drains, the workgroup meets, one lane draws the ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_res[1] = __hip_atomic_fetch_add(&ws->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    return s_res[1] != 0;
}

// ---------------------------------------------------------------------------------------------
// The resolve step inside the histogram launch (single device): run by the LAST workgroup to draw its ticket, i.e. when
// every segment's coarse counts and window are in memory.  1024 threads; lds: the (finished) histogram's LDS.
//   coarse bins (8 copies x NCB) -> scan -> the coarse bin C that holds the k_rem-th key of this digit
//   every segment's window slice for C (16-byte cache-bypassing buffer loads, eight per thread, all independent)
//     -> their sum is the global fine histogram of C -> scan -> the digit; on the last digit the column of tau is every
//        segment's tie count -> scan -> the cut segment
// The digit: keys with (key & pmask) == pval take part, their bin is (key >> shift) & (NCB * 128 - 1).  16-bit dtypes: one digit
// (shift 0, NCB 256, last).  fp32: the high 15 bits (shift 16, NCB 256, not last), then the low 16 (shift 0, NCB 512, last).
// A segment whose window does not cover C although it has magnitudes outside its window is counted again here (one
// workgroup: slow, correct, and only for tensors whose segments live on wildly different scales).
// ---------------------------------------------------------------------------------------------
template <int DT, bool FAST, int NCB>
__device__ __forceinline__ void fused_resolve(const void* in, int64_t numel, int64_t n_items, const SegGeom g, SelWs* ws, uint32_t* lds,
                                              uint32_t k_rem, int64_t k, int shift, uint32_t pmask, uint32_t pval, bool last, int clo2 = -1)
{
    constexpr int VEC = Traits<DT>::VEC;
    constexpr int NC = BFPQ_SELECT_HIST_COPIES;
    constexpr int PER = NCB / 64;
    const uint32_t dmask = (uint32_t)NCB * 128u - 1u;
    // ONE compute unit runs this while the rest of the chip idles: every instruction and every barrier counts.  The scans
    // over the coarse bins, 128 fine bins and 256 segments are each done by ONE wave (DPP scan, a few values per lane); the
    // 32 K window-slice words come as 16-byte cache-bypassing buffer loads, eight per thread, all in flight together, with no
    // branch around them (per segment one LDS word says where its slice for C lies, or that it has none).
    uint32_t* s_off = lds;               // [256] word offset of the segment's slice inside ws->windows | bit 31: no slice
    uint32_t* s_tc = lds + 256;          // [256]
    uint32_t* s_fine = lds + 512;        // [32][128]
    uint32_t* s_r = lds + 4624;          // [16]
    uint32_t* s_h = lds + 4640;          // [128]
    uint32_t* s_segwin = lds + 4768;     // [256]
    uint32_t* s_cv = lds + 5024;         // [NCB <= 512]
    uint32_t* s_segwin2 = lds + 5536;    // [256]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    STAMP(2, 0);
    uint32_t cv = 0;
    if (t < NCB) {
#pragma unroll
        for (int c = 0; c < NC; c++) cv += pub_load(&ws->coarse[c][t]);
    }
    const uint32_t sw = t < g.G ? pub_load(&ws->seg_win[t]) : 0u;
    const uint32_t sw2 = t < g.G ? pub_load(&ws->seg_win2[t]) : 0u;      // second window: first bin | bit 30: there is one | bit 31: elements outside BOTH windows exist
    __syncthreads();                                         // (the histogram's LDS is dead from here on)
    if (t < kMaxSeg) { s_segwin[t] = sw; s_segwin2[t] = sw2; }
    if (t < NCB) s_cv[t] = cv;
    __syncthreads();
    if (wv == 0) {                                           // coarse bins: NCB / 64 per lane
        uint32_t c[PER], mine = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { c[j] = s_cv[PER * lane + j]; mine += c[j]; }
        const uint32_t excl = wave_incl_scan(mine) - mine;
        const bool hit = mine && excl < k_rem && k_rem <= excl + mine;
        uint32_t bin = (uint32_t)(PER * lane), e = excl;
#pragma unroll
        for (int j = 0; j < PER - 1; j++) { if (k_rem > e + c[j]) { e += c[j]; bin = (uint32_t)(PER * lane + j + 1); } else break; }
        const uint32_t C0 = pick_lane(hit, bin), b0 = pick_lane(hit, e);
        if (lane == 0) { s_r[0] = C0; s_r[1] = b0; s_r[2] = 0; s_r[3] = 0; s_r[4] = 0; }      // (k_rem == 0: bin 0, nothing in front of it)
    }
    __syncthreads();
    const uint32_t C = s_r[0], before = s_r[1];
    STAMP(2, 1);
    for (int i = t; i < NC * NCB; i += kSelThreads) ws->coarse[i / NCB][i % NCB] = 0u;        // zero for the next call
    // segments of which neither window can answer for C (although they have elements outside them)
    int miss = 0;
    if (t < g.G) {
        const uint32_t clo = (sw & 0x3fffffffu) >> 7, clob = (sw2 & 0x3fffffffu) >> 7;
        const bool has2 = (sw2 >> 30) & 1u;
        const bool in_a = C - clo < (uint32_t)(kWinBins / 128), in_b = has2 && C - clob < (uint32_t)(kWinBins / 128);
        if (!in_a && !in_b && ((has2 ? sw2 : sw) >> 31)) miss = 1;
    }
    if (__syncthreads_or(miss)) {
        if (t < kMaxSeg) s_tc[t] = (uint32_t)miss;
        __syncthreads();
        for (int s = 0; s < g.G; s++) {
            if (!s_tc[s]) continue;                          // (block-uniform)
            if (t < 128) s_h[t] = 0;
            __syncthreads();
            const int64_t i0 = (int64_t)s * g.L, i1 = i0 + g.L < n_items ? i0 + g.L : n_items;
            for (int64_t it = i0 + t; it < i1; it += kSelThreads) {
                uint32_t r[VEC];
                sweep_load<DT, FAST>(in, it, n_items, numel, r);
#pragma unroll
                for (int j = 0; j < VEC; j++) {
                    const uint32_t key = mag_key<DT>(r[j]);
                    const uint32_t bin = (key >> shift) & dmask;
                    if ((FAST || it * VEC + j < numel) && (key & pmask) == pval && (bin >> 7) == C) atomicAdd(&s_h[bin & 127u], 1u);
                }
            }
            __syncthreads();
            if (t < 128) pub_store(&ws->windows[s][t], s_h[t]);
            if (t == 0) s_segwin[s] = (C << 7) | (1u << 30);                 // a 128-bin window at C
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (t < kMaxSeg) {
        uint32_t off = 0x80000000u;
        if (t < g.G) {
            const uint32_t w = s_segwin[t], clo = (w & 0x3fffffffu) >> 7;
            const bool narrow = (w >> 30) & 1u;
            const uint32_t w2 = s_segwin2[t], clob = (w2 & 0x3fffffffu) >> 7;
            if (narrow ? clo == C : C - clo < (uint32_t)(kWinBins / 128)) off = (uint32_t)t * kWinBins + (narrow ? 0u : (C - clo) * 128u);
            else if (((w2 >> 30) & 1u) && C - clob < (uint32_t)(kWinBins / 128)) off = (uint32_t)(kMaxSeg + t) * kWinBins + (C - clob) * 128u;   // (windows2 lies directly behind windows)
        }
        s_off[t] = off;
    }
    __syncthreads();
    STAMP(2, 2);
    // window slices: thread (sg, q) reads the fine bins 4q..4q+3 of C from the segments sg, sg + 32, ...
    typedef unsigned int v4u __attribute__((vector_size(16)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(&ws->windows[0][0], 0, (int)(sizeof(ws->windows) + sizeof(ws->windows2)), 0x00020000);
    const int sg = t >> 5, q = t & 31;
    v4u cnt[kMaxSeg / 32];
    uint32_t offs[kMaxSeg / 32];
#pragma unroll
    for (int j = 0; j < kMaxSeg / 32; j++) offs[j] = s_off[sg + 32 * j];
#pragma unroll
    for (int j = 0; j < kMaxSeg / 32; j++)
        cnt[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((offs[j] & 0x7fffffffu) + 4u * (uint32_t)q) * 4u), 0, 16 /* sc1: past this CU's caches */);
    v4u sum = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < kMaxSeg / 32; j++) {
        if (offs[j] >> 31) cnt[j] = (v4u){0u, 0u, 0u, 0u};
        sum += cnt[j];
    }
    *reinterpret_cast<v4u*>(&s_fine[sg * 128 + 4 * q]) = sum;
    __syncthreads();
    STAMP(2, 3);
    if (wv == 0) {                                           // the 128 fine bins of C: two per lane
        uint32_t f0 = 0, f1 = 0;
#pragma unroll
        for (int i = 0; i < 32; i++) { const uint2 p = *reinterpret_cast<const uint2*>(&s_fine[i * 128 + 2 * lane]); f0 += p.x; f1 += p.y; }
        const uint32_t mine = f0 + f1;
        const uint32_t excl = before + wave_incl_scan(mine) - mine;
        const bool hit = mine && excl < k_rem && k_rem <= excl + mine;
        const bool second = k_rem > excl + f0;
        const uint32_t dig0 = pick_lane(hit, (C << 7) + 2u * lane + (second ? 1u : 0u));
        const uint32_t run0 = pick_lane(hit, second ? excl + f0 : excl), ties0 = pick_lane(hit, second ? f1 : f0);
        if (lane == 0) { s_r[2] = dig0; s_r[3] = run0; s_r[4] = ties0; }
    }
    __syncthreads();
    const uint32_t digit = s_r[2], run = s_r[3], ties = s_r[4];
    const uint32_t tau = pval | (digit << shift);
    const uint32_t need = k_rem - run;
    STAMP(2, 4);
    if (!last) {                                             // more digits to come: the next launch reads the prefix
        if (t == 0) {
            bfpq_select_state* st = &ws->st;
            st->prefix = tau; st->prefix_mask = pmask | (dmask << shift); st->k_rem = (int64_t)need; st->k = k; st->done = 0;
            st->ties = (int64_t)ties;                        // (elements that share the prefix: the next digit's population)
            ws->ticket = 0u;
        }
        return;
    }
    // the column of tau: every segment's tie count
    if (q == (int)((digit & 127u) >> 2)) {
        const uint32_t comp = digit & 3u;
#pragma unroll
        for (int j = 0; j < kMaxSeg / 32; j++) s_tc[sg + 32 * j] = comp == 0 ? cnt[j][0] : (comp == 1 ? cnt[j][1] : (comp == 2 ? cnt[j][2] : cnt[j][3]));
    }
    __syncthreads();
    if (wv == 0) {                                           // the cut: 256 segments, four per lane
        const uint4 c4 = *reinterpret_cast<const uint4*>(&s_tc[4 * lane]);
        const uint32_t v[4] = {4 * lane < g.G ? c4.x : 0u, 4 * lane + 1 < g.G ? c4.y : 0u, 4 * lane + 2 < g.G ? c4.z : 0u, 4 * lane + 3 < g.G ? c4.w : 0u};
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        const uint32_t incl = wave_incl_scan(mine), excl = incl - mine;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t n_round = (uint32_t)((n_items + 63) / 64 * 64);
        uint32_t lo = 0, hi = 0, within = 0, ctot = 0;
        if (need >= total && need > 0) { lo = hi = n_round; }            // every tie goes ((need == 0: none does)
        else if (need > 0) {
            const bool hit = mine && excl <= need && need < excl + mine;
            uint32_t seg = 4u * lane, e = excl;
            if (need >= e + v[0]) { e += v[0]; seg++; if (need >= e + v[1]) { e += v[1]; seg++; if (need >= e + v[2]) { e += v[2]; seg++; } } }
            const uint32_t B = pick_lane(hit, seg), w0 = pick_lane(hit, need - e);
            ctot = pick_lane(hit, seg == 4u * lane ? v[0] : (seg == 4u * lane + 1 ? v[1] : (seg == 4u * lane + 2 ? v[2] : v[3])));
            const int64_t b0 = (int64_t)B * g.L, b1 = b0 + g.L < n_items ? b0 + g.L : n_items;
            lo = (uint32_t)b0; hi = w0 ? (uint32_t)b1 : (uint32_t)b0; within = w0;
        }
        STAMP(2, 5);
        if (lane == 0) {
            bfpq_select_state* st = &ws->st;
            st->prefix = tau; st->prefix_mask = pmask | (dmask << shift); st->k_rem = (int64_t)need; st->tau = tau; st->done = 1;
            st->need = (int64_t)need; st->ties = (int64_t)ties; st->k = k; st->tie_base = 0;
            st->flags = 1u; st->cut_lo = lo; st->cut_hi = hi; st->cut_within = within; st->cut_total = ctot;
            st->reserved = 0;
            ws->ticket = 0u;                                     // ready for the next call
        }
    }
}

}  // namespace bfpq_dev
