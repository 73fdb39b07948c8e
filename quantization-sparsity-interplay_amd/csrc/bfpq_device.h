// bfpq_device.h -- device helpers shared by the translation units of libbfpq.so: streaming loads/stores, packed 16-bit
// VALU spellings, wave scans, the bookkeeping of the unstructured path (segments, pieces, the ranked region), and the
// small host helpers of the launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <stddef.h>
#include "bfpq.h"
#include "bfpq_common.h"

extern "C" __attribute__((visibility("hidden"))) int bfpq_g_max_grid;    // bfpq_kernels.hip (bfpq_tune)

namespace bfpq_dev {
using namespace bfpq;

#ifndef BFPQ_MAXGRID
#define BFPQ_MAXGRID 1024          // 256 CUs x 4 workgroups, grid-stride beyond that (A/B over 5 shapes: 1024 best or tied)
#endif
#ifndef BFPQ_NT
#define BFPQ_NT 1                  // non-temporal loads/stores on the once-touched streams (A/B: +6..8 %)
#endif
constexpr int kThreads = 256;
#define kMaxGrid bfpq_g_max_grid

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

struct SelWs;
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
    int kb_lo, kb_span;       // hot16 path: dtype exponent fields of the block max it takes (kb_span < 0: never)
    uint32_t maxv_c;          //   (2^m - 1) * 2^(e - m) in dtype bits = (fp32-biased e << MBITS) + maxv_c
    SelWs* selws;             // NM == -1 (global magnitude threshold): the select workspace (threshold + tie bookkeeping)
};

// A list of tensors in one launch (bfpq_fake_quantize_batched): up to kMaxBatch descriptors travel in the kernel arguments
// (no device memory, graph-capturable).  The tensors share dtype / block / mantissa width; each is a flat array of lane
// items cut into chunks of 256 (the last chunk of a tensor is ragged), the chunks of all tensors form one index space
// that the workgroups stride over.  flags bit 0: apply the N:M mask to this tensor (a Linear's weight) or not (its activation).
constexpr int kCodeBitsE4M3 = 108;                 // FusedArgs.code_bits: e4m3 image bytes + E8M0 scales (internal: bfpq_quantize_mx8 on fp32 tensors)
#ifndef BFPQ_MAX_BATCH
#define BFPQ_MAX_BATCH 64          // (A/B, tools_dev/ab_list.py, one box: ViT-L's 144 weights 16 / 32 / 64 / 112 per launch = 419 / 409 / 405-425 / 451 us,
#endif                             //  OPT-125m's 72 weights 128 / 121 / 117-122 / 122 us: longer launches drift apart, shorter ones pay their ramps)
constexpr int kMaxBatch = BFPQ_MAX_BATCH;
struct BatchDesc { const void* in; void* out; int64_t n_items; uint32_t chunk0; uint32_t flags; };
struct BatchArgs { int n; uint32_t total_chunks; BatchDesc d[kMaxBatch]; };


// the fused kernels are compiled once per dtype (bfpq_fused_dt.hip); these are their launchers, dtype in the name
#define BFPQ_HIDDEN __attribute__((visibility("hidden")))
BFPQ_HIDDEN int fused_launch_0(const FusedArgs& a, int M, bool sfirst, hipStream_t s);
BFPQ_HIDDEN int fused_launch_1(const FusedArgs& a, int M, bool sfirst, hipStream_t s);
BFPQ_HIDDEN int fused_launch_2(const FusedArgs& a, int M, bool sfirst, hipStream_t s);
BFPQ_HIDDEN int fused_threshold_0(const FusedArgs& a, hipStream_t s);
BFPQ_HIDDEN int fused_threshold_1(const FusedArgs& a, hipStream_t s);
BFPQ_HIDDEN int fused_threshold_2(const FusedArgs& a, hipStream_t s);
BFPQ_HIDDEN int fused_mx8_0(const FusedArgs& a, hipStream_t s);
BFPQ_HIDDEN int fused_mx8_1(const FusedArgs& a, hipStream_t s);
BFPQ_HIDDEN int fused_mx8_2(const FusedArgs& a, hipStream_t s);
BFPQ_HIDDEN int fused_batched_0(const FusedArgs& a, const BatchArgs& b, int M, bool sfirst, hipStream_t s);
BFPQ_HIDDEN int fused_batched_1(const FusedArgs& a, const BatchArgs& b, int M, bool sfirst, hipStream_t s);
BFPQ_HIDDEN int fused_batched_2(const FusedArgs& a, const BatchArgs& b, int M, bool sfirst, hipStream_t s);
inline int fused_launch(int dtype, const FusedArgs& a, int M, bool sfirst, hipStream_t s)
{
    return dtype == BFPQ_F32 ? fused_launch_0(a, M, sfirst, s) : (dtype == BFPQ_F16 ? fused_launch_1(a, M, sfirst, s) : fused_launch_2(a, M, sfirst, s));
}
inline int fused_threshold(int dtype, const FusedArgs& a, hipStream_t s)
{
    return dtype == BFPQ_F32 ? fused_threshold_0(a, s) : (dtype == BFPQ_F16 ? fused_threshold_1(a, s) : fused_threshold_2(a, s));
}
inline int fused_mx8(int dtype, const FusedArgs& a, hipStream_t s)
{
    return dtype == BFPQ_F32 ? fused_mx8_0(a, s) : (dtype == BFPQ_F16 ? fused_mx8_1(a, s) : fused_mx8_2(a, s));
}
inline int fused_batched(int dtype, const FusedArgs& a, const BatchArgs& b, int M, bool sfirst, hipStream_t s)
{
    return dtype == BFPQ_F32 ? fused_batched_0(a, b, M, sfirst, s) : (dtype == BFPQ_F16 ? fused_batched_1(a, b, M, sfirst, s) : fused_batched_2(a, b, M, sfirst, s));
}

// wave-level inclusive scan (lane order) by DPP: row_shr 1/2/4/8 inside the rows of 16 lanes, then row_bcast15 /
// row_bcast31 carry the row totals forward (six VALU adds; the ds_bpermute form is six LDS round trips)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ int64_t uniform64(int64_t v)           // a wave-uniform value, moved to scalar registers
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63);
}

// ---------------------------------------------------------------------------------------------
// Unstructured pruning: bookkeeping shared by the launches (histogram [+ resolve], apply).
//
// Tie rule: of the elements EQUAL to the threshold, the first `need` in flat index order go (lower ranks
// first on a row-sharded tensor) -- a pure function of the flat index, so a sharded run and a single-device
// run zero the same elements.  Ranks are never materialised per tile.  Instead:
//   * the histogram launch cuts the tensor into <= 256 flat-contiguous SEGMENTS, one workgroup each; the
//     workgroup's private LDS histogram is, at the end, also an exact count of every magnitude inside the
//     segment, and a 2048-bin WINDOW of it around the segment's own k-quantile is left in the workspace;
//   * the resolve step (single device, 16-bit dtypes: the last workgroup of the histogram launch to draw its
//     ticket; otherwise the resolve launch) finds the threshold tau and reads every segment's tie count out of
//     its window.  The cut -- the flat position where the cumulative tie count reaches `need` -- falls into one
//     segment: everything in front of that segment prunes all its ties (compare against tau + 1), everything
//     behind it none (compare against tau);
//   * the apply launch (threshold_apply / the fused quantizer) starts with kCutWGs extra workgroups that own
//     the CUT SEGMENT: each counts the ties in front of its share of the segment's tiles (L2-resident re-reads,
//     in parallel, nobody waits for anybody) and ranks its own tiles with a wave scan.  The ordinary
//     workgroups do not store the tiles of the cut segment.
// ---------------------------------------------------------------------------------------------
constexpr int kSelThreads = 1024;
constexpr int kMaxSeg = BFPQ_SELECT_MAX_SEGMENTS;
constexpr int kWinBins = BFPQ_SELECT_WINDOW_BINS;
constexpr int kFineBins = 32768, kCoarseBins = 256;
constexpr int kCutWGs = 64;                   // workgroups of the apply launch that own the cut segment (dispatched first)

struct SelWs {
    bfpq_select_state st;
    uint32_t ticket;                          // fused histogram + resolve launch: workgroups that have published their segment
    uint32_t pad_[11];
    uint32_t coarse[BFPQ_SELECT_HIST_COPIES][2 * kCoarseBins];   // fused launch: coarse histogram (256 bins of 128; fp32's low-16 digit: 512), zero between calls
    // histogram buffers of the launch-pair path (fp32; diagnostics), one per radix pass; all zero between calls: the APPLY launch
    // clears what the histogram launches of its call dirtied (see thr_setup; bfpq_select_reset after a select with no apply)
    uint32_t hist[3][BFPQ_SELECT_HIST_COPIES][BFPQ_SELECT_HIST_ENTRIES];
    uint32_t seg_ties[kMaxSeg];               // elements equal to tau per segment (resolve launch, window-miss path)
    uint32_t seg_win[kMaxSeg];                // first bin of the segment's window | bit 31: magnitudes outside the window exist | bit 30: 128-bin window
    uint32_t seg_win2[kMaxSeg];               // the segment's SECOND window (fused launch): first bin | bit 30: there is one | bit 31: magnitudes outside both windows exist
    uint32_t windows[kMaxSeg][kWinBins];      // (the apply launch also dumps the ordinary workgroups' cut-segment tiles here)
    uint32_t windows2[kMaxSeg][kWinBins];     // the second window per segment, directly behind `windows`: where the segment's quantile would lie if its rank were off
                                              // by a segment's sampling error (atoms next to the threshold); fp32's low-16 digit: at a position every workgroup
                                              // derives from the state alone (where the threshold lies if the low bits are spread evenly)
};
static_assert(sizeof(SelWs) == BFPQ_SELECT_WS_BYTES, "bfpq.h: BFPQ_SELECT_WS_BYTES");
static_assert(offsetof(SelWs, hist) % 16 == 0 && offsetof(SelWs, windows) % 16 == 0 && offsetof(SelWs, coarse) % 16 == 0, "SelWs: 16-byte aligned arrays");

struct SegGeom { int G; int64_t L; };         // G segments of L lane items (L a multiple of 64)
__host__ __device__ inline SegGeom seg_geom(int64_t n_items)
{
    SegGeom g;
    int64_t g0 = (n_items + 2047) / 2048;
    if (g0 < 1) g0 = 1;
    if (g0 > kMaxSeg) g0 = kMaxSeg;
    int64_t L = (n_items + g0 - 1) / g0;
    L = (L + 63) / 64 * 64;
    if (L < 64) L = 64;
    g.L = L;
    g.G = (int)((n_items + L - 1) / L);
    if (g.G < 1) g.G = 1;
    return g;
}

// exclusive prefix sum of one value per thread over the workgroup (blockDim.x a multiple of 64, <= 1024); every thread
// gets the total.  s_part: 16 words of LDS.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_part, uint32_t* total)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    __syncthreads();                                        // (s_part may still be read from the previous scan)
    if (lane == 63) s_part[w] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (int i = 0; i < nw; i++) { const uint32_t p = s_part[i]; off += i < w ? p : 0u; tot += p; }
    *total = tot;
    return off + incl - v;
}

// threshold state of one apply workgroup, wave-uniform
struct ThrCtx {
    uint32_t tau;
    bool on;
    int64_t cut_lo, cut_hi;    // the cut segment in lane items (multiples of 64; cut_hi possibly n_items); empty: no tile is ranked
    uint32_t within;           // ties to prune inside the cut segment, counted from its start
    uint32_t cut_total;        // ties inside the cut segment
    // per tile:
    uint32_t teff;             // ordinary workgroups: prune keys below this (tau + 1 in front of the cut segment, tau behind it)
    bool ranked;               // cut workgroups: the ties of this tile are ranked ...
    uint32_t before;           //   ... behind this many ties of the segment's earlier tiles
    uint4* dump;               // where the ordinary workgroups' stores of cut-segment tiles go (never read)
};

template <int DT, bool FAST>
__device__ __forceinline__ void sweep_load(const void* in, int64_t item, int64_t n_items, int64_t numel, uint32_t* raw);

template <int DT> __device__ __forceinline__ uint32_t count_eq(const uint32_t* raw, uint32_t tau)
{
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < Traits<DT>::VEC; j++) c += mag_key<DT>(raw[j]) == tau;
    return c;
}

// where the cut lies for a tensor whose per-segment tie counts are v[0..G) (one per thread, 0 beyond G): the three numbers
// the apply launch needs (+ the cut segment's own tie count).  Block-uniform result through s_res (4 words); uses block_excl_scan's barriers.
//   need_local: ties of THIS device to prune (need - tie_base); total: this device's ties
__device__ __forceinline__ void cut_from_seg_ties(uint32_t v, int64_t need_local, const SegGeom& g, int64_t n_items, uint32_t* s_part, uint32_t* s_res)
{
    uint32_t total;
    const uint32_t excl = block_excl_scan(v, s_part, &total);
    const int tid = threadIdx.x;
    const uint32_t n_round = (uint32_t)((n_items + 63) / 64 * 64);
    if (tid == 0) {
        if (need_local <= 0) { s_res[0] = 0; s_res[1] = 0; s_res[2] = 0; s_res[3] = 0; }                         // no tie goes
        else if (need_local >= (int64_t)total) { s_res[0] = n_round; s_res[1] = n_round; s_res[2] = 0; s_res[3] = 0; }   // every tie goes
    }
    if (need_local > 0 && need_local < (int64_t)total && v && (int64_t)excl <= need_local && need_local < (int64_t)excl + v) {
        const int64_t b0 = (int64_t)tid * g.L, b1 = b0 + g.L < n_items ? b0 + g.L : n_items;
        const uint32_t within = (uint32_t)(need_local - excl);
        s_res[0] = (uint32_t)b0; s_res[1] = within ? (uint32_t)b1 : (uint32_t)b0; s_res[2] = within; s_res[3] = v;
    }
    __syncthreads();
}

// Threshold + cut of this tensor for one apply workgroup, from the workspace the resolve step left (block-uniform result;
// blockDim.x == 256 threads; barriers only on the window-miss path).  s_part: 16 words, s_res: 8 words of LDS.
template <int DT>
__device__ __forceinline__ void thr_setup(ThrCtx& t, SelWs* ws, int64_t n_items, uint32_t* s_part, uint32_t* s_res)
{
    const bfpq_select_state* st = &ws->st;
    const uint32_t flags = st->flags;
    if (flags & 2u) {
        // the histograms of this call have been consumed by the resolve launch: clear them for the next call
        constexpr int NC = BFPQ_SELECT_HIST_COPIES;
        const int i0 = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
        if constexpr (DT == BFPQ_F32) {
            for (int i = i0; i < 3 * NC * 512; i += step)                      // passes x copies x 2048 bins
                reinterpret_cast<uint4*>(ws->hist[i / (NC * 512)][(i / 512) % NC])[i % 512] = make_uint4(0, 0, 0, 0);
        } else {
            for (int i = i0; i < NC * BFPQ_SELECT_HIST_ENTRIES / 4; i += step) reinterpret_cast<uint4*>(ws->hist[0][0])[i] = make_uint4(0, 0, 0, 0);
        }
    }
    t.tau = st->tau;
    t.on = st->k > 0;
    t.cut_lo = t.cut_hi = 0; t.within = 0; t.cut_total = 0;  // empty cut segment at the front: no tie is pruned
    t.teff = 0; t.ranked = false; t.before = 0;
    t.dump = reinterpret_cast<uint4*>(ws->windows);
    if (!t.on) { t.tau = 0; return; }                        // (k == 0: nothing is below a threshold of 0)
    if (flags & 1u) { t.cut_lo = st->cut_lo; t.cut_hi = st->cut_hi; t.within = st->cut_within; t.cut_total = st->cut_total; return; }
    // the resolve launch had to recount some segments (window miss): their tie counts are in seg_ties
    const int64_t need = st->need, ties = st->ties;
    const uint32_t n_round = (uint32_t)((n_items + 63) / 64 * 64);
    if (need <= 0) return;
    if (need >= ties) { t.cut_lo = t.cut_hi = n_round; return; }
    const SegGeom g = seg_geom(n_items);
    const uint32_t v = (int)threadIdx.x < g.G ? ws->seg_ties[threadIdx.x] : 0u;
    cut_from_seg_ties(v, need - st->tie_base, g, n_items, s_part, s_res);
    t.cut_lo = s_res[0]; t.cut_hi = s_res[1]; t.within = s_res[2]; t.cut_total = s_res[3];
}

// prune bits of one lane item of a RANKED tile (bit j = element j goes): everything below tau, and the ties whose rank
// (flat order inside the cut segment) is below `within`; the 64 lanes of the wave hold one tile
template <int DT>
__device__ __forceinline__ uint32_t thr_rank_bits(const uint32_t* raw, bool valid, const ThrCtx& t)
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
    uint32_t prune = ltm;
    const uint32_t cnt = __popc(eqm);
    const uint32_t incl = wave_incl_scan(cnt);
    uint32_t r = t.before + incl - cnt;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        if ((eqm >> j) & 1u) { if (r < t.within) prune |= 1u << j; r++; }
    }
    return prune;
}

// A cut workgroup (blockIdx.x < kCutWGs, 256 threads) walks its share of the cut segment's tiles: first the ties in front of
// its first tile (strided re-read with eight independent loads per thread in flight), then its tiles in groups of four (one per
// wave) with the running tie count handed from group to group.  process(item, raw, valid, before) is called once per lane item.
template <int DT, bool FAST, class F>
__device__ __forceinline__ void cut_wg_run(const ThrCtx& t, const void* in, int64_t numel, int64_t n_items, uint32_t* s_x, F&& process)
{
    constexpr int VEC = Traits<DT>::VEC;
    if (!t.on || t.cut_lo >= t.cut_hi) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nB = (t.cut_hi - t.cut_lo + 63) / 64;
    const int64_t T = (nB + kCutWGs - 1) / kCutWGs;
    const int64_t g0 = (int64_t)blockIdx.x * T, g1 = g0 + T < nB ? g0 + T : nB;
    if (g0 >= g1) return;                                                    // (block-uniform)
    // ties in front of this workgroup's first tile: counted from whichever end of the segment is nearer (the segment's total is
    // known), so the longest count is half a segment -- the last workgroups' counts were the tail of the whole launch on large tensors
    uint32_t c = 0;
    const bool from_end = 2 * g0 > nB;
    {
        int64_t it = (from_end ? t.cut_lo + g0 * 64 : t.cut_lo) + tid;
        const int64_t end = from_end ? (t.cut_hi < n_items ? t.cut_hi : n_items) : t.cut_lo + g0 * 64;
        for (; it + 7 * 256 < end; it += 8 * 256) {
            uint32_t r[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; u++) sweep_load<DT, FAST>(in, it + u * 256, n_items, numel, r[u]);
#pragma unroll
            for (int u = 0; u < 8; u++) c += count_eq<DT>(r[u], t.tau);
        }
        for (; it < end; it += 256) {
            uint32_t r[VEC];
            sweep_load<DT, FAST>(in, it, n_items, numel, r);
            c += count_eq<DT>(r, t.tau);
        }
    }
    c = wave_sum(c);
    if (lane == 0) s_x[wave] = c;
    __syncthreads();
    uint32_t running = s_x[0] + s_x[1] + s_x[2] + s_x[3];
    if (from_end) running = t.cut_total - running;
    for (int64_t g = g0; g < g1; g += 4) {                                   // (block-uniform trip count)
        const int64_t tile = g + wave;
        const bool active = tile < g1;
        const int64_t item = t.cut_lo + tile * 64 + lane;
        const bool valid = active && item < n_items;
        uint32_t raw[VEC];
        sweep_load<DT, FAST>(in, item, n_items, numel, raw);
        const uint32_t mine = wave_sum(valid ? count_eq<DT>(raw, t.tau) : 0u);
        __syncthreads();                                                     // (s_x read by everybody)
        if (lane == 0) s_x[wave] = active ? mine : 0u;
        __syncthreads();
        uint32_t before = running;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t x = s_x[w]; before += w < wave ? x : 0u; running += x; }
        if (active) process(item, raw, valid, before);
    }
}

// ---------------------------------------------------------------------------------------------
// Unstructured: radix select on magnitude keys (15 bits in one pass for 16-bit dtypes; 11 + 11 + 9 for fp32).
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline void select_digit(int dtype, int pass, int* shift, int* nbits)
{
    if (dtype == BFPQ_F32) {
        if (pass == 0) { *shift = 20; *nbits = 11; }
        else if (pass == 1) { *shift = 9; *nbits = 11; }
        else { *shift = 0; *nbits = 9; }
    } else { *shift = 0; *nbits = 15; }
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

// one lane item of a sweep.  FAST: pointer 16-B aligned and numel a multiple of the vector width -> unconditional vector
// loads (index clamped), so a prefetch stays in flight; otherwise element loads with bounds checks.
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

// ---------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------
inline float h_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t h_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

inline float h_round_bf16(float f)
{
    uint32_t u = h_f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f;
    u += 0x7fffu + ((u >> 16) & 1u);
    return h_u2f(u & 0xffff0000u);
}

// fp32 -> nearest fp16 (ties to even) -> fp32, via exact double arithmetic on the fp16 grid
inline float h_round_f16(float f)
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

inline float h_round(float f, int dtype) { return dtype == BFPQ_F32 ? f : (dtype == BFPQ_F16 ? h_round_f16(f) : h_round_bf16(f)); }

inline int dtype_vec(int dtype) { return dtype == BFPQ_F32 ? 4 : 8; }
inline int dtype_size(int dtype) { return dtype == BFPQ_F32 ? 4 : 2; }
inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// workgroups for `work_threads` grid-stride work items: at most kMaxGrid, and balanced -- every
// workgroup gets the same number of sweeps (22016 blocks of work -> 18 sweeps x 1224 workgroups, not
// 1280 workgroups of which 256 do one sweep more)
inline int grid_for_cap(int64_t work_threads, int64_t cap)
{
    int64_t g = (work_threads + kThreads - 1) / kThreads;
    if (g < 1) g = 1;
    if (g <= cap) return (int)g;
    const int64_t sweeps = (g + cap - 1) / cap;
    return (int)((g + sweeps - 1) / sweeps);
}
inline int grid_for(int64_t work_threads) { return grid_for_cap(work_threads, kMaxGrid); }
// The packed-output instantiations write a quarter (4-bit codes) or half (e4m3 image) of the bytes they read.  After the r03 diet
// (83 vector instructions per item instead of 148) and with two sweeps of loads in flight they are bound by bytes in flight, and fit
// 64 VGPRs: the cap is two workgroups per SIMD slot pair = 2048, one full round of the chip at 8 waves per SIMD (interleaved sweep on
// [4096,11008] bf16 2:4 packed, tools_dev/ab_packed.py: 1024 workgroups 24.1 us, 1536 24.1, 2048 23.1, 2752 23.4).
inline int grid_for_packed(int64_t work_threads) { return grid_for_cap(work_threads, (int64_t)kMaxGrid * 2); }

}  // namespace bfpq_dev
