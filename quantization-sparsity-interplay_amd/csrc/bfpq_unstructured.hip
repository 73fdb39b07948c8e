// bfpq_unstructured.hip -- the unstructured (global magnitude threshold) path of libbfpq.so: the histogram launch (with the
// resolve step folded into its last workgroup on a single device), the resolve launch of the multi-GPU / fp32 launch-pair
// form, the prune-only apply launch, and their C-ABI entry points (include/bfpq.h).  The fused prune+quantize apply is
// k_fused_flat<.., -1, ..> in bfpq_fused.h.  Reference: src/transformers/bfp/bfp_ops.py:61-71.
#include <hip/hip_runtime.h>
#include "bfpq.h"
#include "bfpq_common.h"
#include "bfpq_device.h"
#include "bfpq_select.h"

using namespace bfpq;
using namespace bfpq_dev;

namespace {

// Launch 1: histogram of the current digit.  One workgroup per flat-contiguous segment (see the block comment at ThrCtx);
// LDS histogram (32 768 bins = 128 KB for 16-bit keys).  On the pass that decides the threshold the workgroup also leaves a
// window of its private histogram in the workspace.
//   fuse != 0 (single device, 16-bit dtypes): only the 256 coarse sums go to the global histogram; the workgroup publishes
//     its window, draws a ticket, and the last one resolves the selection (bfpq_select.h) -- one launch.
//   fuse == 0: the non-zero fine bins are flushed by integer atomics (deterministic) for the resolve launch.
// Where bin b of the 15-bit histogram lives in the LDS: the low five bits (the bank) are mixed with bits 5-14.  A tensor that went
// through the quantizer before it is pruned (first = 'q', or a checkpoint stored in HBFP values) has only a few distinct magnitudes and
// their low mantissa bits are all zero: unmixed, EVERY LDS atomic of a wave landed in bank 0 and the launch took 58 us on [5120,5120] bf16
// instead of 20.  The map is a bijection that keeps each aligned group of 32 bins (hence each coarse bin of 128) in place, so sums over
// coarse bins do not see it; whoever reads a fine bin asks through it.
__device__ __forceinline__ uint32_t swz15(uint32_t b) { return b ^ (((b >> 5) ^ (b >> 10)) & 31u); }

template <int DT, bool FAST>
__global__ void __launch_bounds__(kSelThreads) k_select_hist(const void* in, int64_t numel, int pass, int shift, int nbits, int first, int last,
                                                             SelWs* ws, uint32_t* hist_ext, int64_t k, int64_t numel_global, int fuse)
{
    using T = Traits<DT>;
    constexpr int VEC = T::VEC;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];
    uint32_t* hist = hist_ext ? hist_ext : ws->hist[pass][0];
    __shared__ __attribute__((aligned(16))) uint32_t s_coarse[kCoarseBins];
    __shared__ uint32_t s_res[4];
    const int t = threadIdx.x;
    const int nbins = 1 << nbits;
    STAMP(0, 0);
    const int64_t n_items = (numel + VEC - 1) / VEC;
    const SegGeom g = seg_geom(n_items);
    const int64_t i0 = (int64_t)blockIdx.x * g.L, i1 = i0 + g.L < n_items ? i0 + g.L : n_items;
    // 16-bit dtypes: the thread's first item is asked for before the LDS is cleared, and the barrier behind the clearing doubles as the
    // vote on how the segment is counted (below): no barrier of its own
    [[maybe_unused]] uint4 v_first = make_uint4(0, 0, 0, 0);
    [[maybe_unused]] uint32_t g_first[VEC];
    constexpr bool kFast16 = FAST && VEC == 8;
    const bool may_rep = kFast16 || (nbits == 15 && first != 0);               // the first 15-bit digit of any dtype
    if constexpr (kFast16) {
        const int64_t item0 = i0 + t, lastv = n_items - 1;
        v_first = reinterpret_cast<const uint4*>(in)[item0 < lastv ? item0 : lastv];
    } else sweep_load<DT, FAST>(in, i0 + t, n_items, numel, g_first);
    for (int i = t; i < nbins / 4; i += kSelThreads) reinterpret_cast<uint4*>(s_hist)[i] = make_uint4(0, 0, 0, 0);
    if (t < kCoarseBins) s_coarse[t] = 0;
    bool low_first = !may_rep;
    if constexpr (kFast16) low_first = i0 + t < i1 && ((v_first.x | v_first.y | v_first.z | v_first.w) & 0x001f001fu) != 0u;
    else if (may_rep && i0 + t < i1) {
#pragma unroll
        for (int j = 0; j < VEC; j++) low_first |= (FAST || (i0 + t) * VEC + j < numel) && ((mag_key<DT>(g_first[j]) >> shift) & 31u) != 0u;
    }
    bool rep = __syncthreads_or((int)low_first) == 0;      // (block-uniform) the segment is counted into replicas, see below
    const uint32_t pmask = first ? 0u : ws->st.prefix_mask, pval = first ? 0u : ws->st.prefix;   // pass 0 reads no state
    const uint32_t dmask = (uint32_t)nbins - 1u;
    auto fine_at = [&](uint32_t b) __attribute__((always_inline)) -> uint32_t {      // the count of fine bin b of the 15-bit digit
        if (!rep) return s_hist[swz15(b)];
        if (b & 31u) return 0u;
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { const uint4 c4 = *reinterpret_cast<const uint4*>(&s_hist[b + 4 * q]); sum += c4.x + c4.y + c4.z + c4.w; }
        return sum;
    };
    if constexpr (FAST && VEC == 8) {
        // 16-bit dtypes: one pass over the whole 15-bit key (shift 0, no prefix to match): two keys per packed and/min,
        // unconditional LDS atomics; one-ahead prefetch with a clamped, unconditional load
        const uint32_t absm = T::ABS | (T::ABS << 16), nanc = (T::INF + 1u) | ((T::INF + 1u) << 16);
        // Zeros are counted in a register and added once: a tensor that is already sparse sends half of a wave's atomics to ONE
        // LDS word otherwise (52 % zeros: the launch took twice as long).
        const int64_t lastv = n_items - 1;
        // COARSE-KEYED tensors (r03): when the low five bits of every key of the segment are zero -- HBFP values of up to three mantissa bits
        // in bf16, the quantizer's output under first = 'q' or a stored checkpoint -- 31 of every 32 bins stay empty while a wave's 64 atomics fall
        // on a dozen words.  Such a segment is counted into REPLICAS: lane l adds to word swz15(key) ^ (l & 31) of the key's own aligned group
        // of 32, so equal keys of a wave meet in different words and banks; a bin's count is the sum of its group (fine_at below), the coarse
        // sums see no difference.  The first item of every thread decides (the vote rides on the barrier behind the LDS clearing); a key with low bits met later raises `viol` and the
        // segment is counted again the plain way (a mixed tensor pays twice, the result is the same).
        auto count_pass = [&](auto rep_tag, uint4 v) __attribute__((always_inline)) {      // v: the thread's first item (loaded by the caller)
            constexpr bool REP = decltype(rep_tag)::value;
            const uint32_t l5 = REP ? (uint32_t)(t & 31) : 0u;
            int64_t item = i0 + t;
            uint32_t zeros = 0, low = 0;
            for (; item < i1; item += kSelThreads) {
                const int64_t pf = item + kSelThreads;
                const uint4 nv = reinterpret_cast<const uint4*>(in)[pf < lastv ? pf : lastv];
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t k2 = pk_min_i16_s(d[j] & absm, nanc);
                    if constexpr (REP) low |= k2;
                    const uint32_t s2 = k2 ^ (((k2 >> 5) ^ (k2 >> 10)) & 0x001f001fu);     // swz15 of both halves at once (what crosses the halves is masked away)
                    const uint32_t ka = s2 & 0xffffu, kb = s2 >> 16;                       // (swz15(k) == 0 only for k == 0)
                    if (ka) atomicAdd(&s_hist[ka ^ l5], 1u); else zeros++;
                    if (kb) atomicAdd(&s_hist[kb ^ l5], 1u); else zeros++;
                }
                v = nv;
            }
            if (zeros) atomicAdd(&s_hist[0], zeros);
            return (low & 0x001f001fu) != 0u;
        };
        const uint4 v0 = v_first;
        if (rep) {
            const bool viol = count_pass(std::true_type{}, v0);
            if (__syncthreads_or((int)viol)) {                                             // (block-uniform) not coarse-keyed after all: again, the plain way
                rep = false;
                for (int i = t; i < nbins / 4; i += kSelThreads) reinterpret_cast<uint4*>(s_hist)[i] = make_uint4(0, 0, 0, 0);
                __syncthreads();
                count_pass(std::false_type{}, v0);
            }
        } else count_pass(std::false_type{}, v0);
    } else {
        // (fp32, ragged shapes, later digits) the same two passes for the first 15-bit digit: fp32 tensors that hold HBFP or bf16 values have
        // five zero low bits in it as well
        auto count_generic = [&](auto rep_tag) __attribute__((always_inline)) {
            constexpr bool REP = decltype(rep_tag)::value;
            const uint32_t l5 = REP ? (uint32_t)(t & 31) : 0u;
            uint32_t low = 0, zero_bin = 0;
            int64_t item = i0 + t;
            uint32_t cur[VEC], nxt[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j++) cur[j] = g_first[j];
            for (; item < i1; item += kSelThreads) {
                sweep_load<DT, FAST>(in, item + kSelThreads, n_items, numel, nxt);
#pragma unroll
                for (int j = 0; j < VEC; j++) {
                    const uint32_t key = mag_key<DT>(cur[j]);
                    const bool real = FAST || item * VEC + j < numel;
                    if (real && (key & pmask) == pval) {
                        const uint32_t bin = (key >> shift) & dmask;
                        if constexpr (REP) low |= bin;
                        if (bin) atomicAdd(&s_hist[nbits == 15 ? swz15(bin) ^ l5 : bin], 1u);
                        else zero_bin++;                       // (as in the 16-bit loop: an already-sparse tensor sends half of a wave's atomics to ONE word otherwise)
                    }
                }
#pragma unroll
                for (int j = 0; j < VEC; j++) cur[j] = nxt[j];
            }
            if (zero_bin) atomicAdd(&s_hist[0], zero_bin);
            return (low & 31u) != 0u;
        };
        if (rep) {
            const bool viol = count_generic(std::true_type{});
            if (__syncthreads_or((int)viol)) {
                rep = false;
                for (int i = t; i < nbins / 4; i += kSelThreads) reinterpret_cast<uint4*>(s_hist)[i] = make_uint4(0, 0, 0, 0);
                __syncthreads();
                count_generic(std::false_type{});
            }
        } else count_generic(std::false_type{});
    }
    STAMP(0, 1);
    __syncthreads();
    STAMP(0, 2);
    if (fuse) {                                             // (host: only with nbits == 15 and pass 0; 16-bit dtypes: the only digit, fp32: the high one)
        coarse_from_lds<kCoarseBins, false>(s_hist, s_coarse);
        __syncthreads();
        const double frac = numel_global > 0 ? (double)k / (double)numel_global : 0.0;
        if (!seg_publish_and_ticket<kCoarseBins>(s_coarse, s_res, ws, frac, [&](uint32_t b) { return fine_at(b); })) return;
        STAMP(0, 6);
        fused_resolve<DT, FAST, kCoarseBins>(in, numel, n_items, g, ws, s_hist, (uint32_t)k, k, shift, 0u, 0u, last != 0);
        STAMP(0, 7);
        return;
    }
    const int copy = blockIdx.x % BFPQ_SELECT_HIST_COPIES;
    hist += (size_t)copy * BFPQ_SELECT_HIST_ENTRIES;
    const bool windows = last && nbits == 15;
    if (nbits == 15) {
        // coarse histogram (256 bins of 128): four threads per coarse bin, each sums 32 bins with 16-byte LDS reads in a
        // rotated order (two lanes per bank), then a quad reduction
        const int q = t >> 2, r = t & 3;
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint4 c4 = *reinterpret_cast<const uint4*>(&s_hist[q * 128 + ((j + q) & 7) * 16 + r * 4]);
            sum += c4.x + c4.y + c4.z + c4.w;
        }
        sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
        sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
        if (r == 0) { s_coarse[q] = sum; if (sum) atomicAdd(&hist[kFineBins + q], sum); }
        __syncthreads();
    }
    STAMP(0, 3);
    // window: the 16 coarse bins (2048 bins) around the one that holds the segment's own k-quantile.  The first wave
    // finds it (four coarse sums per lane, one wave scan) while the others already flush.
    if (windows && t < 64) {
        const uint4 c4 = *reinterpret_cast<const uint4*>(&s_coarse[t * 4]);
        const uint32_t mine = c4.x + c4.y + c4.z + c4.w;
        const uint32_t incl = wave_incl_scan(mine);
        const uint32_t seg_elems = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint64_t target = numel_global > 0 ? (uint64_t)((double)seg_elems * ((double)k / (double)numel_global)) : 0ull;   // (an anchor, not a count)
        if (seg_elems && target >= seg_elems) target = seg_elems - 1;
        const uint32_t excl = incl - mine;
        if (t == 0) s_res[0] = 0;
        if (mine && excl <= target && target < (uint64_t)excl + mine) {
            uint32_t e = excl;
            int A = t * 4;
            if (target >= e + c4.x) { e += c4.x; A++; if (target >= e + c4.y) { e += c4.y; A++; if (target >= e + c4.z) A++; } }
            int clo = A - kWinBins / 256;
            clo = clo < 0 ? 0 : (clo > kCoarseBins - kWinBins / 128 ? kCoarseBins - kWinBins / 128 : clo);
            uint32_t inside = 0;
            for (int j = 0; j < kWinBins / 128; j++) inside += s_coarse[clo + j];
            s_res[0] = (uint32_t)clo * 128u;
            pub_store(&ws->seg_win[blockIdx.x], ((uint32_t)clo * 128u) | (inside != seg_elems ? 0x80000000u : 0u));
        }
        if (seg_elems == 0 && t == 0) pub_store(&ws->seg_win[blockIdx.x], 0u);          // (an empty segment: no window, nothing outside it)
    }
    // flush of the non-zero bins by integer atomics (deterministic) into this workgroup's COPY of the histogram: with
    // all workgroups adding into one copy every hot address takes 256 serialised adds (~3 us behind the streaming loop).
    // The LDS reads of a thread are issued together.
    if (nbits == 15) {
        // one coarse bin (128 fine bins, two 256-byte atomic wave-instructions) per wave and trip; the empty ones -- all but
        // 15-20 of the 256 for a weight tensor -- are skipped on their coarse sum
        const int lane = t & 63;
        for (int cb = t >> 6; cb < kCoarseBins; cb += kSelThreads / 64) {
            if (s_coarse[cb] == 0) continue;                   // (wave-uniform)
            const int i = cb * 128 + lane;
            const uint32_t c0 = fine_at((uint32_t)i), c1 = fine_at((uint32_t)i + 64u);
            if (c0) atomicAdd(&hist[i], c0);
            if (c1) atomicAdd(&hist[i + 64], c1);
        }
    } else if (nbits != 15) {
        for (int i = t; i < nbins; i += kSelThreads) {
            const uint32_t c = s_hist[i];
            if (c) atomicAdd(&hist[i], c);
        }
    }
    STAMP(0, 4);
    if (!last) return;
    if (nbits != 15) {                                      // fp32, last digit (512 bins): the whole private histogram
        for (int i = t; i < nbins; i += kSelThreads) ws->windows[blockIdx.x][i] = s_hist[i];
        if (t == 0) ws->seg_win[blockIdx.x] = 0u;
        return;
    }
    __syncthreads();
    const int lo = (int)s_res[0];
    pub_store(&ws->windows[blockIdx.x][t], fine_at((uint32_t)(lo + t)));
    pub_store(&ws->windows[blockIdx.x][kSelThreads + t], fine_at((uint32_t)(lo + kSelThreads + t)));
    STAMP(0, 5);
}

// fp32, second (and last) digit of the single-device selection: the LOW 16 bits of the keys whose high 15 bits equal the prefix the
// first launch left (bits >> 16 == tau_hi: normally ~0.5 % of the elements).  65 536 counters fit the LDS only as 16-bit halves
// (128 KB): enough for what falls into one bin of the first digit, NOT for tie-heavy tensors -- so the workgroup proves that the
// sum of its bins equals the number of elements it counted, and when the proof fails (a half wrapped, or carried into its
// neighbour: either way the sum is off by a multiple of 65 535) it counts its segment again with 32-bit counters: the 512 coarse
// sums in one sweep, the 2048 fine bins of its window in a second.  Publish / ticket / resolve as in the first launch, 512 coarse bins.
template <bool FAST>
__global__ void __launch_bounds__(kSelThreads) k_select_hist_lo16(const void* in, int64_t numel, SelWs* ws)
{
    constexpr int DT = BFPQ_F32;
    constexpr int VEC = 4;
    constexpr int NCB = 2 * kCoarseBins;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];          // 32 768 words = 65 536 16-bit bins
    __shared__ __attribute__((aligned(16))) uint32_t s_coarse[NCB];
    __shared__ uint32_t s_res[4];
    __shared__ uint32_t s_part[16];
    const int t = threadIdx.x;
    for (int i = t; i < kFineBins / 4; i += kSelThreads) reinterpret_cast<uint4*>(s_hist)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const bfpq_select_state* st = &ws->st;
    const uint32_t pmask = st->prefix_mask, pval = st->prefix;                 // (0x7fff0000, tau_hi << 16)
    const uint32_t k_rem = (uint32_t)st->k_rem;
    const int64_t k = st->k, pop = st->ties;                                   // pop: elements that share the prefix, over the whole tensor
    const int64_t n_items = (numel + VEC - 1) / VEC;
    const SegGeom g = seg_geom(n_items);
    const int64_t i0 = (int64_t)blockIdx.x * g.L, i1 = i0 + g.L < n_items ? i0 + g.L : n_items;
    uint32_t matched = 0, zero_low = 0;
    {
        int64_t item = i0 + t;
        uint32_t cur[VEC], nxt[VEC];
        sweep_load<DT, FAST>(in, item, n_items, numel, cur);
        for (; item < i1; item += kSelThreads) {
            sweep_load<DT, FAST>(in, item + kSelThreads, n_items, numel, nxt);
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                const uint32_t key = mag_key<DT>(cur[j]);
                if ((FAST || item * VEC + j < numel) && (key & pmask) == pval) {
                    if (key & 0xffffu) atomicAdd(&s_hist[(key & 0xffffu) >> 1], (key & 1u) ? 0x10000u : 1u);
                    else zero_low++;                           // zeros of an already-sparse tensor, bf16 / HBFP values held in fp32: one word otherwise
                    matched++;
                }
            }
#pragma unroll
            for (int j = 0; j < VEC; j++) cur[j] = nxt[j];
        }
        if (zero_low) atomicAdd(&s_hist[0], zero_low);         // (a count beyond 16 bits carries into the neighbour half: the proof below fails and the segment is recounted, as before)
    }
    __syncthreads();
    coarse_from_lds<NCB, true>(s_hist, s_coarse);
    uint32_t total_matched;
    (void)block_excl_scan(matched, s_part, &total_matched);                    // (its barriers also complete s_coarse)
    uint32_t csum = 0;
    for (int i = t; i < NCB; i += kSelThreads) csum += s_coarse[i];
    uint32_t total_bins;
    (void)block_excl_scan(csum, s_part, &total_bins);
    const bool exact = total_bins == total_matched;                            // (block-uniform)
    const double frac = pop > 0 ? (double)k_rem / (double)pop : 0.0;
    // A segment holds only a few hundred of these keys, so its own quantile is a noisy guess of where the threshold's low bits lie
    // (sigma ~ 1500 of the 65 536 bins): a window around it misses too often.  The low 16 bits of real-valued data are spread almost
    // evenly, so a SECOND window sits where the threshold then is, frac * 65 536 -- the same place in every segment; the window around
    // the segment's own quantile stays for tensors whose low bits are anything but even (few distinct values: there all quantiles agree).
    int clo2 = (int)(frac * 65536.0) / 128 - kWinBins / 256;
    clo2 = clo2 < 0 ? 0 : (clo2 > NCB - kWinBins / 128 ? NCB - kWinBins / 128 : clo2);
    bool is_last;
    if (exact) {
        is_last = seg_publish_and_ticket<NCB>(s_coarse, s_res, ws, frac, [&](uint32_t b) { return lds_bin<true>(s_hist, b); }, clo2);
    } else {
        // a 16-bit counter overflowed: the segment again, with 32-bit counters -- coarse sums, then the window's fine bins
        uint32_t* s_win32 = s_hist;                                            // [2048] the window around the segment's own quantile, [2048] the second one
        __syncthreads();
        for (int i = t; i < NCB; i += kSelThreads) s_coarse[i] = 0;
        for (int i = t; i < 2 * kWinBins; i += kSelThreads) s_win32[i] = 0;
        __syncthreads();
        auto sweep = [&](auto&& f) __attribute__((always_inline)) {
            for (int64_t item = i0 + t; item < i1; item += kSelThreads) {
                uint32_t r[VEC];
                sweep_load<DT, FAST>(in, item, n_items, numel, r);
#pragma unroll
                for (int j = 0; j < VEC; j++) {
                    const uint32_t key = mag_key<DT>(r[j]);
                    if ((FAST || item * VEC + j < numel) && (key & pmask) == pval) f(key & 0xffffu);
                }
            }
        };
        sweep([&](uint32_t low) { atomicAdd(&s_coarse[low >> 7], 1u); });
        __syncthreads();
        // the window's position is decided inside seg_publish_and_ticket (s_res[0]); its fine bins are needed before it stores them:
        // decide the position here the same way (quantile of the coarse sums), count, then publish
        __shared__ uint32_t s_lo;
        if (t < 64) {
            constexpr int PER = NCB / 64;
            uint32_t c[PER], mine = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) { c[j] = s_coarse[t * PER + j]; mine += c[j]; }
            const uint32_t incl = wave_incl_scan(mine);
            const uint32_t seg_elems = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint64_t target = (uint64_t)((double)seg_elems * frac);
            if (seg_elems && target >= seg_elems) target = seg_elems - 1;
            const uint32_t excl = incl - mine;
            const bool hit = mine && excl <= target && target < (uint64_t)excl + mine;
            uint32_t e = excl;
            int A = t * PER;
#pragma unroll
            for (int j = 0; j < PER - 1; j++) { if (target >= (uint64_t)e + c[j]) { e += c[j]; A = t * PER + j + 1; } else break; }
            int clo = (int)pick_lane(hit, (uint32_t)A) - kWinBins / 256;
            clo = clo < 0 ? 0 : (clo > NCB - kWinBins / 128 ? NCB - kWinBins / 128 : clo);
            if (t == 0) s_lo = (uint32_t)clo * 128u;
        }
        __syncthreads();
        const uint32_t lo = s_lo;
        const uint32_t lo2 = (uint32_t)clo2 * 128u;
        sweep([&](uint32_t low) {
            if (low - lo < (uint32_t)kWinBins) atomicAdd(&s_win32[low - lo], 1u);
            if (low - lo2 < (uint32_t)kWinBins) atomicAdd(&s_win32[kWinBins + low - lo2], 1u);
        });
        __syncthreads();
        // (same coarse sums, same fraction: seg_publish_and_ticket places the first window where `lo` is; it asks for bins of either window)
        is_last = seg_publish_and_ticket<NCB>(s_coarse, s_res, ws, frac, [&](uint32_t b) {
            return b - lo < (uint32_t)kWinBins ? s_win32[b - lo] : s_win32[kWinBins + b - lo2]; }, clo2);
    }
    if (!is_last) return;
    fused_resolve<DT, FAST, NCB>(in, numel, n_items, g, ws, s_hist, k_rem, k, 0, pmask, pval, true, clo2);
}

// Launch 2 of the launch-pair form (multi-GPU: between them the all-gather of the histograms; fp32: three pairs): one
// workgroup (256 threads) per segment, every one of them repeats the (tiny) selection: which digit holds the k-th smallest
// key.  hist_all: n_hists histograms of BFPQ_SELECT_HIST_ENTRIES words -- BFPQ_SELECT_HIST_COPIES per rank, rank-major (an
// all-gather of the per-rank buffers; one rank: the local buffer) -- summed on the fly; on the deciding pass the per-rank
// counts of the threshold bin also give, without a second exchange, the ties that lower ranks hold, the windows give every
// segment's tie count, and their scan the cut segment.  Zeroes zero_buf (multi-GPU: the local histogram buffer, which this
// launch does not read -- it reads the gathered copy).
constexpr int kResThreads = 256;
template <int DT, bool FAST>
__global__ void __launch_bounds__(kResThreads) k_select_resolve(const void* in, int64_t numel, int pass, int shift, int nbits, int first, int last,
                                                                const uint32_t* hist_all, int n_ranks, int rank, int nc, int64_t rstride, int64_t k,
                                                                SelWs* ws, uint32_t* zero_buf, int own_hist)
{
    // hist_all: per rank `nc` histograms of BFPQ_SELECT_HIST_ENTRIES words, rank r's set at hist_all + r * rstride (the plain all-gather of the
    // local buffers: nc = BFPQ_SELECT_HIST_COPIES, rstride = nc * ENTRIES; the list exchange folds the copies first: nc = 1, and a rank's
    // row of the gathered [ranks, tensors, ENTRIES] block is `tensors * ENTRIES` words from the next rank's)
    constexpr int VEC = Traits<DT>::VEC;
    constexpr int NC = BFPQ_SELECT_HIST_COPIES;
    __shared__ uint32_t s_part[16];
    __shared__ uint32_t s_res[12];
    const int t = threadIdx.x;
    bfpq_select_state* st = &ws->st;
    STAMP(1, 0);
    if (own_hist) { hist_all = ws->hist[pass][0]; zero_buf = nullptr; n_ranks = 1; rank = 0; nc = NC; rstride = (int64_t)NC * BFPQ_SELECT_HIST_ENTRIES; }
    const int64_t n_items = (numel + VEC - 1) / VEC;
    const SegGeom g = seg_geom(n_items);
    // loads that depend on nothing go first: this segment-window word, the state of the previous pass
    const uint32_t wb = (last && t < g.G) ? ws->seg_win[t] : 0u;
    const uint32_t k_rem = first ? (uint32_t)k : (uint32_t)st->k_rem;
    const uint32_t prefix0 = first ? 0u : st->prefix, pmask0 = first ? 0u : st->prefix_mask;
    const int64_t k0 = first ? k : st->k;
    auto sum_h = [&](int idx) {
        uint32_t s = 0;
        for (int r = 0; r < n_ranks; r++)
            for (int c = 0; c < nc; c++) s += hist_all[(size_t)r * rstride + (size_t)c * BFPQ_SELECT_HIST_ENTRIES + idx];
        return s;
    };
    if (t < 12) s_res[t] = 0;                               // (k_rem == 0: digit 0, nothing in front of it)
    uint32_t total;
    if (nbits == 15) {
        uint32_t v = sum_h(kFineBins + t);                  // 256 coarse bins, one per thread
        uint32_t excl = block_excl_scan(v, s_part, &total);                  // (its barriers also order the s_res reset)
        if (v && excl < k_rem && k_rem <= excl + v) { s_res[0] = (uint32_t)t; s_res[1] = excl; }
        __syncthreads();
        const int C = (int)s_res[0];
        const uint32_t before = s_res[1];
        v = t < 128 ? sum_h(C * 128 + t) : 0u;
        excl = before + block_excl_scan(v, s_part, &total);
        if (v && excl < k_rem && k_rem <= excl + v) { s_res[2] = (uint32_t)(C * 128 + t); s_res[3] = excl; s_res[4] = v; }
    } else {
        const int nbins = 1 << nbits;                       // 2048 or 512: eight or two contiguous bins per thread
        const int per = nbins / kResThreads;
        uint32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mine = 0;
        for (int rc = 0; rc < n_ranks * nc; rc++) {         // a thread's bins are contiguous: 16-byte / 8-byte loads per copy
            const uint32_t* h = hist_all + (size_t)(rc / nc) * rstride + (size_t)(rc % nc) * BFPQ_SELECT_HIST_ENTRIES + t * per;
            if (per == 8) {
                const uint4 a4 = *reinterpret_cast<const uint4*>(h), b4 = *reinterpret_cast<const uint4*>(h + 4);
                v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w; v[4] += b4.x; v[5] += b4.y; v[6] += b4.z; v[7] += b4.w;
            } else {
                const uint2 a2 = *reinterpret_cast<const uint2*>(h);
                v[0] += a2.x; v[1] += a2.y;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) mine += v[j];
        uint32_t excl = block_excl_scan(mine, s_part, &total);
        if (mine && excl < k_rem && k_rem <= excl + mine) {
            int j = 0;
            while (j < per - 1 && k_rem > excl + v[j]) { excl += v[j]; j++; }
            s_res[2] = (uint32_t)(t * per + j); s_res[3] = excl; s_res[4] = v[j];
        }
    }
    __syncthreads();
    STAMP(1, 1);
    const uint32_t digit = s_res[2], run = s_res[3], cnt = s_res[4];
    const uint32_t prefix = prefix0 | (digit << shift);
    // the buffer the next histogram launch accumulates into
    if (zero_buf)
        for (int i = blockIdx.x * kResThreads + t; i < NC * BFPQ_SELECT_HIST_ENTRIES / 4; i += gridDim.x * kResThreads)
            reinterpret_cast<uint4*>(zero_buf)[i] = make_uint4(0, 0, 0, 0);
    if (!last) {
        if (blockIdx.x == 0 && t == 0) {
            st->prefix = prefix; st->prefix_mask = pmask0 | (((1u << nbits) - 1u) << shift);
            st->k_rem = (int64_t)(k_rem - run); st->k = k0; st->done = 0;
        }
        return;
    }
    const uint32_t tau = prefix, need = k_rem - run;
    uint32_t tie_base = 0;
    for (int r = 0; r < rank; r++)
        for (int c = 0; c < nc; c++) tie_base += hist_all[(size_t)r * rstride + (size_t)c * BFPQ_SELECT_HIST_ENTRIES + digit];
    const int W = nbits == 15 ? kWinBins : (1 << nbits);
    // tie count of every segment from its window
    uint32_t tc = 0;
    int miss = 0;
    if (t < g.G) {
        const uint32_t rel = digit - (wb & 0x3fffffffu);
        if (rel < (uint32_t)W) tc = ws->windows[t][rel];
        else if (wb >> 31) miss = 1;
    }
    STAMP(1, 2);
    const int any_miss = __syncthreads_or(miss);
    STAMP(1, 3);
    const bool mine_seg = t == (int)blockIdx.x;
    if (mine_seg) s_res[5] = (uint32_t)miss;
    __syncthreads();
    if (!any_miss) {
        cut_from_seg_ties(tc, (int64_t)need - (int64_t)tie_base, g, n_items, s_part, s_res + 8);     // (every workgroup: the same numbers)
    } else {
        // some segment's window does not cover the threshold: every such segment counts its ties again (its own
        // workgroup, so no workgroup waits for another); the apply launch then finds the cut from seg_ties
        if (s_res[5]) {
            const int64_t i0 = (int64_t)blockIdx.x * g.L, i1 = i0 + g.L < n_items ? i0 + g.L : n_items;
            uint32_t c = 0;
            for (int64_t it = i0 + t; it < i1; it += kResThreads) {
                uint32_t r[VEC];
                sweep_load<DT, FAST>(in, it, n_items, numel, r);
                c += count_eq<DT>(r, tau);
            }
            uint32_t tot;
            (void)block_excl_scan(c, s_part, &tot);
            if (t == 0) ws->seg_ties[blockIdx.x] = tot;
        } else if (mine_seg) ws->seg_ties[t] = tc;
    }
    STAMP(1, 4);
    if (blockIdx.x == 0 && t == 0) {
        st->prefix = prefix; st->prefix_mask = pmask0 | (((1u << nbits) - 1u) << shift);
        st->k_rem = (int64_t)need; st->tau = tau; st->done = 1;
        st->need = (int64_t)need; st->ties = (int64_t)cnt; st->k = k0;
        st->tie_base = (int64_t)tie_base; st->flags = (any_miss ? 0u : 1u) | (own_hist ? 2u : 0u);
        st->cut_lo = any_miss ? 0u : s_res[8]; st->cut_hi = any_miss ? 0u : s_res[9]; st->cut_within = any_miss ? 0u : s_res[10];
        st->cut_total = any_miss ? 0u : s_res[11]; st->reserved = 0;
    }
}

// Launch 3 (prune only: q -> s order, ragged shapes; the fused quantizer is k_fused_flat<.., -1, ..>).  The first kCutWGs
// workgroups own the cut segment, the others sweep the tensor and leave that segment's tiles alone.
template <int DT, bool FAST>
__global__ void __launch_bounds__(kThreads) k_threshold_apply(const void* in, void* out, int64_t numel, SelWs* ws)
{
    using raw_t = typename Traits<DT>::raw_t;
    constexpr int VEC = Traits<DT>::VEC;
    __shared__ uint32_t s_part[16];
    __shared__ uint32_t s_res[8];
    const int64_t n_items = (numel + VEC - 1) / VEC;
    const int64_t n_round = (n_items + 63) / 64 * 64;
    const bool cut_wg = (int)blockIdx.x < kCutWGs;
    const int64_t stride = (int64_t)(gridDim.x - kCutWGs) * kThreads;
    int64_t item = cut_wg ? (int64_t)threadIdx.x : (int64_t)((int)blockIdx.x - kCutWGs) * kThreads + threadIdx.x;
    uint32_t cur[VEC], nxt[VEC];
    sweep_load<DT, FAST>(in, item, n_items, numel, cur);
    ThrCtx t;
    thr_setup<DT>(t, ws, n_items, s_part, s_res);
    auto store = [&](int64_t it, const uint32_t* raw, uint32_t prune) __attribute__((always_inline)) {
        const int64_t e0 = it * VEC;
        if constexpr (FAST) {
            uint32_t r[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j++) r[j] = ((prune >> j) & 1u) ? 0u : raw[j];
            uint4 o;
            if constexpr (VEC == 4) o = make_uint4(r[0], r[1], r[2], r[3]);
            else o = make_uint4(r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16));
            reinterpret_cast<uint4*>(out)[it] = o;
        } else {
#pragma unroll
            for (int j = 0; j < VEC; j++)
                if (e0 + j < numel) reinterpret_cast<raw_t*>(out)[e0 + j] = ((prune >> j) & 1u) ? (raw_t)0 : (raw_t)raw[j];
        }
    };
    if (cut_wg) {
        cut_wg_run<DT, FAST>(t, in, numel, n_items, s_part, [&](int64_t it, const uint32_t* raw, bool valid, uint32_t before) __attribute__((always_inline)) {
            ThrCtx r = t;
            r.ranked = true; r.before = before;
            const uint32_t prune = thr_rank_bits<DT>(raw, valid, r);
            if (valid) store(it, raw, prune);
        });
        return;
    }
    const int lane = threadIdx.x & 63;
    for (; item < n_round; item += stride) {                   // wave-uniform trip count
        sweep_load<DT, FAST>(in, item + stride, n_items, numel, nxt);
        const bool valid = item < n_items;
        const int64_t tile0 = uniform64(item - lane);
        const bool in_cut = tile0 >= t.cut_lo && tile0 < t.cut_hi;      // (the cut workgroups' tiles)
        if (valid && !in_cut) {
            const uint32_t teff = t.tau + (tile0 < t.cut_lo ? 1u : 0u);
            uint32_t prune = 0;
#pragma unroll
            for (int j = 0; j < VEC; j++) prune |= (uint32_t)(mag_key<DT>(cur[j]) < teff) << j;
            store(item, cur, t.on ? prune : 0u);
        }
#pragma unroll
        for (int j = 0; j < VEC; j++) cur[j] = nxt[j];
    }
}

}  // namespace

extern "C" {

#ifdef BFPQ_STAMPS
int bfpq_debug_stamps(void* host_dst) { return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(bfpq_g_stamps), sizeof(bfpq_g_stamps)); }
#endif

int bfpq_select_passes(int dtype) { return dtype == BFPQ_F32 ? 3 : 1; }

int64_t bfpq_select_ws_bytes(void) { return (int64_t)sizeof(SelWs); }

// numel and k are counted in 32 bits on the device
static bool select_args_ok(const void* in, int64_t numel, int dtype, int pass, int64_t k, const void* ws)
{
    return ws && dtype >= 0 && dtype <= 2 && numel >= 0 && (in || numel == 0) && pass >= 0 && pass < bfpq_select_passes(dtype) && k >= 0;
}

static int launch_select_hist(const void* in, int64_t numel, int dtype, int pass, int64_t k, int64_t numel_global,
                              void* ws, uint32_t* hist, int fuse, void* stream)
{
    int shift, nbits;
    select_digit(dtype, pass, &shift, &nbits);
    int first = pass == 0, last = pass == bfpq_select_passes(dtype) - 1;
    if (fuse == 2) { shift = 16; nbits = 15; first = 1; last = 0; }      // fp32, single device: the high 15 bits of the key, resolved in the launch
    const size_t lds = sizeof(uint32_t) << nbits;
    const int vec = dtype_vec(dtype);
    const SegGeom g = seg_geom((numel + vec - 1) / vec);
    const bool fast = (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && numel % vec == 0;
    hipStream_t s = (hipStream_t)stream;
    SelWs* w = (SelWs*)ws;
#define BFPQ_SH(DT, F) do { \
        if (lds > 48 * 1024) { \
            const hipError_t err = hipFuncSetAttribute((const void*)k_select_hist<DT, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (err != hipSuccess) return (int)err; \
        } \
        hipLaunchKernelGGL((k_select_hist<DT, F>), dim3(g.G), dim3(kSelThreads), lds, s, in, numel, pass, shift, nbits, first, last, w, hist, k, numel_global, fuse); \
    } while (0)
    if (dtype == BFPQ_F32) { if (fast) BFPQ_SH(BFPQ_F32, true); else BFPQ_SH(BFPQ_F32, false); }
    else if (dtype == BFPQ_F16) { if (fast) BFPQ_SH(BFPQ_F16, true); else BFPQ_SH(BFPQ_F16, false); }
    else { if (fast) BFPQ_SH(BFPQ_BF16, true); else BFPQ_SH(BFPQ_BF16, false); }
#undef BFPQ_SH
    return (int)hipGetLastError();
}

int bfpq_select_hist(const void* in, int64_t numel, int dtype, int pass, int64_t k, int64_t numel_global,
                     void* ws, uint32_t* hist, void* stream)
{
    if (!select_args_ok(in, numel, dtype, pass, k, ws) || numel_global < numel || k > numel_global) return BFPQ_E_ARG;
    if (numel_global >= ((int64_t)1 << 32)) return BFPQ_E_UNSUPPORTED;
    if (numel == 0) return 0;
    return launch_select_hist(in, numel, dtype, pass, k, numel_global, ws, hist, 0, stream);
}

int bfpq_select_resolve_ex(const void* in, int64_t numel, int dtype, int pass, int64_t k,
                           const uint32_t* hist_all, int n_ranks, int rank, int copies, int64_t rank_stride_words,
                           void* ws, uint32_t* zero_hist, void* stream)
{
    if (!select_args_ok(in, numel, dtype, pass, k, ws) || (hist_all && (n_ranks < 1 || rank < 0 || rank >= n_ranks))) return BFPQ_E_ARG;
    if (hist_all && (copies < 1 || copies > BFPQ_SELECT_HIST_COPIES || rank_stride_words < (int64_t)copies * BFPQ_SELECT_HIST_ENTRIES)) return BFPQ_E_ARG;
    if (k >= ((int64_t)1 << 32) || numel >= ((int64_t)1 << 32)) return BFPQ_E_UNSUPPORTED;
    if (numel == 0) return 0;                                 // an empty slab has nothing to prune (its histogram stayed zero)
    int shift, nbits;
    select_digit(dtype, pass, &shift, &nbits);
    const int vec = dtype_vec(dtype);
    const SegGeom g = seg_geom((numel + vec - 1) / vec);
    const bool fast = (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && numel % vec == 0;
    const int first = pass == 0, last = pass == bfpq_select_passes(dtype) - 1;
    hipStream_t s = (hipStream_t)stream;
    SelWs* w = (SelWs*)ws;
#define BFPQ_SR(DT, F) hipLaunchKernelGGL((k_select_resolve<DT, F>), dim3(g.G), dim3(kResThreads), 0, s, in, numel, pass, shift, nbits, first, last, \
                                          hist_all, n_ranks, rank, copies, rank_stride_words, k, w, zero_hist, hist_all ? 0 : 1)
    if (dtype == BFPQ_F32) { if (fast) BFPQ_SR(BFPQ_F32, true); else BFPQ_SR(BFPQ_F32, false); }
    else if (dtype == BFPQ_F16) { if (fast) BFPQ_SR(BFPQ_F16, true); else BFPQ_SR(BFPQ_F16, false); }
    else { if (fast) BFPQ_SR(BFPQ_BF16, true); else BFPQ_SR(BFPQ_BF16, false); }
#undef BFPQ_SR
    return (int)hipGetLastError();
}

int bfpq_select_resolve(const void* in, int64_t numel, int dtype, int pass, int64_t k,
                        const uint32_t* hist_all, int n_ranks, int rank, void* ws, uint32_t* zero_hist, void* stream)
{
    return bfpq_select_resolve_ex(in, numel, dtype, pass, k, hist_all, n_ranks, rank, BFPQ_SELECT_HIST_COPIES,
                                  (int64_t)BFPQ_SELECT_HIST_COPIES * BFPQ_SELECT_HIST_ENTRIES, ws, zero_hist, stream);
}

// single device: all launches of the selection.  16-bit dtypes: ONE (the histogram launch, whose last workgroup resolves);
// fp32: three histogram / resolve pairs on the histogram buffers inside ws.
int bfpq_select(const void* in, int64_t numel, int dtype, int64_t k, void* ws, void* stream)
{
    if (!select_args_ok(in, numel, dtype, 0, k, ws) || k > numel) return BFPQ_E_ARG;
    if (numel >= ((int64_t)1 << 32)) return BFPQ_E_UNSUPPORTED;
    if (numel == 0) return 0;
    if (dtype != BFPQ_F32) return launch_select_hist(in, numel, dtype, 0, k, numel, ws, nullptr, 1, stream);
    // fp32: two digits, two launches, each ending in its own resolve step -- the high 15 bits of the key (the 16-bit dtypes' kernel on
    // bits >> 16), then the low 16 bits of the keys that share that prefix (16-bit LDS counters, see k_select_hist_lo16)
    int rc = launch_select_hist(in, numel, dtype, 0, k, numel, ws, nullptr, 2, stream);
    if (rc) return rc;
    const size_t lds = sizeof(uint32_t) * kFineBins;
    const SegGeom g = seg_geom((numel + 3) / 4);
    const bool fast = (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && numel % 4 == 0;
    hipStream_t s = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t err = hipFuncSetAttribute((const void*)k_select_hist_lo16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess) err = hipFuncSetAttribute((const void*)k_select_hist_lo16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return (int)err;
        attr_set = true;
    }
    if (fast) hipLaunchKernelGGL((k_select_hist_lo16<true>), dim3(g.G), dim3(kSelThreads), lds, s, in, numel, (SelWs*)ws);
    else hipLaunchKernelGGL((k_select_hist_lo16<false>), dim3(g.G), dim3(kSelThreads), lds, s, in, numel, (SelWs*)ws);
    return (int)hipGetLastError();
}

int bfpq_select_reset(void* ws, void* stream)
{
    if (!ws) return BFPQ_E_ARG;
    SelWs* w = (SelWs*)ws;                                     // ticket, coarse copies and the fine histograms: one contiguous block
    return (int)hipMemsetAsync(&w->ticket, 0, offsetof(SelWs, seg_ties) - offsetof(SelWs, ticket), (hipStream_t)stream);
}

int bfpq_threshold_apply(const void* in, void* out, int64_t numel, int dtype, void* ws, void* stream)
{
    if (!in || !out || !ws || dtype < 0 || dtype > 2 || numel < 0) return BFPQ_E_ARG;
    if (numel == 0) return 0;
    if (in == out) return BFPQ_E_ARG;                         // (tie ranks are counted from the input while other tiles are written)
    hipStream_t s = (hipStream_t)stream;
    SelWs* w = (SelWs*)ws;
    const int vec = dtype_vec(dtype);
    const bool fast = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0 && numel % vec == 0;
    const dim3 grid(kCutWGs + grid_for((numel + vec - 1) / vec)), block(kThreads);
#define BFPQ_TA(DT) do { if (fast) hipLaunchKernelGGL((k_threshold_apply<DT, true>), grid, block, 0, s, in, out, numel, w); \
                         else hipLaunchKernelGGL((k_threshold_apply<DT, false>), grid, block, 0, s, in, out, numel, w); } while (0)
    if (dtype == BFPQ_F32) BFPQ_TA(BFPQ_F32); else if (dtype == BFPQ_F16) BFPQ_TA(BFPQ_F16); else BFPQ_TA(BFPQ_BF16);
#undef BFPQ_TA
    return (int)hipGetLastError();
}

}  // extern "C"
