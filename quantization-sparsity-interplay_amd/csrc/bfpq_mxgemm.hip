// bfpq_mxgemm.hip -- prefill-sized consumer of the packed HBFP format: out[t][n] = sum_k x[t][k] W[n][k]
// for many tokens, with BOTH operands in HBFP (<= 5 bits: mantissas in [-15, 15]) and block 64.
//
// HBFP is a block-scaled format with integer elements and a power-of-two scale per block -- which is what CDNA4's block-scaled
// matrix instruction computes in hardware: v_mfma_scale_f32_32x32x64_f8f6f4 multiplies 32x64 by 64x32 elements and applies an
// E8M0 (2^(s-127)) scale per 32 elements of K to each operand.  The mantissas -15..15 are exact in OCP e4m3 (4 significant
// bits), K = 64 is exactly one HBFP block (both 32-halves carry the block's scale), so one instruction is the exact integer dot
// product of a block times 2^(ew-mw) * 2^(ex-mx), accumulated in fp32 across blocks: the same arithmetic as the decode kernel
// (bfpq_gemm.hip) at twice the bf16 matrix rate, with no per-block VALU work on the accumulators.  The reference runs F.linear
// on the fake-quantised tensors (bfp_ops.py:187-190); the two agree up to fp32 summation order across blocks.
//
// Operand images ("mx8"): e4m3 bytes [rows, K] row-major + E8M0 scale bytes [rows, K/64], made from the packed codes and
// exponents by k_mx8_from_codes (weights: once); activations get theirs in one pass from the tensor (bfpq_quantize_mx8).
//
// Lane maps of the instruction, measured with one-hot operands (tools_dev/mxprobe): first operand lane l = row l&31 of D,
// second operand lane l = column l&31; byte j of lane l of the first operand meets byte j of lane l of the second (so any k
// order works if both sides use the same one); bytes 0-15 of lanes 0-31 and of lanes 32-63 form the first 32-block (scale taken
// from lane row), bytes 16-31 the second (scale taken from lane row+32) -- with one HBFP block per instruction every lane simply
// supplies its row's scale.  D: column = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
//
// Two kernel forms live here.  k_mx8_gemm is the first one (kept as A/B variants 0-2): 128x128 output tile per 256-thread
// workgroup (4 waves as 2x2, 2x2 instruction tiles each), K step of 128 bytes staged global -> LDS by 16-byte LDS-DMA
// (global_load_lds_dwordx4), LDS rows of 128 bytes with the 16-byte slot XOR-swizzled by ((row >> 1) & 7) on the source address and
// on the read (conflict-free ds_read_b128), one or two LDS stages with a full drain at every hand-over.  ring_tile (further down) is
// what runs by default: larger tiles, a ring of 3-4 LDS stages with counted waits and raw barriers, two tile shapes in one launch,
// a split of K for short token counts.  In both, workgroup ids are remapped so that the workgroups of one XCD walk the token tiles
// of the same weight tile (the weight tile is fetched into that XCD's L2 once).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bfpq.h"
#include "bfpq_common.h"

using namespace bfpq;

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int V> struct template_int { static constexpr int value = V; };

constexpr int BK = 128;                                  // bytes of K per LDS stage: two HBFP blocks

// e4m3 byte of an integer of magnitude m <= 15 (exact): 0, 1 = 0x38, then 0x38 + 4m (2..3), 0x40 + 2m (4..7), 0x48 + m (8..15)
__device__ __forceinline__ uint32_t e4m3_of_mag(uint32_t m)
{
    const uint32_t v = m >= 8 ? 0x48u + m : (m >= 4 ? 0x40u + 2u * m : (m >= 2 ? 0x38u + 4u * m : 0x38u));
    return m ? v : 0u;
}
__device__ __forceinline__ uint32_t e4m3_of_int(int c) { const uint32_t m = (uint32_t)(c < 0 ? -c : c); return e4m3_of_mag(m) | (c < 0 ? 0x80u : 0u); }

// codes (4-bit two's complement nibbles, low nibble first, or int8) + int8 exponents -> e4m3 bytes + E8M0 scales
// one thread = 8 elements
template <int CODE_BITS>
__global__ void __launch_bounds__(256) k_mx8_from_codes(const uint8_t* __restrict__ codes, const int8_t* __restrict__ exps, uint8_t* __restrict__ out8,
                                                        uint8_t* __restrict__ outs, int64_t n_items, int64_t n_blocks, int mant_bits)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 256) {
        uint32_t lo = 0, hi = 0;
        if constexpr (CODE_BITS == 4) {
            const uint32_t w = reinterpret_cast<const uint32_t*>(codes)[i];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = (int)((w >> (4 * j)) & 15u);
                const uint32_t b = e4m3_of_int(c >= 8 ? c - 16 : c);
                if (j < 4) lo |= b << (8 * j); else hi |= b << (8 * (j - 4));
            }
        } else {
            const uint2 w = reinterpret_cast<const uint2*>(codes)[i];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                lo |= e4m3_of_int((int)(int8_t)(w.x >> (8 * j))) << (8 * j);
                hi |= e4m3_of_int((int)(int8_t)(w.y >> (8 * j))) << (8 * j);
            }
        }
        reinterpret_cast<uint2*>(out8)[i] = make_uint2(lo, hi);
        if (i < n_blocks) {                                 // the first n_blocks threads also translate the exponents
            const int e = exps[i];
            int s = e - mant_bits + 127;
            s = s < 0 ? 0 : (s > 254 ? 254 : s);
            outs[i] = e == -128 ? 0xffu : (uint8_t)s;       // -128 marks a NaN block: E8M0 0xff is NaN
        }
    }
}

__device__ __forceinline__ v8i read_frag(const uint8_t* tile, int row, int c)       // 32 bytes: 16-byte slots c, c+1 of the row
{
    const int sw = (row >> 1) & 7;
    const v4i a0 = *reinterpret_cast<const v4i*>(tile + row * BK + ((c ^ sw) << 4));
    const v4i a1 = *reinterpret_cast<const v4i*>(tile + row * BK + (((c + 1) ^ sw) << 4));
    return v8i{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
}

// one HBFP block (64 of K) of the wave's TM x TN instruction tiles; OP = which byte of the scale registers
template <int OP, int TM, int TN>
__device__ __forceinline__ void block_mfma(const uint8_t* sA, const uint8_t* sB, int b, int rowA, int rowB, int half, v16f (&acc)[TM][TN],
                                           const int (&sa)[TM], const int (&sb)[TN])
{
    const int c = b * 4 + half * 2;
    v8i aF[TM], bF[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) aF[i] = read_frag(sA, rowA + 32 * i, c);
#pragma unroll
    for (int j = 0; j < TN; j++) bF[j] = read_frag(sB, rowB + 32 * j, c);
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aF[i], bF[j], acc[i][j], 0, 0, OP, sa[i], OP, sb[j]);
}

// WM x WN waves, each TM x TN instruction tiles of 32 x 32: output tile BM x BN = (32 WM TM) x (32 WN TN).
// DBUF: two LDS stages; the next K step's LDS-DMA is issued before the current step's matrix work and waited for after it
// (one barrier per step).  The block scales are ordinary loads: they are fetched one trip (256 of K) ahead, in front of the
// DMA of that trip, so that their first use lies behind a wait that has drained the queue anyway.
template <int OUT_DT, int WM, int WN, int TM, int TN, bool DBUF>
__global__ void __launch_bounds__(64 * WM * WN) k_mx8_gemm(const uint8_t* __restrict__ x8, const uint8_t* __restrict__ xs, const uint8_t* __restrict__ w8,
                                                           const uint8_t* __restrict__ wsc, const void* __restrict__ bias, void* __restrict__ out,
                                                           int T, int N, int K, int tiles_t)
{
    constexpr int NW = WM * WN, BM = 32 * WM * TM, BN = 32 * WN * TN, STAGE = (BM + BN) * BK;
    constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;            // LDS-DMA instructions per wave and stage (8 rows = 1 KB each)
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
    __shared__ __attribute__((aligned(16))) uint8_t lds[(DBUF ? 2 : 1) * STAGE];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    // workgroups that share an XCD (id % 8) get consecutive tile numbers: token tiles fastest, so they share a weight tile
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int t0 = (wg % tiles_t) * BM, n0 = (wg / tiles_t) * BN;
    const int nb = K >> 6;                                           // blocks per row = scale bytes per row

    // staging: instruction i of wave w fills LDS rows (NW i + w) * 8 .. +7 (1 KB, lane-linear); lane l = row l>>3, slot l&7 holds
    // the row's 16-byte piece slot ^ ((row >> 1) & 7).  LDS rows are 128 bytes = half of the 64 banks, so the row's parity picks the
    // half and the swizzle must spread the 8 slots over the row PAIRS of a ds_read_b128 lane group ({0-3,12-15,20-27}, ...):
    // with (row & 7) rows 12 and 20 of a group met on the same banks (2-way conflict on every fragment read)
    size_t offA[GA], offB[GB];
#pragma unroll
    for (int i = 0; i < GA; i++) {
        const int row = (NW * i + w) * 8 + (l >> 3), piece = ((l & 7) ^ ((row >> 1) & 7)) << 4;
        const int ta = t0 + row < T ? t0 + row : T - 1;                                          // rows past the edge repeat the last row
        offA[i] = (size_t)ta * K + piece;
    }
#pragma unroll
    for (int i = 0; i < GB; i++) {
        const int row = (NW * i + w) * 8 + (l >> 3), piece = ((l & 7) ^ ((row >> 1) & 7)) << 4;
        const int na = n0 + row < N ? n0 + row : N - 1;
        offB[i] = (size_t)na * K + piece;
    }
    const int wr = w / WN, wc = w % WN, half = l >> 5;
    const int rowA = wr * (32 * TM) + (l & 31), rowB = wc * (32 * TN) + (l & 31);
    uint32_t sxo[TM], swo[TN];                                       // dword index of the row's first scale
#pragma unroll
    for (int i = 0; i < TM; i++) { const int ta = t0 + rowA + 32 * i < T ? t0 + rowA + 32 * i : T - 1; sxo[i] = (uint32_t)ta * (uint32_t)(nb >> 2); }
#pragma unroll
    for (int j = 0; j < TN; j++) { const int na = n0 + rowB + 32 * j < N ? n0 + rowB + 32 * j : N - 1; swo[j] = (uint32_t)na * (uint32_t)(nb >> 2); }
    const uint32_t* const xs32 = reinterpret_cast<const uint32_t*>(xs);
    const uint32_t* const ws32 = reinterpret_cast<const uint32_t*>(wsc);
    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    auto stage = [&](int k0, int buf) __attribute__((always_inline)) {
        uint8_t* const sA = lds + buf * STAGE;
        uint8_t* const sB = sA + BM * BK;
#pragma unroll
        for (int i = 0; i < GA; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x8 + offA[i] + k0),
                                             (__attribute__((address_space(3))) void*)(sA + (NW * i + w) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < GB; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w8 + offB[i] + k0),
                                             (__attribute__((address_space(3))) void*)(sB + (NW * i + w) * 1024), 16, 0, 0);
    };
    auto landed = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); };

    int sa[TM], sb[TN];
    if constexpr (DBUF) {
        const uint8_t* const sA0 = lds;
        const uint8_t* const sB0 = lds + BM * BK;
        const uint8_t* const sA1 = lds + STAGE;
        const uint8_t* const sB1 = lds + STAGE + BM * BK;
#pragma unroll
        for (int i = 0; i < TM; i++) sa[i] = (int)xs32[sxo[i]];
#pragma unroll
        for (int j = 0; j < TN; j++) sb[j] = (int)ws32[swo[j]];
        stage(0, 0);
        landed();
        const int trips = K >> 8;
        for (int tr = 0; tr < trips; tr++) {
            const int k0 = tr << 8;
            const int trn = tr + 1 < trips ? tr + 1 : tr;                 // (last trip: re-reads its own scales, re-stages its own last step)
            int san[TM], sbn[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) san[i] = (int)xs32[sxo[i] + trn];
#pragma unroll
            for (int j = 0; j < TN; j++) sbn[j] = (int)ws32[swo[j] + trn];
            stage(k0 + BK, 1);
            block_mfma<0, TM, TN>(sA0, sB0, 0, rowA, rowB, half, acc, sa, sb);
            block_mfma<1, TM, TN>(sA0, sB0, 1, rowA, rowB, half, acc, sa, sb);
            landed();
            stage(trn << 8, 0);
            block_mfma<2, TM, TN>(sA1, sB1, 0, rowA, rowB, half, acc, sa, sb);
            block_mfma<3, TM, TN>(sA1, sB1, 1, rowA, rowB, half, acc, sa, sb);
            landed();
#pragma unroll
            for (int i = 0; i < TM; i++) sa[i] = san[i];
#pragma unroll
            for (int j = 0; j < TN; j++) sb[j] = sbn[j];
        }
    } else {
        const uint8_t* const sA = lds;
        const uint8_t* const sB = lds + BM * BK;
        for (int k0 = 0; k0 < K; k0 += 2 * BK) {                         // 4 blocks per trip: one scale dword per operand row
#pragma unroll
            for (int i = 0; i < TM; i++) sa[i] = (int)xs32[sxo[i] + (k0 >> 8)];
#pragma unroll
            for (int j = 0; j < TN; j++) sb[j] = (int)ws32[swo[j] + (k0 >> 8)];
            stage(k0, 0);
            landed();
            block_mfma<0, TM, TN>(sA, sB, 0, rowA, rowB, half, acc, sa, sb);
            block_mfma<1, TM, TN>(sA, sB, 1, rowA, rowB, half, acc, sa, sb);
            __syncthreads();
            stage(k0 + BK, 0);
            landed();
            block_mfma<2, TM, TN>(sA, sB, 0, rowA, rowB, half, acc, sa, sb);
            block_mfma<3, TM, TN>(sA, sB, 1, rowA, rowB, half, acc, sa, sb);
            __syncthreads();
        }
    }

    // epilogue: D column = lane & 31 (n), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (t)
    using raw_t = typename Traits<OUT_DT>::raw_t;
    raw_t* const o = reinterpret_cast<raw_t*>(out);
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wc * (32 * TN) + 32 * j + (l & 31);
        if (n >= N) continue;
        const float bv = bias ? raw_to_f32<OUT_DT>((uint32_t)reinterpret_cast<const raw_t*>(bias)[n]) : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int t = t0 + wr * (32 * TM) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (t < T) o[(size_t)t * N + n] = (raw_t)f32_to_raw<OUT_DT>(acc[i][j][e] + bv);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The pipelined form.  The counters of the kernel above say it is bound by the rate at which operand bytes reach the CU
// (both 128 x 128 structures settle at ~27 B/clk/CU = ~14 TB/s over the chip, whatever their overlap), and that the 256 x 256
// form, which needs half the bytes per flop, is bound by the LATENCY of a two-stage hand-over instead.  So: large tiles AND a
// deep ring.  One LDS stage = ONE HBFP block (64 bytes of K) of every tile row, four stages in the ring, the DMA of step s+3
// issued before the matrix work of step s; waits are COUNTED (vmcnt leaves the two youngest stages in flight) and the barrier is
// the raw s_barrier -- __syncthreads() would drain the DMA queue.  Nothing but LDS-DMA touches global memory in the loop (the
// block scales travel as one dword per tile row and trip of four blocks into their own little LDS ring), so every wait in it is
// one written here; all LDS is one array.
//   LDS rows are 64 bytes = 4 slots of 16; slot' = slot ^ ((row >> 2) & 3) on the DMA source and on the read spreads the 16 rows of
//   a ds_read_b128 lane group over all 64 banks.
// ---------------------------------------------------------------------------------------------------------------------
// 32 bytes of block b (of KB per LDS row), lane half `half`: slots 4 b + 2 half, + 1 of a row of 4 KB slots
template <int KB>
__device__ __forceinline__ v8i read_frag_ring(const uint8_t* tile, int row, int b, int half)
{
    const int sw = KB == 1 ? (row >> 2) & 3 : (row >> 1) & 7;
    const int c = 4 * b + 2 * half;
    const v4i a0 = *reinterpret_cast<const v4i*>(tile + row * (64 * KB) + ((c ^ sw) << 4));
    const v4i a1 = *reinterpret_cast<const v4i*>(tile + row * (64 * KB) + (((c + 1) ^ sw) << 4));
    return v8i{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KB = HBFP blocks (64 B of K) per ring stage, NS = stages in the ring (the DMA runs NS - 1 stages ahead)
template <int WM, int WN, int TM, int TN, int KB, int NS> struct RingCfg {
    static constexpr int NW = WM * WN, NT = 64 * NW, BM = 32 * WM * TM, BN = 32 * WN * TN, STAGE = (BM + BN) * 64 * KB;
    static constexpr int LDS_BYTES = NS * STAGE + 2 * NT * 4;
};

// one output tile: `orig` of `nwg` workgroups of this tile shape (ids remapped per XCD), columns starting at n_base
template <int OUT_DT, int WM, int WN, int TM, int TN, int KB, int NS>
__device__ __forceinline__ void ring_tile(uint8_t* lds, int orig, int nwg, int n_base,
                                          const uint8_t* __restrict__ x8, const uint8_t* __restrict__ xs, const uint8_t* __restrict__ w8,
                                          const uint8_t* __restrict__ wsc, const void* __restrict__ bias, void* __restrict__ out,
                                          int T, int N, int K, int tiles_t, int tr0 = 0, int tr1 = -1)
{
    using C = RingCfg<WM, WN, TM, TN, KB, NS>;
    constexpr int NW = C::NW, NT = C::NT, BM = C::BM, BN = C::BN, STAGE = C::STAGE;
    constexpr int RPP = 16 / KB, SLOTS = 4 * KB;                               // tile rows per DMA piece (1 KB), 16-byte slots per row
    constexpr int GA = BM / RPP / NW, GB = BN / RPP / NW, G = GA + GB;         // DMA pieces per wave and stage
    constexpr int D = NS - 1, SPT = 4 / KB;                                    // stages the DMA runs ahead; stages per trip of 4 blocks (one scale dword)
    static_assert(KB == 1 || KB == 2, "one or two blocks per stage");
    static_assert(D >= 1 && D <= SPT, "the next trip's scales must have landed when the trip begins");
    static_assert(BM % (RPP * NW) == 0 && BN % (RPP * NW) == 0, "tile rows must split evenly over the waves");
    static_assert(NT >= BM + BN, "one thread per tile row carries the row's scales");
    uint8_t* const sscale = lds + NS * STAGE;                                  // [2][NT] dwords: scales of the 4 blocks of a trip, per tile row
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int t0 = (wg % tiles_t) * BM, n0 = n_base + (wg / tiles_t) * BN;
    const int nb = K >> 6;
    const int trips = tr1 < 0 ? nb >> 2 : tr1;                                  // this workgroup's trips of K: [tr0, trips) (all of K unless K is split)
    const int S = trips * (4 / KB);                                            // one past its last stage

    auto swz = [](int row) { return KB == 1 ? (row >> 2) & 3 : (row >> 1) & 7; };
    size_t offA[GA], offB[GB];
#pragma unroll
    for (int i = 0; i < GA; i++) {
        const int row = (NW * i + w) * RPP + l / SLOTS, piece = ((l % SLOTS) ^ swz(row)) << 4;
        const int ta = t0 + row < T ? t0 + row : T - 1;
        offA[i] = (size_t)ta * K + piece;
    }
#pragma unroll
    for (int i = 0; i < GB; i++) {
        const int row = (NW * i + w) * RPP + l / SLOTS, piece = ((l % SLOTS) ^ swz(row)) << 4;
        const int na = n0 + row < N ? n0 + row : N - 1;
        offB[i] = (size_t)na * K + piece;
    }
    // this thread's tile row for the scale DMA: rows 0 .. BM-1 = tokens, BM .. BM+BN-1 = output features (threads past that repeat the last)
    const uint32_t* srow;
    {
        const int tr = (int)threadIdx.x < BM + BN ? (int)threadIdx.x : BM + BN - 1;
        if (tr < BM) { const int ta = t0 + tr < T ? t0 + tr : T - 1; srow = reinterpret_cast<const uint32_t*>(xs) + (size_t)ta * (nb >> 2); }
        else { const int na = n0 + tr - BM < N ? n0 + tr - BM : N - 1; srow = reinterpret_cast<const uint32_t*>(wsc) + (size_t)na * (nb >> 2); }
    }
    const int wr = w / WN, wc = w % WN, half = l >> 5;
    const int rowA = wr * (32 * TM) + (l & 31), rowB = wc * (32 * TN) + (l & 31);
    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    auto stage = [&](int s, int buf) __attribute__((always_inline)) {           // stage s of K (KB blocks) -> ring slot buf
        uint8_t* const sA = lds + buf * STAGE;
        uint8_t* const sB = sA + BM * 64 * KB;
        const int k0 = s * (64 * KB);
#pragma unroll
        for (int i = 0; i < GA; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x8 + offA[i] + k0),
                                             (__attribute__((address_space(3))) void*)(sA + (NW * i + w) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < GB; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w8 + offB[i] + k0),
                                             (__attribute__((address_space(3))) void*)(sB + (NW * i + w) * 1024), 16, 0, 0);
    };
    auto stage_scales = [&](int trip) __attribute__((always_inline)) {          // one dword per tile row -> sscale[trip & 1][thread]
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srow + trip),
                                         (__attribute__((address_space(3))) void*)(sscale + (trip & 1) * NT * 4 + w * 256), 4, 0, 0);
    };
    int sa[TM], sb[TN];
    auto mm = [&](auto op_tag, int buf, int b) __attribute__((always_inline)) {
        constexpr int OP = decltype(op_tag)::value;
        const uint8_t* const sA = lds + buf * STAGE;
        const uint8_t* const sB = sA + BM * 64 * KB;
        v8i aF[TM], bF[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) aF[i] = read_frag_ring<KB>(sA, rowA + 32 * i, b, half);
#pragma unroll
        for (int j = 0; j < TN; j++) bF[j] = read_frag_ring<KB>(sB, rowB + 32 * j, b, half);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aF[i], bF[j], acc[i][j], 0, 0, OP, sa[i], OP, sb[j]);
    };
    // step j of a trip, working on stage s in ring slot `slot`: [j == 0: the next trip's scales,] the DMA of stage s + D, the matrix work of
    // stage s, then the hand-over: stage s + 1 must have landed, i.e. all but the (D - 1) G pieces of stages s+2 .. s+D (and the scale dword while
    // it still sits behind stage s + 1 in the queue: j <= D - 2) may stay in flight; then the raw barrier.
    auto step = [&](auto j_tag, int s, int& slot, int trn) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value;
        if constexpr (J == 0) stage_scales(trn);
        int pslot = slot + D; if (pslot >= NS) pslot -= NS;
        stage(s + D < S ? s + D : S - 1, pslot);
        mm(template_int<J * KB>{}, slot, 0);
        if constexpr (KB == 2) mm(template_int<J * KB + 1>{}, slot, 1);
        wait_vm<(D - 1) * G + (J <= D - 2 ? 1 : 0)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot + 1 == NS ? 0 : slot + 1;
    };

    // prologue: scales of the first trip and its stages 0 .. D-1, drained once
    stage_scales(tr0);
#pragma unroll
    for (int d = 0; d < D; d++) stage(tr0 * SPT + d, d);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int tr = tr0; tr < trips; tr++) {
        const int s = tr * SPT;
        const int trn = tr + 1 < trips ? tr + 1 : tr;
        const uint32_t* const sc = reinterpret_cast<const uint32_t*>(sscale + (tr & 1) * NT * 4);
#pragma unroll
        for (int i = 0; i < TM; i++) sa[i] = (int)sc[rowA + 32 * i];
#pragma unroll
        for (int j = 0; j < TN; j++) sb[j] = (int)sc[BM + rowB + 32 * j];
        step(template_int<0>{}, s, slot, trn);
        if constexpr (SPT > 1) step(template_int<1>{}, s + 1, slot, trn);
        if constexpr (SPT > 2) { step(template_int<2>{}, s + 2, slot, trn); step(template_int<3>{}, s + 3, slot, trn); }
    }
    wait_vm<0>();                                                               // (the clamped re-stages of the tail)

    using raw_t = typename Traits<OUT_DT>::raw_t;
    raw_t* const o = reinterpret_cast<raw_t*>(out);
    constexpr int ES = (int)sizeof(raw_t), EPP = 16 / ES;                       // bytes per element, elements per 16-byte piece
    const bool vec_ok = (N % EPP) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    if (vec_ok) {
        // D has the output column on the lane and the rows in the registers: stored as it lies, every store instruction writes 2-byte
        // elements in 64-byte runs (128 store instructions per wave for the 256 x 256 tile).  Each 32 x 32 instruction tile goes through
        // a wave-private LDS patch instead (the ring is free by now) and leaves as 16-byte pieces, 64 contiguous bytes per row.
        __builtin_amdgcn_s_barrier();                                           // every wave's DMA has landed: the ring may be overwritten
        constexpr int RS = 32 * ES + 16;                                        // patch row stride: +16 bytes keeps lanes l and l+32 (rows r, r+4) on different banks
        uint8_t* const ep = lds + w * (32 * RS);
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nl = n0 + wc * (32 * TN) + 32 * j + (l & 31);
            const float bv = (bias && nl < N) ? raw_to_f32<OUT_DT>((uint32_t)reinterpret_cast<const raw_t*>(bias)[nl]) : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; i++) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * half;
                    *reinterpret_cast<raw_t*>(ep + rr * RS + (l & 31) * ES) = (raw_t)f32_to_raw<OUT_DT>(acc[i][j][e] + bv);
                }
#pragma unroll
                for (int qq = 0; qq < ES; qq++) {                               // 64 ES pieces of 16 bytes per tile: ES per lane
                    const int pc = l + 64 * qq, rr = pc / (2 * ES), cp = pc % (2 * ES);
                    const uint4 v = *reinterpret_cast<const uint4*>(ep + rr * RS + cp * 16);
                    const int t = t0 + wr * (32 * TM) + 32 * i + rr, n = n0 + wc * (32 * TN) + 32 * j + cp * EPP;
                    if (t < T && n < N) *reinterpret_cast<uint4*>(o + (size_t)t * N + n) = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wc * (32 * TN) + 32 * j + (l & 31);
        if (n >= N) continue;
        const float bv = bias ? raw_to_f32<OUT_DT>((uint32_t)reinterpret_cast<const raw_t*>(bias)[n]) : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int t = t0 + wr * (32 * TM) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (t < T) o[(size_t)t * N + n] = (raw_t)f32_to_raw<OUT_DT>(acc[i][j][e] + bv);
            }
    }
}

template <int OUT_DT, int WM, int WN, int TM, int TN, int KB, int NS>
__global__ void __launch_bounds__(64 * WM * WN) k_mx8_gemm_ring(const uint8_t* __restrict__ x8, const uint8_t* __restrict__ xs, const uint8_t* __restrict__ w8,
                                                                const uint8_t* __restrict__ wsc, const void* __restrict__ bias, void* __restrict__ out,
                                                                int T, int N, int K, int tiles_t)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[RingCfg<WM, WN, TM, TN, KB, NS>::LDS_BYTES];
    ring_tile<OUT_DT, WM, WN, TM, TN, KB, NS>(lds, (int)blockIdx.x, (int)gridDim.x, 0, x8, xs, w8, wsc, bias, out, T, N, K, tiles_t);
}

// K split over `splits` workgroups per output tile (short token counts: too few tiles for the chip, e.g. 128 tokens x down_proj = 32 tiles
// over K = 11008): workgroup (tile, part) accumulates its trips of K into the fp32 slab `part`; k_mx8_splitk_reduce adds the slabs in part
// order (deterministic), the bias, and rounds to the output dtype.
template <int WM, int WN, int TM, int TN, int KB, int NS>
__global__ void __launch_bounds__(64 * WM * WN) k_mx8_gemm_ring_splitk(const uint8_t* __restrict__ x8, const uint8_t* __restrict__ xs, const uint8_t* __restrict__ w8,
                                                                       const uint8_t* __restrict__ wsc, float* __restrict__ slabs,
                                                                       int T, int N, int K, int tiles_t, int tiles, int trips_per_part)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[RingCfg<WM, WN, TM, TN, KB, NS>::LDS_BYTES];
    const int part = (int)blockIdx.x / tiles, tile = (int)blockIdx.x % tiles;
    const int trips = K >> 8, a = part * trips_per_part, b = a + trips_per_part < trips ? a + trips_per_part : trips;
    ring_tile<BFPQ_F32, WM, WN, TM, TN, KB, NS>(lds, tile, tiles, 0, x8, xs, w8, wsc, nullptr, slabs + (size_t)part * T * N, T, N, K, tiles_t, a, b);
}

template <int OUT_DT>
__global__ void __launch_bounds__(256) k_mx8_splitk_reduce(const float* __restrict__ slabs, const void* __restrict__ bias, void* __restrict__ out, int64_t TN_, int N, int parts)
{
    using raw_t = typename Traits<OUT_DT>::raw_t;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < TN_; i += (int64_t)gridDim.x * 256) {
        float acc = slabs[i];
        for (int p = 1; p < parts; p++) acc += slabs[(size_t)p * TN_ + i];
        if (bias) acc += raw_to_f32<OUT_DT>((uint32_t)reinterpret_cast<const raw_t*>(bias)[i % N]);
        reinterpret_cast<raw_t*>(out)[i] = (raw_t)f32_to_raw<OUT_DT>(acc);
    }
}

// Two tile shapes in one launch.  256 x 256 tiles need half the operand bytes per flop, but N / 256 x T / 256 of them rarely fill
// the 256 CUs a whole number of times (2048 tokens x 11008 features: 344 tiles = one full round and a third of a second one).  So
// the first n_big workgroups (whole rounds of the chip) take 256 x 256 tiles of the first columns, the remaining columns go in
// 256 x 128 tiles, which are dispatched last and fill the CUs as the big tiles retire.
template <int OUT_DT, int SKB, int SNS>
__global__ void __launch_bounds__(512) k_mx8_gemm_ring_mixed(const uint8_t* __restrict__ x8, const uint8_t* __restrict__ xs, const uint8_t* __restrict__ w8,
                                                             const uint8_t* __restrict__ wsc, const void* __restrict__ bias, void* __restrict__ out,
                                                             int T, int N, int K, int tiles_t, int n_big, int n_base_small)
{
    constexpr int LB = RingCfg<2, 4, 4, 2, 1, 4>::LDS_BYTES, LS = RingCfg<4, 2, 2, 2, SKB, SNS>::LDS_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t lds[LB > LS ? LB : LS];
    if ((int)blockIdx.x < n_big)
        ring_tile<OUT_DT, 2, 4, 4, 2, 1, 4>(lds, (int)blockIdx.x, n_big, 0, x8, xs, w8, wsc, bias, out, T, N, K, tiles_t);
    else
        ring_tile<OUT_DT, 4, 2, 2, 2, SKB, SNS>(lds, (int)blockIdx.x - n_big, (int)gridDim.x - n_big, n_base_small, x8, xs, w8, wsc, bias, out, T, N, K, tiles_t);
}

// Tile variants (bfpq_tune key BFPQ_TUNE_MX8_VARIANT; measured with tools_dev/ab_mx8.py, interleaved in one process, all bit-identical).
// gate_proj [2048 x 4096 x 11008] / 8192 tokens / q_proj [2048 x 4096 x 4096] / down_proj [2048 x 11008 x 4096], us:
//   0: 128 x 128, 4 waves, one LDS stage of 128 B of K, ~3 workgroups per CU                      119 / 469 / 49 / 108
//   1: 128 x 128, two stages (next step's DMA under the matrix work), 2 workgroups per CU           127 / 536 / 47 / 109
//   2: 256 x 256, 8 waves, two stages, 1 workgroup per CU                                           145 / 446 / 75 / 161
//   3: 256 x 256, 8 waves, ring of four stages of 64 B of K, counted waits                         109 / 360 / 55 / 129   (default when its tiles fill whole rounds)
//   4: 256 x 128, 8 waves, ring of three stages of 128 B of K                                        95 / 487 / 38 /  86
//   5: 3 for whole rounds of the chip + 4 for the remaining columns, one launch                      90 / 389 / 37 /  86   (default otherwise)
//   6: 128 x 128, 4 waves, ring of three stages of 128 B of K (default up to 256 tokens: 128 tokens x gate_proj 23 us, variant 0: 39, variant 4: 30;
//      128 tokens x down_proj -- 32 tiles, K = 11008 -- 50 us unsplit, 22 us with K split over 8 workgroups per tile: bfpq_hbfp_linear_mx8_splitk)
//   (the ring's epilogue through a wave-private LDS patch, 16-byte stores: +1-3 % over lane-by-lane 2-byte stores; s_setprio(1) around the matrix
//    instructions of a stage: -5...15 %, removed; a 256 x 128 ring with 64-byte stages -- four matrix instructions per barrier -- ran 113 / 515 / 45 / 106; 256 x 128 and 128 x 256 in the
//    two-stage form 10-25 % behind variant 0; weight fragments loaded straight into registers, 32 rows x 64 B per load instruction, 1.5x slower: removed)
struct MxCfg { int bm, bn; };
constexpr MxCfg kMxCfg[] = {{128, 128}, {128, 128}, {256, 256}, {256, 256}, {256, 128}, {256, 256}, {128, 128}};
constexpr int kMxVariants = 7;

template <int OUT_DT>
int launch_mx8(int variant, const uint8_t* a, const uint8_t* as, const uint8_t* b, const uint8_t* bs, const void* bias, void* out,
               int T, int N, int K, hipStream_t s)
{
    const int bm = kMxCfg[variant].bm, bn = kMxCfg[variant].bn;
    const int tiles_t = (T + bm - 1) / bm, tiles_n = (N + bn - 1) / bn;
    const dim3 grid((unsigned)(tiles_t * tiles_n));
    if (variant == 5) {
        // whole rounds of the chip in 256 x 256 tiles, the rest of the columns in 256 x 128 tiles (kCUs: the part has 256)
        constexpr int kCUs = 256;
        const int rounds = tiles_t * tiles_n / kCUs;
        int big_cols = rounds > 0 ? rounds * kCUs / tiles_t : 0;               // column tiles of 256 that go as big tiles
        if (big_cols > tiles_n) big_cols = tiles_n;
        const int n_big = big_cols * tiles_t, n_base = big_cols * 256;
        const int small_cols = n_base < N ? (N - n_base + 127) / 128 : 0;
        const dim3 g2((unsigned)(n_big + small_cols * tiles_t));
        hipLaunchKernelGGL((k_mx8_gemm_ring_mixed<OUT_DT, 2, 3>), g2, dim3(512), 0, s, a, as, b, bs, bias, out, T, N, K, tiles_t, n_big, n_base);
        return (int)hipGetLastError();
    }
    switch (variant) {
        case 0: hipLaunchKernelGGL((k_mx8_gemm<OUT_DT, 2, 2, 2, 2, false>), grid, dim3(256), 0, s, a, as, b, bs, bias, out, T, N, K, tiles_t); break;
        case 1: hipLaunchKernelGGL((k_mx8_gemm<OUT_DT, 2, 2, 2, 2, true>), grid, dim3(256), 0, s, a, as, b, bs, bias, out, T, N, K, tiles_t); break;
        case 2: hipLaunchKernelGGL((k_mx8_gemm<OUT_DT, 2, 4, 4, 2, true>), grid, dim3(512), 0, s, a, as, b, bs, bias, out, T, N, K, tiles_t); break;
        case 3: hipLaunchKernelGGL((k_mx8_gemm_ring<OUT_DT, 2, 4, 4, 2, 1, 4>), grid, dim3(512), 0, s, a, as, b, bs, bias, out, T, N, K, tiles_t); break;
        case 6: hipLaunchKernelGGL((k_mx8_gemm_ring<OUT_DT, 2, 2, 2, 2, 2, 3>), grid, dim3(256), 0, s, a, as, b, bs, bias, out, T, N, K, tiles_t); break;
        default: hipLaunchKernelGGL((k_mx8_gemm_ring<OUT_DT, 4, 2, 2, 2, 2, 3>), grid, dim3(512), 0, s, a, as, b, bs, bias, out, T, N, K, tiles_t); break;
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int bfpq_mx8_from_hbfp(const void* codes, const int8_t* exps, void* out8, void* out_scale, int64_t rows, int64_t cols,
                       int code_bits, int mant_bits, void* stream)
{
    if (rows < 0 || cols < 0 || (code_bits != 4 && code_bits != 8) || mant_bits < 1 || mant_bits > 4 || cols % 64 != 0) return BFPQ_E_ARG;
    if (code_bits == 4 && mant_bits > 3) return BFPQ_E_ARG;
    if (rows * cols == 0) return 0;
    if (!codes || !exps || !out8 || !out_scale) return BFPQ_E_ARG;
    const int64_t n_items = rows * cols / 8, n_blocks = rows * cols / 64;
    int64_t g = (n_items + 255) / 256;
    if (g > 8192) g = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (code_bits == 4) hipLaunchKernelGGL((k_mx8_from_codes<4>), dim3((unsigned)g), dim3(256), 0, s, (const uint8_t*)codes, exps, (uint8_t*)out8, (uint8_t*)out_scale, n_items, n_blocks, mant_bits);
    else hipLaunchKernelGGL((k_mx8_from_codes<8>), dim3((unsigned)g), dim3(256), 0, s, (const uint8_t*)codes, exps, (uint8_t*)out8, (uint8_t*)out_scale, n_items, n_blocks, mant_bits);
    return (int)hipGetLastError();
}

int bfpq_hbfp_linear_mx8_ok(int64_t T, int64_t N, int64_t K)
{
    return T >= 1 && N >= 1 && K >= 256 && K % 256 == 0 && T * (K / 256) < ((int64_t)1 << 32) && N * (K / 256) < ((int64_t)1 << 32) &&
           ((T + 127) / 128) * ((N + 127) / 128) < ((int64_t)1 << 30);
}

// the 128 x 128 tile (and, through bfpq_hbfp_linear_mx8_parts, a split of K) whenever the larger tiles would leave half the chip idle
static bool mx8_short(int64_t T, int64_t N) { return T <= 256 || ((T + 255) / 256) * ((N + 127) / 128) < 128; }

__attribute__((visibility("hidden"))) int bfpq_g_mx8_variant = -1;             // -1 = choose; 0..6 force (bfpq_tune, BFPQ_TUNE_MX8_VARIANT)

int bfpq_hbfp_linear_mx8(const void* x8, const void* xs, const void* w8, const void* ws, const void* bias, void* out,
                         int64_t T, int64_t N, int64_t K, int out_dtype, void* stream)
{
    if (out_dtype < 0 || out_dtype > 2 || T < 0 || N < 0) return BFPQ_E_ARG;
    if (T == 0 || N == 0) return 0;
    if (!bfpq_hbfp_linear_mx8_ok(T, N, K)) return BFPQ_E_UNSUPPORTED;
    if (!x8 || !xs || !w8 || !ws || !out) return BFPQ_E_ARG;
    if ((reinterpret_cast<uintptr_t>(x8) | reinterpret_cast<uintptr_t>(w8)) & 15u) return BFPQ_E_ARG;
    if ((reinterpret_cast<uintptr_t>(xs) | reinterpret_cast<uintptr_t>(ws)) & 3u) return BFPQ_E_ARG;
    int variant = bfpq_g_mx8_variant;
    if (variant < 0 || variant >= kMxVariants) {
        // rounds of the chip, measured: a 256 x 128 tile costs 0.60 of a 256 x 256 one, a partly filled last round 0.72 of a full one
        // (gate_proj 2048 tokens: all-big 1.72 -> 109 us, mixed 1.43 -> 90 us; 13B gate_proj: all-big 1.72 -> 136 us, mixed 2.03 -> 147 us;
        //  8192 tokens: all-big 5.72 -> 360 us, mixed 5.43 -> 389 us: the estimate flatters the mixed plan by ~8 %, hence the margin)
        const int64_t tt = (T + 255) / 256, tn = (N + 255) / 256, tiles = tt * tn, R = tiles / 256;
        const int64_t big_cols = R > 0 ? (R * 256 / tt < tn ? R * 256 / tt : tn) : 0;
        const int64_t n_small = big_cols * 256 < N ? (N - big_cols * 256 + 127) / 128 * tt : 0;
        const double est_big = (double)R + (tiles % 256 ? 0.72 : 0.0);
        const double est_mixed = (double)(big_cols * tt) / 256.0 + 0.60 * ((double)(n_small / 256) + (n_small % 256 ? 0.72 : 0.0));
        variant = mx8_short(T, N) ? 6 : (est_mixed < 0.92 * est_big ? 5 : 3);
    }
    hipStream_t s = (hipStream_t)stream;
    const uint8_t *a = (const uint8_t*)x8, *as = (const uint8_t*)xs, *b = (const uint8_t*)w8, *bs = (const uint8_t*)ws;
    if (out_dtype == BFPQ_F32) return launch_mx8<BFPQ_F32>(variant, a, as, b, bs, bias, out, (int)T, (int)N, (int)K, s);
    if (out_dtype == BFPQ_F16) return launch_mx8<BFPQ_F16>(variant, a, as, b, bs, bias, out, (int)T, (int)N, (int)K, s);
    return launch_mx8<BFPQ_BF16>(variant, a, as, b, bs, bias, out, (int)T, (int)N, (int)K, s);
}

/* K split for short token counts: how many parts the plan wants (1: call bfpq_hbfp_linear_mx8), and the call that takes the
 * fp32 slabs [parts, T, N] */
int bfpq_hbfp_linear_mx8_parts(int64_t T, int64_t N, int64_t K)
{
    if (!bfpq_hbfp_linear_mx8_ok(T, N, K) || !mx8_short(T, N) || (bfpq_g_mx8_variant >= 0 && bfpq_g_mx8_variant != 6)) return 1;
    const int64_t tiles = ((T + 127) / 128) * ((N + 127) / 128), trips = K / 256;
    if (tiles > 128 || trips < 4) return 1;
    int64_t parts = 256 / tiles;
    if (parts > 8) parts = 8;
    if (parts > trips / 2) parts = trips / 2;                                   // at least two trips per part
    return parts < 2 ? 1 : (int)parts;
}

int bfpq_hbfp_linear_mx8_splitk(const void* x8, const void* xs, const void* w8, const void* ws, const void* bias, void* out, float* slabs, int parts,
                                int64_t T, int64_t N, int64_t K, int out_dtype, void* stream)
{
    if (out_dtype < 0 || out_dtype > 2 || T < 0 || N < 0 || parts < 1 || parts > 64) return BFPQ_E_ARG;
    if (T == 0 || N == 0) return 0;
    if (!bfpq_hbfp_linear_mx8_ok(T, N, K)) return BFPQ_E_UNSUPPORTED;
    if (!x8 || !xs || !w8 || !ws || !out || !slabs) return BFPQ_E_ARG;
    if ((reinterpret_cast<uintptr_t>(x8) | reinterpret_cast<uintptr_t>(w8) | reinterpret_cast<uintptr_t>(slabs)) & 15u) return BFPQ_E_ARG;
    if ((reinterpret_cast<uintptr_t>(xs) | reinterpret_cast<uintptr_t>(ws)) & 3u) return BFPQ_E_ARG;
    const int trips = (int)(K / 256);
    if (parts > trips) parts = trips;
    const int tpp = (trips + parts - 1) / parts;
    parts = (trips + tpp - 1) / tpp;                                            // (no empty part)
    const int tiles_t = (int)((T + 127) / 128), tiles = tiles_t * (int)((N + 127) / 128);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((k_mx8_gemm_ring_splitk<2, 2, 2, 2, 2, 3>), dim3((unsigned)(tiles * parts)), dim3(256), 0, s, (const uint8_t*)x8, (const uint8_t*)xs,
                       (const uint8_t*)w8, (const uint8_t*)ws, slabs, (int)T, (int)N, (int)K, tiles_t, tiles, tpp);
    const int64_t tn = T * N;
    int64_t g = (tn + 255) / 256;
    if (g > 2048) g = 2048;
    if (out_dtype == BFPQ_F32) hipLaunchKernelGGL((k_mx8_splitk_reduce<BFPQ_F32>), dim3((unsigned)g), dim3(256), 0, s, (const float*)slabs, bias, out, tn, (int)N, parts);
    else if (out_dtype == BFPQ_F16) hipLaunchKernelGGL((k_mx8_splitk_reduce<BFPQ_F16>), dim3((unsigned)g), dim3(256), 0, s, (const float*)slabs, bias, out, tn, (int)N, parts);
    else hipLaunchKernelGGL((k_mx8_splitk_reduce<BFPQ_BF16>), dim3((unsigned)g), dim3(256), 0, s, (const float*)slabs, bias, out, tn, (int)N, parts);
    return (int)hipGetLastError();
}

}  // extern "C"
