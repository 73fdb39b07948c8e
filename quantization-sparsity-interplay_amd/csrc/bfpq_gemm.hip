// bfpq_gemm.hip -- consumer of the packed HBFP format: out[t][n] = sum_k x[t][k] * W[n][k] for a
// handful of tokens (decode), with W in packed HBFP4 (int4 codes + int8 exponent per block of 64, as written by
// bfpq_quantize_nm) and x in packed HBFP8 (int8 codes + int8 exponent per block of 64).
//
// This is HBFP arithmetic proper: inside a block the dot product is an exact INTEGER sum of mantissa products
// (v_mfma_i32_16x16x64_i8: K = 64 is exactly one block), multiplied once by 2^(ew-3) * 2^(ex-7) and accumulated
// across blocks in fp32.  The reference instead runs an ordinary fp GEMM on the fake-quantised tensors
// (bfp_ops.py:187-190); the two agree up to fp32 summation order.
//
// A wave owns 16 output rows and one slice of K.  Lane l = (r = l & 15, q = l >> 4):
//   A operand: 16 weight codes of row n0 + r, k = 64 b + 16 q + j: 8 bytes of nibbles -> unsigned u = code + 8
//              (nibble ^ 8; the -8 * sum_k x_k correction comes from a second MFMA with an all-ones A operand)
//   B operand: 16 x codes of token r, same k
//   C        : lane holds rows 4 q + j (j = 0..3) for token r
// The k order inside the instruction is irrelevant as long as A and B use the same one.
// Split-K partial sums go to slabs [slices][16][N] (plain stores), a second tiny kernel adds them in slice order
// (deterministic, no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bfpq.h"
#include "bfpq_common.h"

using namespace bfpq;

namespace {

typedef int int4v __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_hbfp_linear_decode(const uint8_t* __restrict__ wcodes, const int8_t* __restrict__ wexp,
                                                           const int8_t* __restrict__ xcodes, const int8_t* __restrict__ xexp,
                                                           float* __restrict__ slabs,
                                                           int N, int K, int groups_per_slice, int slices, int wm, int xm)
{
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int row_tiles = N / 16;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int rt = wave % row_tiles, slice = wave / row_tiles;
    if (slice >= slices) return;                            // grid rounded up to whole workgroups
    const int nb = K / 64;                                  // blocks per row
    const int g0 = slice * groups_per_slice;                // a group = 4 consecutive blocks
    int g1 = g0 + groups_per_slice;
    if (g1 > nb / 4) g1 = nb / 4;
    const int n0 = rt * 16;
    const uint8_t* wrow = wcodes + (size_t)(n0 + r) * (K / 2) + q * 8;
    const int8_t* xrow = xcodes + (size_t)r * K + q * 16;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = g0; g < g1; g++) {
        // exponents / sums of the 4 blocks of this group: one dword (or int4) per source row
        uint32_t we[4];
#pragma unroll
        for (int j = 0; j < 4; j++) we[j] = *reinterpret_cast<const uint32_t*>(wexp + (size_t)(n0 + 4 * q + j) * nb + 4 * g);
        const uint32_t xe = *reinterpret_cast<const uint32_t*>(xexp + (size_t)r * nb + 4 * g);
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
            const int b = 4 * g + bb;
            const uint2 pk = *reinterpret_cast<const uint2*>(wrow + (size_t)b * 32);
            const int4v bv = *reinterpret_cast<const int4v*>(xrow + (size_t)b * 64);
            const uint32_t lo0 = (pk.x & 0x0f0f0f0fu) ^ 0x08080808u, hi0 = ((pk.x >> 4) & 0x0f0f0f0fu) ^ 0x08080808u;
            const uint32_t lo1 = (pk.y & 0x0f0f0f0fu) ^ 0x08080808u, hi1 = ((pk.y >> 4) & 0x0f0f0f0fu) ^ 0x08080808u;
            int4v av;
            av.x = (int)__builtin_amdgcn_perm(hi0, lo0, 0x05010400u);
            av.y = (int)__builtin_amdgcn_perm(hi0, lo0, 0x07030602u);
            av.z = (int)__builtin_amdgcn_perm(hi1, lo1, 0x05010400u);
            av.w = (int)__builtin_amdgcn_perm(hi1, lo1, 0x07030602u);
            const int4v zero = {0, 0, 0, 0};
            const int4v ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
            const int4v c = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, zero, 0, 0, 0);
            // sum of the 64 x codes of this block for token r: every row of ones x B holds it (the matrix pipe is idle)
            const int4v csum = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, bv, zero, 0, 0, 0);
            const int xs_b = csum.x;
            const int xe_b = (int)(int8_t)(xe >> (8 * bb));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int we_b = (int)(int8_t)(we[j] >> (8 * bb));
                const int cj = j == 0 ? c.x : (j == 1 ? c.y : (j == 2 ? c.z : c.w));
                acc[j] += (float)(cj - 8 * xs_b) * ldexpf(1.0f, we_b - wm + xe_b - xm);
            }
        }
    }
    float4* dst = reinterpret_cast<float4*>(slabs + ((size_t)slice * 16 + r) * N + n0 + 4 * q);
    *dst = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// Same product from the MFMA-TILED weight layout (PackedBFP.to_mfma_tiles(), a one-time repack):
//   wtiles [N/16][K/128][64 lanes][16 B]: lane l = r + 16 q holds, for the block pair p, bytes 0-7 = the 16 codes
//          (k = 16 q .. 16 q + 15) of row n0 + r in block 2p, bytes 8-15 = the same for block 2p + 1
//   wexpt  [N/16][K/128][16 rows][2]: exponents of the pair's two blocks, rows contiguous (lane reads rows 4q..4q+3)
// so that a wave's A operand for two blocks is ONE fully coalesced 1-KiB load (16 B per lane) instead of sixteen
// 32-byte row segments.
// One workgroup owns RT adjacent 16-row tiles and all of K; each of its waves owns one K slice and multiplies it into all
// RT tiles, so the x operand (read from L2, 2 bytes per weight byte for one tile) is loaded once per RT tiles.  The slices
// meet in LDS and are summed in slice order: the result does not depend on scheduling, no second kernel, no workspace.
template <int DT, int RT>
__global__ void __launch_bounds__(RT == 1 ? 1024 : 512) k_hbfp_linear_decode_tiled(const uint4* __restrict__ wtiles, const uint2* __restrict__ wexpt,
                                                                  const int8_t* __restrict__ xcodes, const int8_t* __restrict__ xexp,
                                                                  void* __restrict__ out, int T, int N, int K, int pairs_per_slice,
                                                                  int wm, int xm)
{
    extern __shared__ float red[];                                  // [slice][tile][token][row]
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    // the slice is wave-uniform: telling the compiler keeps every pair index in SGPRs, so each load is "scalar base +
    // constant per-lane offset" and no address lives in a VGPR
    const int rt0 = blockIdx.x * RT, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), slices = blockDim.x >> 6;
    const int P = K / 128, nb = K / 64;
    const int p0 = slice * pairs_per_slice;
    int p1 = p0 + pairs_per_slice;
    if (p1 > P) p1 = P;
    const char* wbase = reinterpret_cast<const char*>(wtiles) + (size_t)rt0 * P * 1024;   // tile i, pair p: + (i * P + p) * 1024
    const char* ebase = reinterpret_cast<const char*>(wexpt) + (size_t)rt0 * P * 32;      // 16 rows x 2 B per pair
    const uint32_t wlane = lane * 16, elane = q * 8;                                      // lane takes rows 4q..4q+3 of the exponents
    // token columns past T repeat token T-1 (their results are dropped): identical addresses coalesce, so the x traffic
    // from L2 scales with the tokens actually present instead of always being 16 rows
    // more than 16 tokens: blockIdx.y walks the groups of 16 (the weight is then re-read once per group, from the
    // 256 MB Infinity Cache for any weight that fits it)
    const int tok0 = blockIdx.y * 16;
    const int Tg = T - tok0 < 16 ? T - tok0 : 16;                 // tokens of this group, 1..16
    const int rx = tok0 + (r < Tg ? r : Tg - 1);
    const uint32_t xlane = (uint32_t)rx * K + q * 16, xelane = (uint32_t)rx * nb;
    const int4v zero = {0, 0, 0, 0};
    const int4v ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    float acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; i++) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
    auto unpack = [&](uint32_t s0, uint32_t s1) __attribute__((always_inline)) {
        // 16 nibbles -> 16 bytes u = code + 8 (two's-complement nibble ^ 8), in k order
        const uint32_t lo0 = (s0 & 0x0f0f0f0fu) ^ 0x08080808u, hi0 = ((s0 >> 4) & 0x0f0f0f0fu) ^ 0x08080808u;
        const uint32_t lo1 = (s1 & 0x0f0f0f0fu) ^ 0x08080808u, hi1 = ((s1 >> 4) & 0x0f0f0f0fu) ^ 0x08080808u;
        int4v av;
        av.x = (int)__builtin_amdgcn_perm(hi0, lo0, 0x05010400u);
        av.y = (int)__builtin_amdgcn_perm(hi0, lo0, 0x07030602u);
        av.z = (int)__builtin_amdgcn_perm(hi1, lo1, 0x05010400u);
        av.w = (int)__builtin_amdgcn_perm(hi1, lo1, 0x07030602u);
        return av;
    };
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    struct Pair { int4v w[RT]; u32x2 e[RT]; int4v b0, b1; uint32_t xe4; int xsh; };
    const int plast = p1 - 1;
    auto load = [&](int p) __attribute__((always_inline)) {
        p = p < plast ? p : plast;                                  // clamped: every load is unconditional (no vmcnt(0) stalls)
        Pair t;
#pragma unroll
        for (int i = 0; i < RT; i++) {
            t.w[i] = *reinterpret_cast<const int4v*>(wbase + ((size_t)i * P + p) * 1024 + wlane);
            t.e[i] = *reinterpret_cast<const u32x2*>(ebase + ((size_t)i * P + p) * 32 + elane);
        }
        const int8_t* xp = xcodes + (size_t)p * 128;
        t.b0 = *reinterpret_cast<const int4v*>(xp + xlane);
        t.b1 = *reinterpret_cast<const int4v*>(xp + 64 + xlane);
        // the pair's two x exponents as part of a 4-byte load that never leaves the row (K >= 256): a 2-byte load would be
        // zero-extended by an instruction the compiler hoists above the prefetch, which then waits for the whole set
        const int xo = 2 * p < nb - 4 ? 2 * p : nb - 4;
        __builtin_memcpy(&t.xe4, xexp + xo + xelane, 4);
        t.xsh = (2 * p - xo) * 8;
        return t;
    };
    auto compute = [&](const Pair& t, bool valid) __attribute__((always_inline)) {
        // sum of the 64 x codes of the block for token r: every row of ones x B holds it (the matrix pipe is idle)
        const int4v s0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, t.b0, zero, 0, 0, 0);
        const int4v s1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, t.b1, zero, 0, 0, 0);
        const int eoff = valid ? -(wm + xm) : -1000;                // past-the-end pair (wave-uniform): 2^-1000 flushes its scale to +0
        const uint32_t xe2 = t.xe4 >> t.xsh;
        const int xe0 = (int)(int8_t)(xe2 & 0xff) + eoff, xe1 = (int)(int8_t)((xe2 >> 8) & 0xff) + eoff;
#pragma unroll
        for (int i = 0; i < RT; i++) {
            const int4v c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(unpack((uint32_t)t.w[i].x, (uint32_t)t.w[i].y), t.b0, zero, 0, 0, 0);
            const int4v c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(unpack((uint32_t)t.w[i].z, (uint32_t)t.w[i].w), t.b1, zero, 0, 0, 0);
            const int c0v[4] = {c0.x, c0.y, c0.z, c0.w}, c1v[4] = {c1.x, c1.y, c1.z, c1.w};
            const uint32_t ew[2] = {t.e[i].x, t.e[i].y};            // rows 4q..4q+3: byte 2j = block 2p, byte 2j+1 = block 2p+1
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t pr = (ew[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                const int we0 = (int)(int8_t)(pr & 0xff), we1 = (int)(int8_t)(pr >> 8);
                acc[i][j] += (float)(c0v[j] - 8 * s0.x) * ldexpf(1.0f, we0 + xe0);
                acc[i][j] += (float)(c1v[j] - 8 * s1.x) * ldexpf(1.0f, we1 + xe1);
            }
        }
    };
    if (p0 < p1) {
        // three named register sets: while one pair is being multiplied the next two are in flight.  pin() is an empty asm
        // that "touches" memory and a pair's registers: loads cannot cross it and the pair's multiplies cannot start before
        // it, so the wait for a pair comes AFTER the next prefetch was issued (with no store in the loop the compiler
        // otherwise sinks every load down to its first use).  Loads past the slice are clamped and their products masked.
        auto pin = [&](Pair& t) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < RT; i++) asm volatile("" : "+v"(t.w[i]), "+v"(t.e[i]) : : "memory");
            asm volatile("" : "+v"(t.b0), "+v"(t.b1), "+v"(t.xe4) : : "memory");
        };
        Pair A = load(p0), B = load(p0 + 1);
        for (int p = p0; p < p1; p += 3) {
            Pair C = load(p + 2);
            pin(A);
            compute(A, true);
            A = load(p + 3);
            pin(B);
            compute(B, p + 1 < p1);
            B = load(p + 4);
            pin(C);
            compute(C, p + 2 < p1);
        }
    }
#pragma unroll
    for (int i = 0; i < RT; i++)
        *reinterpret_cast<float4*>(&red[(((size_t)slice * RT + i) * 16 + r) * 16 + 4 * q]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    __syncthreads();
    for (int o = threadIdx.x; o < RT * 256; o += blockDim.x) {
        const int i = o >> 8, tok = (o >> 4) & 15, n = o & 15;
        float sum = 0.f;
        for (int w = 0; w < slices; w++) sum += red[(((size_t)w * RT + i) * 16 + tok) * 16 + n];
        if (tok < Tg) {
            const size_t d = (size_t)(tok0 + tok) * N + (rt0 + i) * 16 + n;
            if constexpr (DT == BFPQ_F32) reinterpret_cast<float*>(out)[d] = sum;
            else reinterpret_cast<uint16_t*>(out)[d] = (uint16_t)f32_to_raw<DT>(sum);
        }
    }
}

template <int DT>
__global__ void __launch_bounds__(256) k_slab_reduce(const float* __restrict__ slabs, void* __restrict__ out, int T, int N, int slices)
{
    using raw_t = typename Traits<DT>::raw_t;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)T * N) return;
    const int t = (int)(i / N), n = (int)(i - (int64_t)t * N);
    float s = 0.f;
    for (int sl = 0; sl < slices; sl++) s += slabs[((size_t)sl * 16 + t) * N + n];
    reinterpret_cast<raw_t*>(out)[i] = (raw_t)f32_to_raw<DT>(s);
}

}  // namespace

extern "C" {

int bfpq_hbfp_linear_slices(int64_t N, int64_t K)
{
    if (N <= 0 || K <= 0 || N % 16 != 0 || K % 256 != 0) return BFPQ_E_UNSUPPORTED;
    const int64_t groups = K / 256, row_tiles = N / 16;
    int64_t slices = (4096 + row_tiles - 1) / row_tiles;      // aim at ~4096 waves on the chip
    if (slices > groups) slices = groups;
    if (slices < 1) slices = 1;
    const int64_t gps = (groups + slices - 1) / slices;
    return (int)((groups + gps - 1) / gps);
}

int bfpq_hbfp_linear_decode(const void* wcodes, const int8_t* wexp, const int8_t* xcodes, const int8_t* xexp,
                            void* out, float* slabs, int64_t T, int64_t N, int64_t K, int out_dtype,
                            int w_mant_bits, int x_mant_bits, void* stream)
{
    if (T < 1 || T > 16 || out_dtype < 0 || out_dtype > 2) return BFPQ_E_ARG;
    if (w_mant_bits < 1 || w_mant_bits > 3 || x_mant_bits < 1 || x_mant_bits > 7) return BFPQ_E_ARG;
    const int slices = bfpq_hbfp_linear_slices(N, K);
    if (slices < 0) return slices;
    if (!wcodes || !wexp || !xcodes || !xexp || !out || !slabs) return BFPQ_E_ARG;
    const int64_t groups = K / 256;
    const int gps = (int)((groups + slices - 1) / slices);
    const int64_t waves = (N / 16) * slices;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_hbfp_linear_decode, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,
                       (const uint8_t*)wcodes, wexp, xcodes, xexp, slabs, (int)N, (int)K, gps, slices, w_mant_bits, x_mant_bits);
    const dim3 rg((unsigned)((T * N + 255) / 256));
    if (out_dtype == BFPQ_F32) hipLaunchKernelGGL((k_slab_reduce<BFPQ_F32>), rg, dim3(256), 0, s, (const float*)slabs, out, (int)T, (int)N, slices);
    else if (out_dtype == BFPQ_F16) hipLaunchKernelGGL((k_slab_reduce<BFPQ_F16>), rg, dim3(256), 0, s, (const float*)slabs, out, (int)T, (int)N, slices);
    else hipLaunchKernelGGL((k_slab_reduce<BFPQ_BF16>), rg, dim3(256), 0, s, (const float*)slabs, out, (int)T, (int)N, slices);
    return (int)hipGetLastError();
}

__attribute__((visibility("hidden"))) int bfpq_g_gemm_rt = 0;                                                  // 0 = choose; 1, 2, 4 force (bfpq_tune, BFPQ_TUNE_GEMM_ROW_TILES)
static int gemm_rt(int64_t N, int64_t T)
{
    if (bfpq_g_gemm_rt) return (N % (16 * bfpq_g_gemm_rt) == 0) ? bfpq_g_gemm_rt : 1;
    // A few tokens: x traffic is small (token columns coalesce), one tile per wave and the most waves win.  More tokens:
    // the x operand read from L2 is 2 bytes per weight byte unless tiles share it -- as many tiles per wave as still
    // leave >= 160 workgroups for the 256 CUs (measured: tools_dev/ab_gemm.py).
    if (T <= 4) return 1;
    const int64_t G = (T + 15) / 16;                                // token groups: each is its own set of workgroups
    for (int rt = 4; rt > 1; rt >>= 1)
        if (N % (16 * rt) == 0 && N / (16 * rt) * G >= 160) return rt;
    return 1;
}
static int gemm_slices(int64_t N, int64_t K, int rt)
{
    const int64_t pairs = K / 128, groups = N / 16 / rt;
    int64_t slices = (4096 / rt + groups - 1) / groups;             // ~4096 tile-slices on the chip, at most 16 waves per workgroup
    const int64_t cap = rt == 1 ? 16 : 8;                           // RT > 1 needs > 128 VGPRs: 512-thread workgroups
    if (slices > cap) slices = cap;
    if (slices > pairs) slices = pairs;
    if (slices < 1) slices = 1;
    const int64_t pps = (pairs + slices - 1) / slices;
    return (int)((pairs + pps - 1) / pps);
}

int bfpq_hbfp_linear_tiled_ok(int64_t N, int64_t K)
{
    return (N > 0 && K >= 256 && N % 16 == 0 && K % 128 == 0) ? 1 : 0;
}

int bfpq_hbfp_linear_decode_tiled(const void* wtiles, const void* wexpt, const int8_t* xcodes, const int8_t* xexp,
                                  void* out, int64_t T, int64_t N, int64_t K, int out_dtype,
                                  int w_mant_bits, int x_mant_bits, void* stream)
{
    if (T < 1 || T > 64 || out_dtype < 0 || out_dtype > 2) return BFPQ_E_ARG;
    if (w_mant_bits < 1 || w_mant_bits > 3 || x_mant_bits < 1 || x_mant_bits > 7) return BFPQ_E_ARG;
    if (!bfpq_hbfp_linear_tiled_ok(N, K)) return BFPQ_E_UNSUPPORTED;
    if (!wtiles || !wexpt || !xcodes || !xexp || !out) return BFPQ_E_ARG;
    const int rt = gemm_rt(N, T);
    const int slices = gemm_slices(N, K, rt);
    const int64_t pairs = K / 128;
    const int pps = (int)((pairs + slices - 1) / slices);
    const dim3 grid((unsigned)(N / 16 / rt), (unsigned)((T + 15) / 16)), wg((unsigned)(64 * slices));
    const size_t lds = (size_t)slices * rt * 256 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define BFPQ_LAUNCH_TILED(DT, RT)                                                                                                \
    hipLaunchKernelGGL((k_hbfp_linear_decode_tiled<DT, RT>), grid, wg, lds, s, (const uint4*)wtiles, (const uint2*)wexpt, xcodes, \
                       xexp, out, (int)T, (int)N, (int)K, pps, w_mant_bits, x_mant_bits)
#define BFPQ_LAUNCH_TILED_DT(RT)                                                                                                 \
    do {                                                                                                                          \
        if (out_dtype == BFPQ_F32) BFPQ_LAUNCH_TILED(BFPQ_F32, RT);                                                               \
        else if (out_dtype == BFPQ_F16) BFPQ_LAUNCH_TILED(BFPQ_F16, RT);                                                          \
        else BFPQ_LAUNCH_TILED(BFPQ_BF16, RT);                                                                                    \
    } while (0)
    if (rt == 4) BFPQ_LAUNCH_TILED_DT(4);
    else if (rt == 2) BFPQ_LAUNCH_TILED_DT(2);
    else BFPQ_LAUNCH_TILED_DT(1);
#undef BFPQ_LAUNCH_TILED_DT
#undef BFPQ_LAUNCH_TILED
    return (int)hipGetLastError();
}

}  // extern "C"
