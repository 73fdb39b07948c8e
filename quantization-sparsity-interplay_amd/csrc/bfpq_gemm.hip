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

}  // extern "C"
