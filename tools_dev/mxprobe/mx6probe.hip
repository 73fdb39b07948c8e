// Probe of the e2m3 (FP6) form of v_mfma_scale_f32_32x32x64_f8f6f4 (development tool): are small integers packed as a little-endian
// stream of 6-bit fields (value j of a lane at bits 6j .. 6j+5 of its 192-bit operand) multiplied exactly, and how long does it take?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int FMT>
__global__ void k_one(const v8i* A, const v8i* B, const int* SA, const int* SB, float* D)
{
    const int w = blockIdx.x, l = threadIdx.x;
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[(size_t)w * 64 + l], B[(size_t)w * 64 + l], acc, FMT, FMT, 0, SA[(size_t)w * 64 + l], 0, SB[(size_t)w * 64 + l]);
    for (int r = 0; r < 16; ++r) D[((size_t)w * 64 + l) * 16 + r] = acc[r];
}
// issue-rate: N dependent-free instructions per wave, 4 accumulators
template <int FMT>
__global__ void k_rate(const v8i* A, float* D, int iters, long long* cyc)
{
    const int l = threadIdx.x & 63;
    const v8i a = A[l], b = A[64 + l];
    v16f c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    D[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
static uint32_t e2m3(int v) { static const int tab[8] = {0, 8, 16, 20, 24, 26, 28, 30}; int m = v < 0 ? -v : v; return (uint32_t)tab[m] | (v < 0 ? 0x20u : 0u); }
static uint8_t e4m3(int v) { int m = v < 0 ? -v : v; uint8_t b = m == 0 ? 0 : m == 1 ? 0x38 : m < 4 ? 0x40 + 4 * (m - 2) : m < 8 ? 0x48 + 2 * (m - 4) : 0x50 + (m - 8); return b | (v < 0 ? 0x80 : 0); }
int main()
{
    const int NW = 64;
    std::vector<uint32_t> A(NW * 64 * 8, 0), B(NW * 64 * 8, 0), SA(NW * 64), SB(NW * 64);
    std::vector<int> av(NW * 64 * 32), bv(NW * 64 * 32), ea(NW * 64), eb(NW * 64);
    uint32_t s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int w = 0; w < NW; ++w) for (int l = 0; l < 64; ++l) {
        const int idx = w * 64 + l;
        // one HBFP block per instruction: lanes l and l ^ 32 share the row's scale
        ea[idx] = 120 + (int)((w * 7 + (l & 31) * 3) % 14); eb[idx] = 118 + (int)((w * 5 + (l & 31)) % 16);
        SA[idx] = ea[idx] | 0x01010100u * (rnd() & 0xff); SB[idx] = eb[idx] | 0x01010100u * (rnd() & 0xff);
        for (int j = 0; j < 32; ++j) {
            av[idx * 32 + j] = (int)(rnd() % 15) - 7; bv[idx * 32 + j] = (int)(rnd() % 15) - 7;
            const int bit = 6 * j;
            uint64_t fa = (uint64_t)e2m3(av[idx * 32 + j]) << (bit & 31), fb = (uint64_t)e2m3(bv[idx * 32 + j]) << (bit & 31);
            A[idx * 8 + (bit >> 5)] |= (uint32_t)fa; if ((bit >> 5) + 1 < 8) A[idx * 8 + (bit >> 5) + 1] |= (uint32_t)(fa >> 32);
            B[idx * 8 + (bit >> 5)] |= (uint32_t)fb; if ((bit >> 5) + 1 < 8) B[idx * 8 + (bit >> 5) + 1] |= (uint32_t)(fb >> 32);
        }
    }
    void *dA, *dB, *dSA, *dSB, *dD; long long* dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dSA, SA.size() * 4)); CK(hipMalloc(&dSB, SB.size() * 4));
    CK(hipMalloc(&dD, NW * 64 * 16 * 4)); CK(hipMalloc(&dC, 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dSA, SA.data(), SA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dSB, SB.data(), SB.size() * 4, hipMemcpyHostToDevice));
    k_one<2><<<NW, 64>>>((v8i*)dA, (v8i*)dB, (int*)dSA, (int*)dSB, (float*)dD);
    CK(hipDeviceSynchronize());
    std::vector<float> o(NW * 64 * 16);
    CK(hipMemcpy(o.data(), dD, o.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; double worst = 0;
    for (int w = 0; w < NW; ++w) for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
        double acc = 0;
        for (int g = 0; g < 2; ++g) { const int la = w * 64 + r + 32 * g, lb = w * 64 + c + 32 * g; long sum = 0; for (int j = 0; j < 32; ++j) sum += av[la * 32 + j] * bv[lb * 32 + j];
            acc += (double)sum * ldexp(1.0, ea[w * 64 + r] - 127 + eb[w * 64 + c] - 127); }
        const int lane = c + 32 * ((r >> 2) & 1), reg = (r & 3) + 4 * (r >> 3);
        const double g = o[((size_t)w * 64 + lane) * 16 + reg], d = fabs(g - acc);
        if (d > 1e-6 * (fabs(acc) + 1)) { if (bad < 5) printf("  mismatch w%d (%d,%d): got %g want %g\n", w, r, c, g, acc); ++bad; }
        if (d > worst) worst = d;
    }
    printf("e2m3 operands as little-endian 6-bit streams, random integers -7..7 x block scales vs host: %d mismatches of %d, worst %g\n", bad, NW * 1024, worst);
    // issue rate: e4m3 (fmt 0) vs e2m3 (fmt 2) vs e2m1 (fmt 4), one wave per SIMD
    for (int rep = 0; rep < 2; ++rep) {
        long long c0 = 0, c2 = 0, c4 = 0; const int iters = 4096;
        k_rate<0><<<1, 64>>>((v8i*)dA, (float*)dD, iters, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(&c0, dC, 8, hipMemcpyDeviceToHost));
        k_rate<2><<<1, 64>>>((v8i*)dA, (float*)dD, iters, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(&c2, dC, 8, hipMemcpyDeviceToHost));
        k_rate<4><<<1, 64>>>((v8i*)dA, (float*)dD, iters, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(&c4, dC, 8, hipMemcpyDeviceToHost));
        printf("s_memtime ticks per instruction (one wave, 4 accumulators): e4m3 %.1f, e2m3 %.1f, e2m1 %.1f\n", (double)c0 / (4.0 * iters), (double)c2 / (4.0 * iters), (double)c4 / (4.0 * iters));
    }
    return 0;
}
