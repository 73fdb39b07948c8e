// streaming-structure exploration: pure 16 B/lane copy kernels, same buffers/rotation as bench.py
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ u4v ld(const u4v* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void st(u4v* p, u4v v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// K0: grid-stride, prefetch distance 1 (the current kernel's loop)
template <bool NT, int T> __global__ void __launch_bounds__(T) k_gs_pf1(const u4v* in, u4v* out, long n)
{
    const long stride = (long)gridDim.x * T, last = n - 1;
    long item = (long)blockIdx.x * T + threadIdx.x;
    u4v cur = ld<NT>(in + (item < last ? item : last));
    for (; item < n; item += stride) {
        const long pf = item + stride;
        u4v nxt = ld<NT>(in + (pf < last ? pf : last));
        st<NT>(out + item, cur);
        cur = nxt;
    }
}
// K1: contiguous chunk per workgroup, prefetch distance D
template <bool NT, int T, int D> __global__ void __launch_bounds__(T) k_chunk(const u4v* in, u4v* out, long n)
{
    const long per = ((n + gridDim.x - 1) / gridDim.x + T - 1) / T * T;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n, last = n - 1;
    long item = lo + threadIdx.x;
    u4v buf[D];
#pragma unroll
    for (int d = 0; d < D; d++) { const long p = item + (long)d * T; buf[d] = ld<NT>(in + (p < last ? p : last)); }
    for (; item < hi; item += (long)D * T) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const long cur = item + (long)d * T;
            const long p = cur + (long)D * T;
            u4v v = buf[d];
            buf[d] = ld<NT>(in + (p < last ? p : last));
            if (cur < hi) st<NT>(out + cur, v);
        }
    }
}
// K2: one item per thread, no loop
template <bool NT, int T> __global__ void __launch_bounds__(T) k_flat(const u4v* in, u4v* out, long n)
{
    const long item = (long)blockIdx.x * T + threadIdx.x;
    if (item < n) st<NT>(out + item, ld<NT>(in + item));
}
// K3: grid-stride, U independent items per iteration (all loads first, then stores)
template <bool NT, int T, int U> __global__ void __launch_bounds__(T) k_gs_u(const u4v* in, u4v* out, long n)
{
    const long stride = (long)gridDim.x * T;
    for (long item = (long)blockIdx.x * T + threadIdx.x; item < n; item += stride * U) {
        u4v v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const long p = item + u * stride; if (p < n) v[u] = ld<NT>(in + p); }
#pragma unroll
        for (int u = 0; u < U; u++) { const long p = item + u * stride; if (p < n) st<NT>(out + p, v[u]); }
    }
}

// K4: grid-stride, each wave moves TWO ADJACENT 1-KB tiles per sweep (lane l: items base+l and base+64+l), one sweep ahead
template <bool NT, int T> __global__ void __launch_bounds__(T) k_gs_adj2(const u4v* in, u4v* out, long n)
{
    const long wave = ((long)blockIdx.x * T + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const long nw = (long)gridDim.x * T / 64, last = n - 1;
    long base = wave * 128 + lane;
    u4v a = ld<NT>(in + (base < last ? base : last)), b = ld<NT>(in + (base + 64 < last ? base + 64 : last));
    for (; base - lane < n; base += nw * 128) {
        const long pf = base + nw * 128;
        u4v na = ld<NT>(in + (pf < last ? pf : last)), nb = ld<NT>(in + (pf + 64 < last ? pf + 64 : last));
        if (base < n) st<NT>(out + base, a);
        if (base + 64 < n) st<NT>(out + base + 64, b);
        a = na; b = nb;
    }
}
// K5: 32 contiguous bytes per lane (two 16-B accesses at lane*32 and lane*32+16), one sweep ahead
template <bool NT, int T> __global__ void __launch_bounds__(T) k_gs_wide(const u4v* in, u4v* out, long n)
{
    const long stride = (long)gridDim.x * T * 2, last = n - 1;
    long base = ((long)blockIdx.x * T + threadIdx.x) * 2;
    u4v a = ld<NT>(in + (base < last ? base : last)), b = ld<NT>(in + (base + 1 < last ? base + 1 : last));
    for (; base < n; base += stride) {
        const long pf = base + stride;
        u4v na = ld<NT>(in + (pf < last ? pf : last)), nb = ld<NT>(in + (pf + 1 < last ? pf + 1 : last));
        st<NT>(out + base, a);
        if (base + 1 < n) st<NT>(out + base + 1, b);
        a = na; b = nb;
    }
}
// K6: like K0 but no prefetch (load, store)
template <bool NT, int T> __global__ void __launch_bounds__(T) k_gs_nopf(const u4v* in, u4v* out, long n)
{
    const long stride = (long)gridDim.x * T;
    for (long item = (long)blockIdx.x * T + threadIdx.x; item < n; item += stride) st<NT>(out + item, ld<NT>(in + item));
}

// K7: K0 with an XCD-aware block remap: XCD x (blocks b % 8 == x) streams a contiguous eighth of every sweep
template <bool NT, int T> __global__ void __launch_bounds__(T) k_gs_pf1_xcd(const u4v* in, u4v* out, long n)
{
    const long G = gridDim.x, q = G / 8, r = G % 8, x = blockIdx.x % 8;
    const long bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blockIdx.x / 8;     // bijective for any G
    const long stride = G * T, last = n - 1;
    long item = bid * T + threadIdx.x;
    u4v cur = ld<NT>(in + (item < last ? item : last));
    for (; item < n; item += stride) {
        const long pf = item + stride;
        u4v nxt = ld<NT>(in + (pf < last ? pf : last));
        st<NT>(out + item, cur);
        cur = nxt;
    }
}
template <bool NT, int T> __global__ void __launch_bounds__(T) k_flat_xcd(const u4v* in, u4v* out, long n)
{
    const long G = gridDim.x, q = G / 8, r = G % 8, x = blockIdx.x % 8;
    const long bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blockIdx.x / 8;
    const long item = bid * T + threadIdx.x;
    if (item < n) st<NT>(out + item, ld<NT>(in + item));
}

struct Var { std::string name; void (*fn)(const u4v*, u4v*, long); int grid, threads; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main()
{
    const long rows = 4096, cols = 11008, n = rows * cols * 2 / 16;
    const int R = 8, L = 100, ROUNDS = 7;
    std::vector<u4v*> in(R), out(R);
    for (int r = 0; r < R; r++) { CK(hipMalloc(&in[r], n * 16)); CK(hipMalloc(&out[r], n * 16)); CK(hipMemset(in[r], r + 1, n * 16)); }
    std::vector<Var> vars;
    const int gfull256 = (int)((n + 255) / 256);
#define ADD(nm, kern, g, t) vars.push_back({nm, kern, g, t})
    ADD("gs_pf1 g2048", (k_gs_pf1<false, 256>), 2048, 256);
    ADD("gs_pf1 nt g2048", (k_gs_pf1<true, 256>), 2048, 256);
    ADD("gs_pf1 nt g1024", (k_gs_pf1<true, 256>), 1024, 256);
    ADD("gs_pf1 nt g1280", (k_gs_pf1<true, 256>), 1280, 256);
    ADD("gs_pf1 nt g1024 t512", (k_gs_pf1<true, 512>), 1024, 512);
    ADD("gs_pf1 nt g512 t512", (k_gs_pf1<true, 512>), 512, 512);
    ADD("gs_pf1 nt g256 t1024", (k_gs_pf1<true, 1024>), 256, 1024);
    ADD("gs_pf1 nt g512 t1024", (k_gs_pf1<true, 1024>), 512, 1024);
    ADD("chunk d1 nt g1024", (k_chunk<true, 256, 1>), 1024, 256);
    ADD("chunk d2 nt g1024", (k_chunk<true, 256, 2>), 1024, 256);
    ADD("chunk d4 nt g1024", (k_chunk<true, 256, 4>), 1024, 256);
    ADD("chunk d2 nt g2048", (k_chunk<true, 256, 2>), 2048, 256);
    ADD("chunk d2 nt g512", (k_chunk<true, 256, 2>), 512, 256);
    ADD("chunk d4 nt g512", (k_chunk<true, 256, 4>), 512, 256);
    ADD("chunk d2 g1024", (k_chunk<false, 256, 2>), 1024, 256);
    ADD("chunk d2 nt g2752", (k_chunk<true, 256, 2>), 2752, 256);
    ADD("flat", (k_flat<false, 256>), gfull256, 256);
    ADD("flat nt", (k_flat<true, 256>), gfull256, 256);
    ADD("gs_u2 nt g1024", (k_gs_u<true, 256, 2>), 1024, 256);
    ADD("gs_u4 nt g1024", (k_gs_u<true, 256, 4>), 1024, 256);
    ADD("gs_u2 nt g2048", (k_gs_u<true, 256, 2>), 2048, 256);
    ADD("gs_u4 nt g512", (k_gs_u<true, 256, 4>), 512, 256);
    ADD("gs_u4 g1024", (k_gs_u<false, 256, 4>), 1024, 256);
    ADD("gs_pf1_xcd nt g1224", (k_gs_pf1_xcd<true, 256>), 1224, 256);
    ADD("gs_pf1_xcd nt g1024", (k_gs_pf1_xcd<true, 256>), 1024, 256);
    ADD("gs_pf1_xcd nt g2048", (k_gs_pf1_xcd<true, 256>), 2048, 256);
    ADD("gs_pf1_xcd nt g2448", (k_gs_pf1_xcd<true, 256>), 2448, 256);
    ADD("gs_pf1_xcd g1224", (k_gs_pf1_xcd<false, 256>), 1224, 256);
    ADD("flat_xcd nt", (k_flat_xcd<true, 256>), gfull256, 256);
    ADD("gs_adj2 nt g1024", (k_gs_adj2<true, 256>), 1024, 256);
    ADD("gs_adj2 nt g612", (k_gs_adj2<true, 256>), 612, 256);
    ADD("gs_adj2 nt g2048", (k_gs_adj2<true, 256>), 2048, 256);
    ADD("gs_wide nt g1024", (k_gs_wide<true, 256>), 1024, 256);
    ADD("gs_wide nt g612", (k_gs_wide<true, 256>), 612, 256);
    ADD("gs_nopf nt g1224", (k_gs_nopf<true, 256>), 1224, 256);
    ADD("gs_nopf nt g2048", (k_gs_nopf<true, 256>), 2048, 256);
    ADD("gs_nopf nt g4096", (k_gs_nopf<true, 256>), 4096, 256);
    ADD("gs_pf1 nt g1224", (k_gs_pf1<true, 256>), 1224, 256);
    ADD("gs_pf1 nt g1376", (k_gs_pf1<true, 256>), 1376, 256);
    ADD("gs_pf1 nt g1101", (k_gs_pf1<true, 256>), 1101, 256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> t(vars.size());
    for (int round = 0; round < ROUNDS; round++)
        for (size_t v = 0; v < vars.size(); v++) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < L; i++) hipLaunchKernelGGL(vars[v].fn, dim3(vars[v].grid), dim3(vars[v].threads), 0, 0, in[i % R], out[i % R], n);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round) t[v].push_back(ms * 1e3f / L);
        }
    for (size_t v = 0; v < vars.size(); v++) {
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2];
        printf("%-24s median %6.2f us min %6.2f  -> %6.0f GB/s (%4.1f%%)\n", vars[v].name.c_str(), med, t[v][0], n * 32.0 / med / 1e3, n * 32.0 / med / 1e3 / 80.0);
    }
    // correctness spot check of the last variant family
    return 0;
}
