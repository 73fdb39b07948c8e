// Layout probe for v_mfma_scale_f32_{32x32x64,16x16x128}_f8f6f4 with e4m3 operands (development tool).
// One wave per case; the host builds one-hot register images and reads the maps off the results.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SHAPE, int OPA, int OPB>
__global__ void k_probe(const v8i* A, const v8i* B, const int* SA, const int* SB, float* D)
{
    const int w = blockIdx.x, l = threadIdx.x;
    const v8i a = A[(size_t)w * 64 + l], b = B[(size_t)w * 64 + l];
    const int sa = SA[(size_t)w * 64 + l], sb = SB[(size_t)w * 64 + l];
    if constexpr (SHAPE == 32) {
        v16f acc = {};
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, OPA, sa, OPB, sb);
        for (int r = 0; r < 16; ++r) D[((size_t)w * 64 + l) * 16 + r] = acc[r];
    } else {
        v4f acc = {};
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, OPA, sa, OPB, sb);
        for (int r = 0; r < 4; ++r) D[((size_t)w * 64 + l) * 16 + r] = acc[r];
    }
}

struct Case { uint8_t a[64][32], b[64][32]; uint32_t sa[64], sb[64]; };
static const uint8_t ONE = 0x38;

template <int SHAPE>
static int run(std::vector<Case>& cs, std::vector<float>& out, int opa = 0, int opb = 0)
{
    const size_t n = cs.size();
    std::vector<uint32_t> A(n * 64 * 8), B(n * 64 * 8), SA(n * 64), SB(n * 64);
    for (size_t w = 0; w < n; ++w) {
        memcpy(&A[w * 512], cs[w].a, 2048); memcpy(&B[w * 512], cs[w].b, 2048);
        memcpy(&SA[w * 64], cs[w].sa, 256); memcpy(&SB[w * 64], cs[w].sb, 256);
    }
    void *dA, *dB, *dSA, *dSB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dSA, SA.size() * 4)); CK(hipMalloc(&dSB, SB.size() * 4));
    CK(hipMalloc(&dD, n * 64 * 16 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dSA, SA.data(), SA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dSB, SB.data(), SB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dD, 0, n * 64 * 16 * 4));
    if (opa == 0 && opb == 0) k_probe<SHAPE, 0, 0><<<n, 64>>>((v8i*)dA, (v8i*)dB, (int*)dSA, (int*)dSB, (float*)dD);
    else if (opa == 1 && opb == 2) k_probe<SHAPE, 1, 2><<<n, 64>>>((v8i*)dA, (v8i*)dB, (int*)dSA, (int*)dSB, (float*)dD);
    else k_probe<SHAPE, 3, 3><<<n, 64>>>((v8i*)dA, (v8i*)dB, (int*)dSA, (int*)dSB, (float*)dD);
    CK(hipDeviceSynchronize());
    out.resize(n * 64 * 16);
    CK(hipMemcpy(out.data(), dD, out.size() * 4, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dB); hipFree(dSA); hipFree(dSB); hipFree(dD);
    return 0;
}

// D element (row, col) of case w under the guide's C/D map
template <int SHAPE> static float dget(const std::vector<float>& o, size_t w, int row, int col)
{
    if (SHAPE == 32) { int lane = col + 32 * ((row >> 2) & 1); int reg = (row & 3) + 4 * (row >> 3); return o[(w * 64 + lane) * 16 + reg]; }
    int lane = col + 16 * (row >> 2); int reg = row & 3; return o[(w * 64 + lane) * 16 + reg];
}
static void fill(Case& c, uint8_t av, uint8_t bv, uint32_t s = 0x7f7f7f7fu)
{
    memset(c.a, av, sizeof c.a); memset(c.b, bv, sizeof c.b);
    for (int l = 0; l < 64; ++l) { c.sa[l] = s; c.sb[l] = s; }
}

template <int SHAPE> static int probe()
{
    const int MN = SHAPE, K = SHAPE == 32 ? 64 : 128;
    printf("==== shape %dx%dx%d ====\n", MN, MN, K);
    std::vector<Case> cs; std::vector<float> o;
    // 1. A one-hot -> row of every (lane, byte)
    cs.assign(2048, Case());
    for (int p = 0; p < 2048; ++p) { fill(cs[p], 0, ONE); cs[p].a[p / 32][p % 32] = ONE; }
    if (run<SHAPE>(cs, o)) return 1;
    std::vector<int> rowA(2048, -1), colB(2048, -1);
    for (int p = 0; p < 2048; ++p) { int cnt = 0; for (int r = 0; r < MN; ++r) if (dget<SHAPE>(o, p, r, 0) != 0.f) { rowA[p] = r; ++cnt; } if (cnt != 1) printf("A pos %d: %d rows hit\n", p, cnt); }
    for (int p = 0; p < 2048; ++p) { fill(cs[p], ONE, 0); cs[p].b[p / 32][p % 32] = ONE; }
    if (run<SHAPE>(cs, o)) return 1;
    for (int p = 0; p < 2048; ++p) { int cnt = 0; for (int c = 0; c < MN; ++c) if (dget<SHAPE>(o, p, 0, c) != 0.f) { colB[p] = c; ++cnt; } if (cnt != 1) printf("B pos %d: %d cols hit\n", p, cnt); }
    int badA = 0, badB = 0;
    for (int p = 0; p < 2048; ++p) { int l = p / 32; badA += rowA[p] != (l & (MN - 1)); badB += colB[p] != (l & (MN - 1)); }
    printf("hypothesis row(A)=lane&%d: %d mismatches; col(B)=lane&%d: %d mismatches\n", MN - 1, badA, MN - 1, badB);
    if (badA || badB) for (int l = 0; l < 64; l += 9) { printf(" lane %d rowA:", l); for (int j = 0; j < 32; ++j) printf(" %d", rowA[l * 32 + j]); printf("\n"); }
    // 2. k pairing: A positions of row 0 x B positions of col 0
    std::vector<int> pa, pb;
    for (int p = 0; p < 2048; ++p) { if (rowA[p] == 0) pa.push_back(p); if (colB[p] == 0) pb.push_back(p); }
    printf("positions in row 0 of A: %zu, col 0 of B: %zu\n", pa.size(), pb.size());
    cs.assign(pa.size() * pb.size(), Case());
    for (size_t i = 0; i < pa.size(); ++i) for (size_t j = 0; j < pb.size(); ++j) { Case& c = cs[i * pb.size() + j]; fill(c, 0, 0); c.a[pa[i] / 32][pa[i] % 32] = ONE; c.b[pb[j] / 32][pb[j] % 32] = ONE; }
    if (run<SHAPE>(cs, o)) return 1;
    int same = 0, other = 0;
    for (size_t i = 0; i < pa.size(); ++i) { int hits = 0, partner = -1; for (size_t j = 0; j < pb.size(); ++j) if (dget<SHAPE>(o, i * pb.size() + j, 0, 0) != 0.f) { ++hits; partner = pb[j]; }
        if (hits == 1 && partner == pa[i]) ++same; else { ++other; if (other < 8) printf(" A pos (lane %d, byte %d) pairs with %d B positions, last (lane %d, byte %d)\n", pa[i] / 32, pa[i] % 32, hits, partner / 32, partner % 32); } }
    printf("k pairing: A(lane,byte) <-> B(same lane, same byte) for %d of %zu positions (%d other)\n", same, pa.size(), other);
    // 3. scale maps: A and B all ones; one lane's scale register = 0x80 in byte `sel` -> which (row, k positions) double
    for (int which = 0; which < 2; ++which) for (int sel = 0; sel < 4; sel += 3) {
        // one-hot A (row 0 positions) against all-ones B, scale one-hot over lanes
        const std::vector<int>& pp = which == 0 ? pa : pb;
        cs.assign(pp.size() * 64, Case());
        for (size_t i = 0; i < pp.size(); ++i) for (int ls = 0; ls < 64; ++ls) { Case& c = cs[i * 64 + ls]; fill(c, which == 0 ? 0 : ONE, which == 0 ? ONE : 0);
            if (which == 0) c.a[pp[i] / 32][pp[i] % 32] = ONE; else c.b[pp[i] / 32][pp[i] % 32] = ONE;
            uint32_t v = 0x7f7f7f7fu; v = (v & ~(0xffu << (8 * sel))) | (0x80u << (8 * sel));
            (which == 0 ? c.sa : c.sb)[ls] = v; }
        const int op = sel == 0 ? 0 : 3;
        if (run<SHAPE>(cs, o, op, op)) return 1;
        int ok = 0, bad = 0;
        for (size_t i = 0; i < pp.size(); ++i) { int nh = 0, lh = -1; for (int ls = 0; ls < 64; ++ls) { float v = dget<SHAPE>(o, i * 64 + ls, 0, 0); if (v == 2.f) { ++nh; lh = ls; } else if (v != 1.f) printf("  unexpected value %g\n", v); }
            const int lane = pp[i] / 32;
            if (nh == 1 && lh == lane) ++ok; else { ++bad; if (bad < 12) printf("  %c pos (lane %d, byte %d): scaled by %d lanes, last %d\n", which ? 'B' : 'A', lane, pp[i] % 32, nh, lh); } }
        printf("scale %c, op_sel %d (byte %d): element (lane,byte) takes the scale of ITS OWN lane for %d of %zu positions (%d other)\n", which ? 'B' : 'A', op, sel, ok, pp.size(), bad);
    }
    // 4. exactness: random small integers + random scales against a host sum, under the confirmed maps (lane&(MN-1), k = 32*(lane/MN)+byte)
    {
        auto enc = [](int v) -> uint8_t { int m = v < 0 ? -v : v; uint8_t b = m == 0 ? 0 : m == 1 ? 0x38 : m < 4 ? 0x40 + 4 * (m - 2) : m < 8 ? 0x48 + 2 * (m - 4) : 0x50 + (m - 8); return b | (v < 0 ? 0x80 : 0); };
        cs.assign(64, Case()); std::vector<double> want(64 * MN * MN, 0.0);
        uint32_t s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
        for (int w = 0; w < 64; ++w) { Case& c = cs[w];
            static int av[64][32], bv[64][32]; static int ea[64], eb[64];
            for (int l = 0; l < 64; ++l) { ea[l] = 120 + rnd() % 14; eb[l] = 120 + rnd() % 14; c.sa[l] = 0x01010100u * (rnd() & 0xff) | ea[l]; c.sb[l] = 0x01010100u * (rnd() & 0xff) | eb[l];
                for (int j = 0; j < 32; ++j) { av[l][j] = (int)(rnd() % 15) - 7; bv[l][j] = (int)(rnd() % 31) - 15; c.a[l][j] = enc(av[l][j]); c.b[l][j] = enc(bv[l][j]); } }
            for (int r = 0; r < MN; ++r) for (int cc = 0; cc < MN; ++cc) { double acc = 0; for (int g = 0; g < 64 / MN; ++g) { const int la = r + MN * g, lb = cc + MN * g; long sum = 0; for (int j = 0; j < 32; ++j) sum += av[la][j] * bv[lb][j];
                    acc += (double)sum * ldexp(1.0, ea[la] - 127 + eb[lb] - 127); } want[(w * MN + r) * MN + cc] = acc; } }
        if (run<SHAPE>(cs, o)) return 1;
        int bad = 0; double worst = 0;
        for (int w = 0; w < 64; ++w) for (int r = 0; r < MN; ++r) for (int cc = 0; cc < MN; ++cc) { double g = dget<SHAPE>(o, w, r, cc), e = want[(w * MN + r) * MN + cc]; double d = fabs(g - e); if (d > 1e-6 * (fabs(e) + 1)) { if (bad < 5) printf("  mismatch w%d (%d,%d): got %g want %g\n", w, r, cc, g, e); ++bad; } if (d > worst) worst = d; }
        printf("random integers x random scales vs host: %d mismatches of %d, worst abs diff %g\n", bad, 64 * MN * MN, worst);
    }
    return 0;
}
int main() { if (probe<32>()) return 1; if (probe<16>()) return 1; return 0; }
