// tests/cabi/cabi_check.cpp -- torch-free check of the C ABI (include/bfpq.h): plain hipMalloc'd buffers,
// host-built tables, bfpq_quantize_nm / bfpq_nm_sparsify / the unstructured steps, compared bit-for-bit with the
// CPU oracle (oracle/libbfp_oracle.so -- test infrastructure).  Built and run by tests/test_cabi.py (-m gpu).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bfpq.h"

extern "C" {
int oracle_float_to_bfp_blocked(const void* in, void* out, int64_t rows, int64_t cols, int dtype, int quantize, int block_size,
                                int mant_bits, double epsilon, int sparsity_mode, int N, int M, double frac, int sparsify_first);
int oracle_unstructured_sparsify(const void* in, void* out, int64_t numel, int dtype, double frac, float* tau_out, int64_t* k_out);
}

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define RC(x) do { int r_ = (x); if (r_ != 0) { printf("bfpq error %d (%s) at line %d\n", r_, bfpq_error_string(r_), __LINE__); return 3; } } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }

int main()
{
    const int64_t rows = 512, cols = 1024, n = rows * cols;
    // bf16 test data: sign | exponent in [112,124] | 7 random mantissa bits  (values ~ 1e-5 .. 0.2)
    std::vector<uint16_t> h_in(n), h_out(n), h_ref(n);
    for (int64_t i = 0; i < n; i++) {
        const uint32_t r = rnd();
        h_in[i] = (uint16_t)(((r & 1u) << 15) | ((112u + (r >> 8) % 13u) << 7) | ((r >> 16) & 0x7fu));
    }
    uint8_t win[BFPQ_EXP_WIN_ENTRIES], lut[BFPQ_NM4_LUT_ENTRIES];
    RC(bfpq_exp_window_host(BFPQ_BF16, win));
    RC(bfpq_nm4_lut_host(2, lut));
    void *d_in, *d_out; uint8_t *d_win, *d_lut, *d_codes; int8_t* d_exp;
    HIP(hipMalloc(&d_in, n * 2)); HIP(hipMalloc(&d_out, n * 2));
    HIP(hipMalloc(&d_win, sizeof win)); HIP(hipMalloc(&d_lut, sizeof lut));
    HIP(hipMalloc(&d_codes, n / 2)); HIP(hipMalloc(&d_exp, n / 64));
    HIP(hipMemcpy(d_in, h_in.data(), n * 2, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_win, win, sizeof win, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
    hipStream_t s; HIP(hipStreamCreate(&s));
    int fails = 0;

    // 1. headline configuration, both orders: drop-in output vs oracle
    for (int sfirst = 1; sfirst >= 0; sfirst--) {
        RC(bfpq_quantize_nm(d_in, d_out, nullptr, nullptr, rows, cols, BFPQ_BF16, 64, 3, 1e-8, 2, 4, sfirst, 0, 0, d_win, d_lut, nullptr, s));
        HIP(hipStreamSynchronize(s));
        HIP(hipMemcpy(h_out.data(), d_out, n * 2, hipMemcpyDeviceToHost));
        if (oracle_float_to_bfp_blocked(h_in.data(), h_ref.data(), rows, cols, BFPQ_BF16, 1, 64, 3, 1e-8, 1, 2, 4, 0.5, sfirst)) return 4;
        const bool ok = memcmp(h_out.data(), h_ref.data(), n * 2) == 0;
        printf("quantize_nm bf16 2:4 HBFP4 b64 first=%s : %s\n", sfirst ? "s" : "q", ok ? "bit-exact" : "MISMATCH");
        fails += !ok;
    }
    // 2. packed outputs decode to the same values
    {
        RC(bfpq_quantize_nm(d_in, d_out, d_codes, d_exp, rows, cols, BFPQ_BF16, 64, 3, 1e-8, 2, 4, 1, 4, 0, d_win, d_lut, nullptr, s));
        HIP(hipStreamSynchronize(s));
        std::vector<uint8_t> codes(n / 2); std::vector<int8_t> ex(n / 64);
        HIP(hipMemcpy(codes.data(), d_codes, n / 2, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(ex.data(), d_exp, n / 64, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(h_out.data(), d_out, n * 2, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int64_t i = 0; i < n; i++) {
            int c = (codes[i / 2] >> (4 * (i & 1))) & 0xf; if (c > 7) c -= 16;
            const float v = (float)c * ldexpf(1.0f, ex[i / 64] - 3);
            uint32_t u; memcpy(&u, &v, 4);
            bad += (uint16_t)(u >> 16) != (uint16_t)(h_out[i] & (c == 0 ? 0x7fff : 0xffff)) && !(c == 0 && (h_out[i] & 0x7fff) == 0);
        }
        printf("packed int4 codes + int8 exponents decode : %s\n", bad == 0 ? "ok" : "MISMATCH");
        fails += bad != 0;
    }
    // 3. unstructured: threshold / count equal to the oracle's, elements below the threshold zeroed
    {
        void* d_state;                                      // the select workspace begins with the bfpq_select_state
        HIP(hipMalloc(&d_state, BFPQ_SELECT_WS_BYTES));
        HIP(hipMemset(d_state, 0, BFPQ_SELECT_WS_BYTES));
        const int64_t k = (int64_t)((double)n * 0.5);
        float tau; int64_t kk;
        if (oracle_unstructured_sparsify(h_in.data(), h_ref.data(), n, BFPQ_BF16, 0.5, &tau, &kk)) return 5;
        uint32_t tb; memcpy(&tb, &tau, 4);
        bfpq_select_state st;
        int64_t zeros = 0, zref = 0, diff_outside = 0;
        std::vector<uint16_t> h_first;
        bool same_both = true;
        for (int form = 0; form < 2; form++) {              // 0: the one-launch selection; 1: the launch-pair form (what multi-GPU callers issue)
            HIP(hipMemset(d_out, 0xff, n * 2));
            if (form == 0) RC(bfpq_select(d_in, n, BFPQ_BF16, k, d_state, s));
            else
                for (int p = 0; p < bfpq_select_passes(BFPQ_BF16); p++) {       // single device: the workspace's own histogram buffers
                    RC(bfpq_select_hist(d_in, n, BFPQ_BF16, p, k, n, d_state, nullptr, s));
                    RC(bfpq_select_resolve(d_in, n, BFPQ_BF16, p, k, nullptr, 1, 0, d_state, nullptr, s));
                }
            RC(bfpq_threshold_apply(d_in, d_out, n, BFPQ_BF16, d_state, s));
            HIP(hipStreamSynchronize(s));
            HIP(hipMemcpy(&st, d_state, sizeof st, hipMemcpyDeviceToHost));
            HIP(hipMemcpy(h_out.data(), d_out, n * 2, hipMemcpyDeviceToHost));
            if (form == 0) h_first = h_out; else same_both = h_first == h_out;
            zeros = zref = diff_outside = 0;
            for (int64_t i = 0; i < n; i++) {
                zeros += (h_out[i] & 0x7fff) == 0; zref += (h_ref[i] & 0x7fff) == 0;
                if ((uint32_t)(h_in[i] & 0x7fff) != (tb >> 16)) diff_outside += h_out[i] != h_ref[i];
            }
            if (!(st.tau == (tb >> 16) && st.k == kk && zeros == zref && diff_outside == 0 && st.done == 1)) break;
        }
        fails += !same_both;
        const bool ok = st.tau == (tb >> 16) && st.k == kk && zeros == zref && diff_outside == 0 && st.done == 1;
        printf("unstructured 50%% : tau 0x%x (oracle 0x%x) zeros %lld (oracle %lld) outside-tie diffs %lld : %s\n", st.tau, tb >> 16,
               (long long)zeros, (long long)zref, (long long)diff_outside, ok ? "ok" : "MISMATCH");
        fails += !ok;
    }
    // 3b. prefill consumer: out[T, N] = Q(x) Q(w)^T on the block-scaled matrix instruction, against the double-precision product of
    //     the oracle's two fake-quantised tensors (w = the first 256 rows of the test data: 2:4 -> HBFP4; x = 200 other rows: dense HBFP4)
    {
        const int64_t T = 200, N = 256, K = cols;
        const uint16_t* hx = h_in.data() + 300 * cols;
        std::vector<uint16_t> wq(N * K), xq(T * K);
        if (oracle_float_to_bfp_blocked(h_in.data(), wq.data(), N, K, BFPQ_BF16, 1, 64, 3, 1e-8, 1, 2, 4, 0.5, 1)) return 6;
        if (oracle_float_to_bfp_blocked(hx, xq.data(), T, K, BFPQ_BF16, 1, 64, 3, 1e-8, 0, 0, 0, 0.5, 1)) return 6;
        uint8_t *d_w8, *d_ws, *d_x8, *d_xs; float* d_o;
        HIP(hipMalloc(&d_w8, N * K)); HIP(hipMalloc(&d_ws, N * K / 64)); HIP(hipMalloc(&d_x8, T * K)); HIP(hipMalloc(&d_xs, T * K / 64)); HIP(hipMalloc(&d_o, T * N * 4));
        RC(bfpq_quantize_nm(d_in, nullptr, d_codes, d_exp, N, K, BFPQ_BF16, 64, 3, 1e-8, 2, 4, 1, 4, 0, d_win, d_lut, nullptr, s));     // the packed weight
        RC(bfpq_mx8_from_hbfp(d_codes, d_exp, d_w8, d_ws, N, K, 4, 3, s));                                                          // its image, once
        RC(bfpq_quantize_mx8((const uint16_t*)d_in + 300 * cols, d_x8, d_xs, T, K, BFPQ_BF16, 3, 1e-8, d_win, s));                 // the activation's image, one pass
        RC(bfpq_hbfp_linear_mx8(d_x8, d_xs, d_w8, d_ws, nullptr, d_o, T, N, K, BFPQ_F32, s));
        HIP(hipStreamSynchronize(s));
        std::vector<float> o(T * N);
        HIP(hipMemcpy(o.data(), d_o, T * N * 4, hipMemcpyDeviceToHost));
        auto f = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float v; memcpy(&v, &u, 4); return (double)v; };
        double worst = 0, big = 0;
        for (int64_t t = 0; t < T; t++)
            for (int64_t j = 0; j < N; j++) {
                double acc = 0;
                for (int64_t k = 0; k < K; k++) acc += f(xq[t * K + k]) * f(wq[j * K + k]);
                const double d = acc - (double)o[t * N + j];
                if ((d < 0 ? -d : d) > worst) worst = d < 0 ? -d : d;
                if ((acc < 0 ? -acc : acc) > big) big = acc < 0 ? -acc : acc;
            }
        const bool ok = worst <= 2e-6 * big;
        printf("prefill from the packed weight (block-scaled matrix instruction) : max error %.3g of %.3g : %s\n", worst, big, ok ? "ok" : "MISMATCH");
        fails += !ok;
    }
    // 4. argument errors come back as codes, nothing is launched
    fails += bfpq_quantize_nm(nullptr, d_out, nullptr, nullptr, rows, cols, BFPQ_BF16, 64, 3, 1e-8, 2, 4, 1, 0, 0, d_win, d_lut, nullptr, s) != BFPQ_E_ARG;
    fails += bfpq_quantize_nm(d_in, d_out, nullptr, nullptr, rows, cols, 7, 64, 3, 1e-8, 2, 4, 1, 0, 0, d_win, d_lut, nullptr, s) != BFPQ_E_ARG;
    fails += bfpq_quantize_nm(d_in, d_out, nullptr, nullptr, rows, cols, BFPQ_BF16, 64, 3, 1e-8, 5, 4, 1, 0, 0, d_win, d_lut, nullptr, s) != BFPQ_E_ARG;
    fails += bfpq_quantize_nm(d_in, d_out, nullptr, nullptr, rows, cols, BFPQ_BF16, 64, 3, 1e-8, 2, 128, 1, 0, 0, d_win, d_lut, nullptr, s) != BFPQ_E_UNSUPPORTED;
    printf("argument validation : %s\n", fails ? "see above" : "ok");
    printf(fails ? "CABI CHECK FAILED (%d)\n" : "CABI CHECK PASSED\n", fails);
    return fails ? 1 : 0;
}
