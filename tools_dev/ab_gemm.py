"""A/B of the tiled decode kernel alone (activation codes prepared once): row tiles per wave (x reuse) by token count."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops

L = native.load_library()


def timeit(fn, iters=100, rounds=7):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return statistics.median(ts)


for N, K in ((4096, 11008), (11008, 4096), (4096, 4096), (8192, 28672)):
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    x = torch.randn(16, K, device="cuda").to(torch.bfloat16)
    pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
    tiles, expt = native.mfma_tiles(pw.codes, pw.exps)
    xc = torch.empty((16, K), dtype=torch.int8, device="cuda"); xe = torch.empty((16, K // 64), dtype=torch.int8, device="cuda")
    native.quantize_nm(x, 64, 7, 1e-8, want_deq=False, code_bits=8, want_exp=True, codes_out=xc, exps_out=xe)
    out = torch.empty((16, N), dtype=torch.bfloat16, device="cuda")
    res = {}
    for T in (1, 4, 16):
      for mode in (1, 2, 4):
        L.bfpq_tune(1, mode)
        fn = lambda: L.bfpq_hbfp_linear_decode_tiled(tiles.data_ptr(), expt.data_ptr(), xc.data_ptr(), xe.data_ptr(), out.data_ptr(), T, N, K, 2, 3, 7, torch.cuda.current_stream().cuda_stream)
        res[mode] = timeit(fn)
      print(f"   T={T}: RT1 {res[1]:.2f}  RT2 {res[2]:.2f}  RT4 {res[4]:.2f} us", flush=True)
    L.bfpq_tune(1, 0)
    auto = timeit(fn)
    lin = timeit(lambda: torch.nn.functional.linear(x, w))
    mb = (N * K / 2 + N * K / 64) / 1e6
    print(f"N={N} K={K} packed={mb:.1f}MB  T=16 auto {auto:.2f} us ({mb / auto:.2f} TB/s)  F.linear bf16 {lin:.2f} us", flush=True)
