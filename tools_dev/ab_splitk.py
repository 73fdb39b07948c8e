#!/usr/bin/env python3
"""K split of the prefill kernel at short token counts: native.hbfp_linear_mx8 with and without it (the activation image included),
next to F.linear on bf16 operands; hipGraph of 20 calls, medians of interleaved rounds."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops
dev = "cuda:0"
native.SHARE_ACT_IMAGE = False
for T, K, N in [(128, 11008, 4096), (128, 4096, 11008), (256, 11008, 4096), (96, 4096, 4096), (128, 13824, 5120), (200, 4096, 4096), (512, 4096, 4096), (512, 11008, 4096),
                (384, 4096, 4096), (768, 4096, 4096), (512, 4096, 11008)]:
    x = torch.randn(T, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
    w8, wsc = pw._mx8_image()
    graphs = {}
    L = native.load_library()
    for name in ("split", "one", "plan5", "F.linear"):
        def run(name=name):
            if name == "F.linear":
                return torch.nn.functional.linear(x, w)
            native.SPLIT_K = name == "split"
            L.bfpq_tune(2, 5 if name == "plan5" else -1)               # plan5: 256 x 256 / 256 x 128 ring tiles whatever the shape
            r = native.hbfp_linear_mx8(x, w8, wsc, 3)
            L.bfpq_tune(2, -1)
            return r
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                run()
        g.replay(); torch.cuda.synchronize()
        graphs[name] = g
    native.SPLIT_K = True
    ts = {k: [] for k in graphs}
    for _ in range(7):
        for k, g in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) * 50)
    parts = native.load_library().bfpq_hbfp_linear_mx8_parts(T, N, K)
    print(f"T={T} K={K} N={N} parts={parts}: " + " | ".join(f"{k} {statistics.median(v):6.1f} us" for k, v in ts.items()), flush=True)
