#!/usr/bin/env python3
"""A/B of the tile variants of bfpq_hbfp_linear_mx8 (bfpq_tune key 2) on prefill shapes: interleaved rounds in one process,
hipGraph of 10 launches each, medians; results of every variant compared with variant 0 (same k order: bit-identical)."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops

L = native.load_library()
dev = "cuda:0"
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4, 5]
shapes = [(int(a) for a in sh.split('x')) for sh in os.environ['SHAPES'].split(',')] if os.environ.get('SHAPES') else [(2048, 4096, 4096), (2048, 4096, 11008), (2048, 11008, 4096), (8192, 4096, 11008), (512, 4096, 11008), (2048, 5120, 13824)]
for T, K, N in shapes:
    x = (torch.randn(T, K, device=dev)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
    w8, wsc = pw._mx8_image()
    x8, xs = native.quantize_mx8(x, 3)
    outs, graphs = {}, {}
    for v in variants:
        L.bfpq_tune(2, v)
        out = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
        def run():
            native.check(L.bfpq_hbfp_linear_mx8(x8.data_ptr(), xs.data_ptr(), w8.data_ptr(), wsc.data_ptr(), None, out.data_ptr(), T, N, K, 2,
                                               torch.cuda.current_stream().cuda_stream), "mx8")
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                run()
        g.replay(); torch.cuda.synchronize()
        outs[v], graphs[v] = out, g
    ts = {v: [] for v in variants}
    for _ in range(7):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); graphs[v].replay(); e1.record(); torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) * 100)
    flop = 2.0 * T * K * N
    line = f"T={T} K={K} N={N}: "
    for v in variants:
        med = statistics.median(ts[v])
        same = torch.equal(outs[v], outs[variants[0]])
        line += f" v{v} {med:7.1f} us {flop / med / 1e6:6.0f} TF{'' if same else ' DIFF'} |"
    print(line, flush=True)
L.bfpq_tune(2, -1)
