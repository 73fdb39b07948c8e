#!/usr/bin/env python3
"""Why is a whole-model pass (cfg3: 60.9 %) below the single-tensor figure (68 %)?  Suspects, separated here per shape
(bf16, 2:4 -> HBFP4): (a) the memory footprint of the pass (R distinct in/out pairs: address translation), (b) the list kernel
itself (k_fused_batched against one k_fused_flat launch per tensor at the SAME footprint), and what two streams buy for
per-tensor launches (the tail of one tensor beside the ramp of the next).  us per tensor, median of 5, 64 tensors per pass."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd import native

dev = torch.device("cuda:0")
fq = native.FastQuant(64, 3, 1e-8, 2, 4, True)
L = 64


def timed(fn, per):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / per)
    return statistics.median(ts)


sides = [torch.cuda.Stream(device=dev) for _ in range(3)]
for rows, cols in ((4096, 11008), (4096, 4096), (1024, 4096)):
    for R in (8, 64):
        g = torch.Generator(device=dev).manual_seed(7)
        ins = [(torch.randn(rows, cols, generator=g, device=dev) * 0.02).to(torch.bfloat16) for _ in range(R)]
        outs = [torch.empty_like(x) for x in ins]

        def lanes(S):
            def f():
                main = torch.cuda.current_stream()
                for s in sides[:S - 1]:
                    s.wait_stream(main)
                for i in range(L):
                    if i % S:
                        with torch.cuda.stream(sides[i % S - 1]):
                            fq(ins[i % R], out=outs[i % R])
                    else:
                        fq(ins[i % R], out=outs[i % R])
                for s in sides[:S - 1]:
                    main.wait_stream(s)
            return f
        res = []
        for S in (1, 2, 3, 4):
            f = lanes(S)
            f(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                f()
            res.append(timed(gr.replay, L))
            del gr
        pl = native.PreparedList(fq, [ins[i % R] for i in range(L)], outs=[outs[i % R] for i in range(L)])
        pl._aux = None
        t_list = timed(pl.run, L)
        print(f"[{rows},{cols}] R = {R:3d} pairs ({R * rows * cols * 4 / 1e9:5.1f} GB): one launch per tensor over 1 / 2 / 3 / 4 streams "
              f"{res[0]:6.2f} / {res[1]:6.2f} / {res[2]:6.2f} / {res[3]:6.2f} us, the list call on one stream {t_list:6.2f} us / tensor", flush=True)
        del ins, outs, pl
        torch.cuda.empty_cache()
