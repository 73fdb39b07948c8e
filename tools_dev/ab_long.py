#!/usr/bin/env python3
"""Does a LONG sweep of the flat kernel lose speed (workgroups drifting apart)?  One [R*4096, 11008] bf16 tensor (cut into 2 GiB
launches by the library) against R launches of [4096,11008]; and the list kernel on the same bytes as R descriptors."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd import native
dev = torch.device("cuda:0")
fq = native.FastQuant(64, 3, 1e-8, 2, 4, True)


def timed(fn, per):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / per)
    return statistics.median(ts)


for R in (8, 20, 64):
    x = torch.empty(R * 4096, 11008, dtype=torch.bfloat16, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    for r in range(R):
        x[r * 4096:(r + 1) * 4096] = (torch.randn(4096, 11008, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    y = torch.empty_like(x)
    t_one = timed(lambda: fq(x, out=y), R)
    parts = [x[r * 4096:(r + 1) * 4096] for r in range(R)]
    outs = [y[r * 4096:(r + 1) * 4096] for r in range(R)]
    def flat():
        for a, b in zip(parts, outs):
            fq(a, out=b)
    t_flat = timed(flat, R)
    pl = native.PreparedList(fq, parts, outs=outs)
    pl._aux = None
    t_l1 = timed(pl.run, R)
    pl2 = native.PreparedList(fq, parts, outs=outs)
    t_l2 = timed(pl2.run, R)
    print(f"{R:3d} x [4096,11008]: ONE tensor {t_one:6.2f} us per 4096 rows | {R} launches eager {t_flat:6.2f} | list call, one lane {t_l1:6.2f} | two lanes {t_l2:6.2f}", flush=True)
    del x, y, parts, outs, pl, pl2
    torch.cuda.empty_cache()
