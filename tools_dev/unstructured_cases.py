#!/usr/bin/env python3
"""The unstructured path (bf16 [5120,5120], 50 % -> HBFP4) on inputs whose threshold sits in awkward places: us per call
(hipGraph, 8 rotating inputs), and the result checked against a plain torch restatement of the contract (count, threshold)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd.bfp import bfp_ops
dev = torch.device("cuda:0")
c = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cuda', block_size=64, num_format='bfp', weight_mant_bits=15,
         in_sparsity=False, w_sparsity=True, grad_sparsity=False, sparsity_frac=0.5, N=2, M=4, sparsity_num_format='bfp', first='s',
         sparsity_mode='unstructured')


def make(kind, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(5120, 5120, generator=g, device=dev) * 0.02
    if kind == "half of the elements already zero (random places)":
        x = x * (torch.rand(5120, 5120, generator=g, device=dev) < 0.5)
    elif kind == "52 % already zero":
        x = x * (torch.rand(5120, 5120, generator=g, device=dev) < 0.48)
    elif kind == "48 % already zero":
        x = x * (torch.rand(5120, 5120, generator=g, device=dev) < 0.52)
    elif kind == "already pruned to exactly 50 % by this engine":
        p = dict(c, sparsity_num_format='fp32')
        x = bfp_ops.float_to_bfp_blocked(x.to(torch.bfloat16), **p, identifier='w').float()
    elif kind == "already HBFP4 values (the quantizer ran first: ~140 distinct magnitudes, low mantissa bits zero)":
        p = dict(c, w_sparsity=False)
        x = bfp_ops.float_to_bfp_blocked(x.to(torch.bfloat16), **p, identifier='w').float()
    elif kind == "rows on 8 binades of scale":
        x = x * torch.logspace(-4, 4, 5120, base=2.0, device=dev).view(-1, 1)
    elif kind == "two populations (half the rows x 1000)":
        x[::2] *= 1000.0
    return x.to(torch.bfloat16).contiguous()


for kind in ("randn * 0.02", "half of the elements already zero (random places)", "52 % already zero", "48 % already zero",
             "already pruned to exactly 50 % by this engine", "already HBFP4 values (the quantizer ran first: ~140 distinct magnitudes, low mantissa bits zero)",
             "rows on 8 binades of scale", "two populations (half the rows x 1000)"):
    xs = [make(kind, 1 + i) for i in range(8)]
    def run():
        for i in range(16):
            bfp_ops.float_to_bfp_blocked(xs[i % 8], **c, identifier='w')
    run(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 16)
    y = bfp_ops.float_to_bfp_blocked(xs[0], **dict(c, sparsity_num_format='fp32'), identifier='w')
    k = xs[0].numel() // 2
    mag = xs[0].float().abs().flatten()
    tau = mag.kthvalue(k).values
    kept_min = mag[(y.flatten() != 0)].min() if (y != 0).any() else torch.tensor(float('inf'))
    ok = int((y == 0).sum()) >= k and float(kept_min) >= float(tau) and int(((y == 0).flatten() & (mag > tau)).sum()) == 0
    print(f"{kind[:60]:60s} {statistics.median(ts):9.1f} us   contract {'ok' if ok else 'VIOLATED'}", flush=True)
