#!/usr/bin/env python3
"""fp32 unstructured (two-digit selection) on inputs of different tie structure: us per call (hipGraph, 8 rotating inputs)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd.bfp import bfp_ops
from quantization_sparsity_interplay_amd import native
dev = torch.device("cuda:0")
c = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cuda', block_size=64, num_format='bfp', weight_mant_bits=15,
         in_sparsity=False, w_sparsity=True, grad_sparsity=False, sparsity_frac=0.5, N=2, M=4, sparsity_num_format='bfp', first='s',
         sparsity_mode='unstructured')
dense = dict(c, w_sparsity=False)


def make(kind, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(5120, 5120, generator=g, device=dev) * 0.02
    if kind == "already HBFP4 (8 magnitudes per block: low 16 bits all zero)":
        x = bfp_ops.float_to_bfp_blocked(x, **dense, identifier='w')
    elif kind == "bf16 values held in fp32 (low 16 bits all zero, 100 K-element tie classes)":
        x = x.to(torch.bfloat16).float()
    elif kind == "rows on 40 binades of scale":
        x = x * torch.logspace(-20, 20, 5120, base=2.0, device=dev).view(-1, 1)
    return x.contiguous()


for kind in ("randn * 0.02", "already HBFP4 (8 magnitudes per block: low 16 bits all zero)",
             "bf16 values held in fp32 (low 16 bits all zero, 100 K-element tie classes)", "rows on 40 binades of scale"):
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
    print(f"{kind:80s} {statistics.median(ts):8.1f} us", flush=True)
