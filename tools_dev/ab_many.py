#!/usr/bin/env python3
"""Lists of large tensors through configurations without a list kernel: tensor after tensor against bfp_ops._many_over_streams
(us per tensor, 16 x [4096,11008] bf16, eager, median of 5)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd.bfp import bfp_ops
dev = torch.device("cuda:0")
base = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cuda', block_size=64, num_format='bfp', weight_mant_bits=15, in_sparsity=False,
            w_sparsity=True, grad_sparsity=False, sparsity_frac=0.5, N=2, M=4, sparsity_num_format='bfp', first='s', sparsity_mode='structured')
g = torch.Generator(device=dev).manual_seed(3)
xs = [(torch.randn(4096, 11008, generator=g, device=dev) * 0.02).to(torch.bfloat16) for _ in range(16)]


def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / len(xs))
    return statistics.median(ts)


for name, kw in (("unstructured 50 %, quantize first (q -> s)", dict(sparsity_mode='unstructured', first='q')),
                 ("stochastic rounding, 2:4 s", dict(rounding_mode='stoc')),
                 ("int8 per-row weights", dict(sparsity_num_format='int', mant_bits=8, w_sparsity=False)),
                 ("4:8 s", dict(N=4, M=8))):
    c = dict(base, **kw)
    res = []
    for lanes in (1, 2, 3, 4):
        bfp_ops.MANY_LANES = lanes
        res.append(timed(lambda: bfp_ops.float_to_bfp_blocked_many(xs, identifier='w', **c)))
    print(f"{name:44s} streams 1 / 2 / 3 / 4: " + " / ".join(f"{r:6.1f}" for r in res) + " us per tensor", flush=True)
