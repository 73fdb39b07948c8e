"""fp32 unstructured on an already half-sparse tensor (the generic selection loop): us per call, hipGraph over 8 rotating inputs"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_sparsity_interplay_amd.bfp import bfp_ops
dev = torch.device("cuda:0")
c = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cuda', block_size=64, num_format='bfp', weight_mant_bits=15, in_sparsity=False, w_sparsity=True,
         grad_sparsity=False, sparsity_frac=0.5, N=2, M=4, sparsity_num_format='bfp', first='s', sparsity_mode='unstructured')
for kind, keep in (("dense", 1.1), ("52 % zeros", 0.48), ("30 % zeros", 0.7)):
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.randn(5120, 5120, generator=g, device=dev) * 0.02 * (torch.rand(5120, 5120, generator=g, device=dev) < keep) for _ in range(8)]
    def run():
        for i in range(16): bfp_ops.float_to_bfp_blocked(xs[i % 8], **c, identifier='w')
    run(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): run()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 16)
    print(f"fp32 [5120,5120] {kind:12s} {statistics.median(ts):8.1f} us", flush=True)
