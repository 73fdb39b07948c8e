#!/usr/bin/env python3
"""LLaMA-13B's 280 Linear weights through the unstructured list form: serial / pipelined, eager / hipGraph (one MI355X)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd.bfp import bfp_ops
from quantization_sparsity_interplay_amd import native

layers = int(os.environ.get("LAYERS", "40"))
shapes = [(5120, 5120)] * 4 * layers + [(13824, 5120)] * 2 * layers + [(5120, 13824)] * layers
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
ws = [(torch.randn(r, c, generator=g, device=dev) * 0.02).to(torch.bfloat16) for r, c in shapes]
numel = sum(w.numel() for w in ws)
c = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cuda', block_size=64, num_format='bfp', weight_mant_bits=15,
         in_sparsity=False, w_sparsity=True, grad_sparsity=False, sparsity_frac=0.5, N=2, M=4, sparsity_num_format='bfp', first='s',
         sparsity_mode='unstructured')
prep = bfp_ops.PreparedMany(ws, identifier='w', **c)
pl = prep._groups[0][1]


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


# single tensor [5120,5120]: hipGraph over 8 rotating inputs
L = native.load_library()
xs = ws[:8] if shapes[0] == (5120, 5120) else [(torch.randn(5120, 5120, generator=g, device=dev) * 0.02).to(torch.bfloat16) for _ in range(8)]
outs = [torch.empty_like(x) for x in xs]
sw = bfp_ops._workspace(dev)
def one():
    for i in range(40):
        native.prune_quantize(xs[i % 8], xs[i % 8].numel() // 2, sw, 64, 3, 1e-8, out=outs[i % 8])
one(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    one()
us = timed(gr.replay) / 40
n1 = 5120 * 5120
print(f"single tensor [5120,5120] two launches {us:7.2f} us  {n1*6/us/80e3:5.1f} % on the two-read 6 B/elem  {n1*4/us/80e3:5.1f} % on 4 B/elem", flush=True)
if os.environ.get("SINGLE_ONLY"):
    sys.exit(0)
for name, fn in (("serial eager", lambda: pl.run(pipelined=False, graph=False)), ("pipelined eager", lambda: pl.run(pipelined=True, graph=False)),
                 ("pipelined, own hipGraph", lambda: pl.run())):
    us = timed(fn)
    print(f"{name:24s} {us/1e3:8.2f} ms  {numel*4/us/1e3:7.0f} GB/s on 4 B/elem ({numel*4/us/80e3:5.1f} %)  {numel*6/us/80e3:5.1f} % on the two-read 6 B/elem", flush=True)
for name, pipe in (("serial hipGraph", False), ("pipelined hipGraph", True)):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        pl.run(pipelined=pipe, graph=False)
    us = timed(gr.replay)
    print(f"{name:24s} {us/1e3:8.2f} ms  {numel*4/us/1e3:7.0f} GB/s on 4 B/elem ({numel*4/us/80e3:5.1f} %)  {numel*6/us/80e3:5.1f} % on the two-read 6 B/elem", flush=True)

# whole tensors alternating over S independent streams (each tensor's two launches back to back on its stream, a workspace per
# stream): no events between the streams except the fork and the join
for S in (2, 3, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    wss = [native.SelectWorkspace(dev) for _ in range(S)]
    outs_all = prep.run()
    torch.cuda.synchronize()

    def indep():
        main = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(main)
        for i, w in enumerate(ws):
            with torch.cuda.stream(streams[i % S]):
                native.prune_quantize(w, w.numel() // 2, wss[i % S], 64, 3, 1e-8, out=outs_all[i])
        for s in streams:
            main.wait_stream(s)
    indep(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        indep()
    us = timed(gr.replay)
    print(f"{S} independent streams, hipGraph {us/1e3:8.2f} ms  {numel*4/us/1e3:7.0f} GB/s on 4 B/elem ({numel*4/us/80e3:5.1f} %)  {numel*6/us/80e3:5.1f} % on the two-read 6 B/elem", flush=True)
    del gr
