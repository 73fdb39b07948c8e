#!/usr/bin/env python3
"""Does ONE large tensor gain from being cut into row pieces spread over two lanes (fork / join per call)?  [4096,11008] bf16 2:4 -> HBFP4,
64 calls in a hipGraph over 8 rotating buffers: whole (one launch) against 2 / 3 / 4 pieces through bfpq_fake_quantize_list."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd import native
dev = torch.device("cuda:0")
rows, cols, R, L = int(os.environ.get("ROWS", 4096)), int(os.environ.get("COLS", 11008)), 8, 64
fq = native.FastQuant(64, 3, 1e-8, 2, 4, True)
native.load_library().bfpq_tune(3, 8)            # pieces from 8 MB on get launches of their own
g = torch.Generator(device=dev).manual_seed(7)
ins = [(torch.randn(rows, cols, generator=g, device=dev) * 0.02).to(torch.bfloat16) for _ in range(R)]
outs = [torch.empty_like(x) for x in ins]


def timed(fn):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / L)
    return statistics.median(ts)


def whole():
    for i in range(L):
        fq(ins[i % R], out=outs[i % R])
print(f"whole tensor, one launch per call: {timed(whole):6.2f} us", flush=True)
for P in (2, 3, 4):
    cuts = [rows * p // P for p in range(P + 1)]
    lists = [native.PreparedList(fq, [ins[r][cuts[p]:cuts[p + 1]] for p in range(P)], outs=[outs[r][cuts[p]:cuts[p + 1]] for p in range(P)]) for r in range(R)]
    def pieces():
        for i in range(L):
            lists[i % R].run()
    print(f"{P} row pieces over two lanes, fork / join per call: {timed(pieces):6.2f} us", flush=True)
