#!/usr/bin/env python3
"""Host cost of the Python boundary: the same call issued back to back WITHOUT a hipGraph (how BFPLinear.forward really
calls the engine) next to its hipGraph replay.  Per case: eager us/call at three levels of the stack (the ctypes binding
native.quantize_nm, the reference-shaped bfp_ops.float_to_bfp_blocked, and a BFPLinear.forward), and the graph figure.
A call is asynchronous, so eager us/call = max(host issue time, kernel time); with the kernel time known from the graph
the host share is visible.  Writes a markdown table to stdout and JSON to --out."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quantization_sparsity_interplay_amd as pkg
from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops

DEV = torch.device("cuda:0")


def cfg(**kw):
    base = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cuda', block_size=64, num_format='bfp',
                weight_mant_bits=15, in_sparsity=False, w_sparsity=True, grad_sparsity=False, sparsity_frac=0.5, N=2, M=4,
                sparsity_num_format='bfp', first='s', sparsity_mode='structured')
    base.update(kw)
    return base


def time_eager(fn, n, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / n, t_issue * 1e6 / n


def time_graph(fn, n):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "eager.json"))
    ap.add_argument("--n", type=int, default=300)
    args = ap.parse_args()
    cases = [
        ("headline [4096,11008] bf16 2:4 HBFP4 b64", 4096, 11008, torch.bfloat16, cfg()),
        ("cfg1 [768,768] f32 HBFP8 b32 dense", 768, 768, torch.float32, cfg(mant_bits=7, block_size=32, w_sparsity=False)),
        ("cfg5 fc1 [4096,1024] f32 HBFP8 b16 1:4", 4096, 1024, torch.float32, cfg(mant_bits=7, block_size=16, N=1, M=4)),
        ("cfg2 [4096,4096] bf16 HBFP4 b64 dense", 4096, 4096, torch.bfloat16, cfg(w_sparsity=False)),
    ]
    rows = []
    for name, r, c, dt, cf in cases:
        R = 4
        xs = [(torch.randn(r, c, generator=torch.Generator().manual_seed(s)) * 0.02).to(dt).to(DEV) for s in range(R)]
        outs = [torch.empty_like(x) for x in xs]
        N, M = (cf['N'], cf['M']) if cf['w_sparsity'] else (0, 0)
        i = [0]

        def f_native():
            k = i[0] = (i[0] + 1) % R
            native.quantize_nm(xs[k], cf['block_size'], cf['mant_bits'], 1e-8, N=N, M=M, out=outs[k])

        def f_ops():
            k = i[0] = (i[0] + 1) % R
            bfp_ops.float_to_bfp_blocked(xs[k], **cf, identifier='w')

        g_us = time_graph(f_native, 100)
        e_nat, i_nat = time_eager(f_native, args.n)
        e_ops, i_ops = time_eager(f_ops, args.n)
        rows.append(dict(case=name, graph_us=g_us, eager_native_us=e_nat, issue_native_us=i_nat, eager_ops_us=e_ops, issue_ops_us=i_ops))
        print(f"{name:48s} graph {g_us:7.2f} | native eager {e_nat:7.2f} (host issue {i_nat:6.2f}) | bfp_ops eager {e_ops:7.2f} (host issue {i_ops:6.2f})", flush=True)
    # BFPLinear.forward, decode-sized activation
    for name, fin, fout, tokens, cache in (("BFPLinear down_proj 1 token (reference semantics)", 11008, 4096, 1, False),
                                           ("BFPLinear down_proj 1 token (weight cache)", 11008, 4096, 1, True),
                                           ("BFPLinear q_proj 16 tokens (weight cache)", 4096, 4096, 16, True)):
        lin = bfp_ops.BFPLinear(fin, fout, bias=False, **cfg(device="cuda")).to(DEV).to(torch.bfloat16).eval()
        if cache:
            lin.enable_weight_cache()
        x = torch.randn(tokens, fin, device=DEV, dtype=torch.bfloat16)
        ref = torch.nn.Linear(fin, fout, bias=False).to(DEV).to(torch.bfloat16)
        with torch.no_grad():
            e_lin, i_lin = time_eager(lambda: lin(x), args.n)
            e_ref, i_ref = time_eager(lambda: ref(x), args.n)
            g_lin = time_graph(lambda: lin(x), 50)
            g_ref = time_graph(lambda: ref(x), 50)
        rows.append(dict(case=name, graph_us=g_lin, eager_us=e_lin, issue_us=i_lin, flinear_graph_us=g_ref, flinear_eager_us=e_ref))
        print(f"{name:48s} graph {g_lin:7.2f} (F.linear {g_ref:6.2f}) | eager {e_lin:7.2f} (host issue {i_lin:6.2f}) | F.linear eager {e_ref:6.2f}", flush=True)
    # prefill from the packed weight / the cached module on the matrix unit: activation image + matrix kernel per call
    native.SHARE_ACT_IMAGE = False
    for name, fin, fout, tokens in (("PackedBFP.linear q_proj 2048 tokens", 4096, 4096, 2048), ("PackedBFP.linear gate_proj 2048 tokens", 4096, 11008, 2048),
                                    ("PackedBFP.linear gate_proj 256 tokens", 4096, 11008, 256)):
        w = (torch.randn(fout, fin, device=DEV) * 0.02).to(torch.bfloat16)
        x = torch.randn(tokens, fin, device=DEV, dtype=torch.bfloat16)
        pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
        lin = bfp_ops.BFPLinear(fin, fout, bias=False, **cfg(device="cuda")).to(DEV).to(torch.bfloat16).eval().enable_weight_cache(matrix_unit=True)
        with torch.no_grad():
            e_p, i_p = time_eager(lambda: pw.linear(x, x_mant_bits=3), args.n)
            g_p = time_graph(lambda: pw.linear(x, x_mant_bits=3), 20)
            e_m, i_m = time_eager(lambda: lin(x), args.n)
            e_ref, i_ref = time_eager(lambda: torch.nn.functional.linear(x, w), args.n)
        rows.append(dict(case=name, graph_us=g_p, eager_us=e_p, issue_us=i_p, module_eager_us=e_m, module_issue_us=i_m, flinear_eager_us=e_ref))
        print(f"{name:48s} graph {g_p:7.2f} | eager {e_p:7.2f} (host issue {i_p:6.2f}) | BFPLinear matrix unit eager {e_m:7.2f} (host issue {i_m:6.2f}) | F.linear eager {e_ref:6.2f}", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
