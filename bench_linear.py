#!/usr/bin/env python3
"""bench_linear.py -- end-to-end cost of one patched Linear forward at LLaMA-7B shapes on one MI355X:
plain F.linear, BFPLinear with the reference's semantics (activation AND weight re-quantized every call,
bfp_ops.py:151-166) and BFPLinear with the opt-in quantized-weight cache.  Writes a markdown table."""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import torch
import quantization_sparsity_interplay_amd as bfpq
from quantization_sparsity_interplay_amd.bfp import bfp_ops
from quantization_sparsity_interplay_amd import native


def timeit(fn, iters=50, rounds=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return statistics.median(ts)


def main():
    dev = "cuda:0"
    rows = []
    cfg = bfpq.BFPConfig.hbfp(4, 64, w_sparsity=True, N=2, M=4, sparsity_mode='structured', first='s').to_kwargs()
    for name, tokens, fin, fout in (("q_proj", 2048, 4096, 4096), ("gate_proj", 2048, 4096, 11008), ("down_proj", 2048, 11008, 4096),
                                    ("down_proj decode", 16, 11008, 4096), ("gate_proj decode", 16, 4096, 11008), ("down_proj decode 1 token", 1, 11008, 4096),
                                    ("down_proj decode 64 tokens", 64, 11008, 4096)):
        x = (torch.randn(tokens, fin, device=dev) * 1.0).to(torch.bfloat16)
        lin = bfp_ops.BFPLinear(fin, fout, False, **dict(cfg)).to(dev).to(torch.bfloat16).eval()     # (the weight cache is inference-only)
        with torch.no_grad():
            plain = timeit(lambda: torch.nn.functional.linear(x, lin.weight))
            ref = timeit(lambda: lin(x))
            lin.enable_weight_cache()
            cached = timeit(lambda: lin(x))
            wq = timeit(lambda: bfp_ops.float_to_bfp_blocked(lin.weight, **cfg, identifier='w'))
            aq = timeit(lambda: bfp_ops.float_to_bfp_blocked(x, **cfg, identifier='in'))
            # from the packed weight: <= 64 tokens the integer block-dot-product kernel (HBFP8 activations), more tokens the block-scaled
            # matrix instruction (HBFP4 activations, as the module's configuration says); every call quantizes its activation
            native.SHARE_ACT_IMAGE = False
            pw = bfp_ops.PackedBFP.quantize(lin.weight, 3, 64, N=2, M=4)
            packed = timeit(lambda: pw.linear(x)) if tokens <= 64 else timeit(lambda: pw.linear(x, x_mant_bits=3))
        rows.append(dict(layer=name, tokens=tokens, in_features=fin, out_features=fout, f_linear_us=plain, bfplinear_us=ref,
                         bfplinear_cached_us=cached, weight_quant_us=wq, act_quant_us=aq, packed_decode_us=packed))
        print(rows[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "linear.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
