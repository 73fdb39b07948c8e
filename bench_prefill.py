#!/usr/bin/env python3
"""bench_prefill.py -- prefill-sized Linear forward from the PACKED weight on one MI355X:
`PackedBFP.linear` on the block-scaled matrix instruction (bfpq_hbfp_linear_mx8) next to the library GEMM on bf16 operands
(what the reference runs on the two fake-quantised tensors, bfp_ops.py:187-190) and to the cached BFPLinear forward.
HBFP4 weights (2:4) and HBFP4 activations, block 64.  hipGraph of `iters` calls, median of 5."""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import torch
import quantization_sparsity_interplay_amd as bfpq
from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops


def timeit(fn, iters=20, rounds=5):
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
    native.SHARE_ACT_IMAGE = False          # the loop re-uses one input tensor: every call must quantize it, as a first projection does
    rows = []
    cfg = bfpq.BFPConfig.hbfp(4, 64, w_sparsity=True, N=2, M=4, sparsity_mode='structured', first='s').to_kwargs()
    shapes = (("q_proj", 2048, 4096, 4096), ("gate_proj", 2048, 4096, 11008), ("down_proj", 2048, 11008, 4096),
              ("gate_proj 8192 tokens", 8192, 4096, 11008), ("gate_proj 512 tokens", 512, 4096, 11008), ("13B gate_proj", 2048, 5120, 13824))
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, tokens, fin, fout in shapes:
        if only and only not in name:
            continue
        x = (torch.randn(tokens, fin, device=dev) * 1.0).to(torch.bfloat16)
        lin = bfp_ops.BFPLinear(fin, fout, False, **dict(cfg)).to(dev).to(torch.bfloat16).eval()
        with torch.no_grad():
            plain = timeit(lambda: torch.nn.functional.linear(x, lin.weight))
            lin.enable_weight_cache()
            cached = timeit(lambda: lin(x))
            pw = bfp_ops.PackedBFP.quantize(lin.weight, 3, 64, N=2, M=4)
            packed = timeit(lambda: pw.linear(x, x_mant_bits=3))
            # the matrix kernel alone, operands prepared
            w8, wsc = pw._mx8_image()
            x8, xs = native.quantize_mx8(x, 3)
            out = torch.empty((tokens, fout), dtype=torch.bfloat16, device=dev)
            L = native.load_library()
            st = torch.cuda.current_stream().cuda_stream

            def gemm_only():
                native.check(L.bfpq_hbfp_linear_mx8(x8.data_ptr(), xs.data_ptr(), w8.data_ptr(), wsc.data_ptr(), None, out.data_ptr(),
                                                   tokens, fout, fin, 2, torch.cuda.current_stream().cuda_stream), "mx8")
            gemm = timeit(gemm_only)
            aq = timeit(lambda: native.quantize_mx8(x, 3))
        flop = 2.0 * tokens * fin * fout
        rows.append(dict(layer=name, tokens=tokens, in_features=fin, out_features=fout, f_linear_us=plain, bfplinear_cached_us=cached,
                         packed_prefill_us=packed, mx8_gemm_us=gemm, act_image_us=aq,
                         f_linear_tflops=flop / plain / 1e6, mx8_gemm_tflops=flop / gemm / 1e6))
        print(rows[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "prefill.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
