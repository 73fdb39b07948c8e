#!/usr/bin/env python3
"""bench_model.py -- one LLaMA-7B-shaped decoder layer (hidden 4096, intermediate 11008, 32 heads, random init, bf16)
through upstream transformers, with its seven Linear layers (a) plain, (b) BFPLinear with the reference's semantics
(activation AND weight re-quantized every call), (c) BFPLinear with the opt-in weight cache, (d) PackedBFPLinear (weights
held as 4-bit codes).  HBFP4 block 64, weights 2:4 (sparsify -> quantize), round-half-even.  The whole model forward
is captured in a hipGraph (falls back to eager timing if capture fails) -- decode (1 token) and prefill (2048 tokens)."""
import copy, json, os, statistics, sys
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import torch
import quantization_sparsity_interplay_amd as bfpq
from quantization_sparsity_interplay_amd.patch import patch_linear_layers, pack_linear_layers


def timeit(fn, iters, rounds=5):
    fn(); torch.cuda.synchronize()
    how = "hipGraph"
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        run = g.replay
    except Exception as e:                                   # noqa: BLE001
        how = f"eager ({type(e).__name__})"
        torch.cuda.synchronize()

        def run():
            for _ in range(iters):
                fn()
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return statistics.median(ts), how


def main():
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=32,
                      vocab_size=1024, max_position_embeddings=2048)
    base = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    args = bfpq.BFPConfig.hbfp(4, 64, w_sparsity=True, N=2, M=4, sparsity_mode='structured', first='s').to_kwargs()
    models = {}
    models["plain"] = copy.deepcopy(base).to("cuda:0")
    m = copy.deepcopy(base); patch_linear_layers(m, args); models["BFPLinear (reference semantics)"] = m.to("cuda:0")
    m = copy.deepcopy(base); patch_linear_layers(m, args, cache_weights=True); models["BFPLinear + weight cache"] = m.to("cuda:0")
    m = copy.deepcopy(base); patch_linear_layers(m, args, cache_weights=True, matrix_unit=True); models["BFPLinear + weight cache + matrix unit"] = m.to("cuda:0")
    m = copy.deepcopy(base).to("cuda:0"); pack_linear_layers(m, args); models["PackedBFPLinear (4-bit weights)"] = m
    rows = []
    for tokens, iters in ((1, 20), (16, 20), (128, 20), (2048, 3)):
        ids = torch.randint(0, 1024, (1, tokens), generator=torch.Generator().manual_seed(1)).to("cuda:0")
        for name, model in models.items():
            with torch.no_grad():
                us, how = timeit(lambda: model(ids, use_cache=False).logits, iters)
            rows.append(dict(tokens=tokens, mode=name, us_per_forward=us, timing=how))
            print(rows[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "model.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
