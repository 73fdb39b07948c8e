#!/usr/bin/env python3
"""Layer-streaming perplexity evaluation of a causal LM whose Linear layers run through the BFP engine --
the counterpart of the reference's opt_eval / llama_eval (examples/pytorch/language-modeling/run_opt.py:210-308,
run_llama.py:208-301): capture the inputs of decoder layer 0, then move ONE layer at a time to the GPU, run all
samples through it, move it back; final norm + lm_head + cross-entropy -> ppl = exp(sum nll / (nsamples * seqlen)).

The reference fetches models and WikiText-2 from the hub; this script takes local paths only (no network):
    --model DIR      a local HF causal-LM directory        (default: a small randomly initialised LLaMA)
    --tokens FILE    a torch-saved LongTensor [1, n_tokens] (default: synthetic uniform tokens)
    --config YAML    an `hbfp:` config in the reference's format (default: the package's bfp_config.yaml)
"""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantization_sparsity_interplay_amd as bfpq                         # noqa: E402
from quantization_sparsity_interplay_amd.bfp import bfp_util                # noqa: E402
from quantization_sparsity_interplay_amd.patch import patch_linear_layers   # noqa: E402


class _Catcher(nn.Module):
    """records the positional / keyword inputs of the first decoder layer, then aborts the forward"""

    def __init__(self, inner, store):
        super().__init__()
        self.inner, self.store = inner, store

    def forward(self, hidden, *args, **kwargs):
        self.store.append((hidden.detach(), args, {k: v for k, v in kwargs.items()}))
        raise StopIteration


@torch.no_grad()
def layerwise_perplexity(model, tokens, seqlen, dev, log=print):
    layers = model.model.layers
    nsamples = tokens.numel() // seqlen
    model.model.embed_tokens.to(dev)
    if hasattr(model.model, "rotary_emb"):
        model.model.rotary_emb.to(dev)
    caught = []
    layers[0] = _Catcher(layers[0], caught)
    for i in range(nsamples):
        try:
            model(tokens[:, i * seqlen:(i + 1) * seqlen].to(dev), use_cache=False)
        except StopIteration:
            pass
    layers[0] = layers[0].inner
    model.model.embed_tokens.cpu()
    hidden = [c[0] for c in caught]
    for li, layer in enumerate(layers):                      # one layer at a time on the device
        layer.to(dev)
        for i in range(nsamples):
            _, args, kwargs = caught[i]
            out = layer(hidden[i], *args, **kwargs)
            hidden[i] = out[0] if isinstance(out, tuple) else out
        layer.cpu()
        torch.cuda.empty_cache()
        log(f"layer {li} done")
    model.model.norm.to(dev)
    model.lm_head.to(dev)
    nll = 0.0
    for i in range(nsamples):
        logits = model.lm_head(model.model.norm(hidden[i]))
        shift = logits[:, :-1, :].float()
        labels = tokens[:, i * seqlen:(i + 1) * seqlen][:, 1:].to(dev)
        loss = nn.functional.cross_entropy(shift.reshape(-1, shift.size(-1)), labels.reshape(-1))
        nll += float(loss) * seqlen
    return float(torch.exp(torch.tensor(nll / (nsamples * seqlen))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="")
    ap.add_argument("--tokens", default="")
    ap.add_argument("--config", default="")
    ap.add_argument("--seqlen", type=int, default=128)
    ap.add_argument("--nsamples", type=int, default=8)
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--cache-weights", action="store_true")
    args = ap.parse_args()
    from transformers import AutoModelForCausalLM, LlamaConfig, LlamaForCausalLM
    dt = getattr(torch, args.dtype)
    if args.model:
        model = AutoModelForCausalLM.from_pretrained(args.model, torch_dtype=dt, local_files_only=True)
    else:
        torch.manual_seed(0)
        model = LlamaForCausalLM(LlamaConfig(hidden_size=512, intermediate_size=1408, num_hidden_layers=4, num_attention_heads=8,
                                             num_key_value_heads=8, vocab_size=4096, max_position_embeddings=args.seqlen)).to(dt)
    model.eval()
    bfp_args = bfp_util.get_bfp_args(args.config or None)
    names = patch_linear_layers(model, bfp_args, cache_weights=args.cache_weights)
    print(f"patched {len(names)} Linear layers with {bfp_args}")
    if args.tokens:
        tokens = torch.load(args.tokens, weights_only=True)
    else:
        tokens = torch.randint(0, model.config.vocab_size, (1, args.seqlen * args.nsamples), generator=torch.Generator().manual_seed(1))
    ppl = layerwise_perplexity(model, tokens, args.seqlen, torch.device("cuda:0"))
    print(f"perplexity {ppl:.4f}")


if __name__ == "__main__":
    main()
