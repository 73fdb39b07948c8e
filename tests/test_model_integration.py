"""Drop-in at model level: the Linear layers of a small randomly-initialised upstream LLaMA (the model family of
BASELINE configs 2-4) become BFPLinear; every patched layer's GPU output is checked against
F.linear(oracle(x, 'in'), oracle(W, 'w')) computed from the very input that layer saw (no error accumulation
across layers), and the layer-streaming perplexity harness of examples/ runs end to end."""
import os
import sys

import pytest
import torch

import quantization_sparsity_interplay_amd as pkg
from quantization_sparsity_interplay_amd.bfp import bfp_ops
from quantization_sparsity_interplay_amd.patch import patch_linear_layers, pack_linear_layers, prime_weight_caches, PackedBFPLinear
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_llama(dtype):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=1000, max_position_embeddings=64)
    return LlamaForCausalLM(cfg).to(dtype).eval()


def test_patch_keeps_state_dict_and_skips_lm_head():
    model = _tiny_llama(torch.float32)
    keys = sorted(model.state_dict().keys())
    args = pkg.BFPConfig.hbfp(8, 32, w_sparsity=True, N=2, M=4, sparsity_mode='structured').to_kwargs()
    names = patch_linear_layers(model, args)
    assert len(names) == 2 * 7 and all(n.endswith("_proj") for n in names)
    assert type(model.lm_head) is torch.nn.Linear
    assert sorted(model.state_dict().keys()) == keys                      # stock checkpoints still load
    assert isinstance(model.model.layers[0].mlp.down_proj, bfp_ops.BFPLinear)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patched_llama_layers_match_oracle(dtype):
    model = _tiny_llama(dtype)
    args = pkg.BFPConfig.hbfp(8, 32, w_sparsity=True, N=2, M=4, sparsity_mode='structured', in_sparsity=False).to_kwargs()
    names = patch_linear_layers(model, args)
    model.to("cuda:0")
    seen = {}
    hooks = []
    for n, mod in model.named_modules():
        if isinstance(mod, bfp_ops.BFPLinear):
            hooks.append(mod.register_forward_hook(lambda m, inp, out, n=n: seen.__setitem__(n, (inp[0].detach().cpu(), out.detach().cpu()))))
    tokens = torch.randint(0, 1000, (2, 48), generator=torch.Generator().manual_seed(3)).to("cuda:0")
    with torch.no_grad():
        logits = model(tokens).logits
    assert torch.isfinite(logits.float()).all()
    assert set(seen) == set(names)
    for n, (x, y) in seen.items():
        mod = dict(model.named_modules())[n]
        w = mod.weight.detach().cpu()
        xq = O.float_to_bfp_blocked(x, **args, identifier='in')
        wq = O.float_to_bfp_blocked(w, **args, identifier='w')
        want = torch.nn.functional.linear(xq.float(), wq.float())
        tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(y.float(), want, **tol, msg=lambda m, n=n: f"{n}: {m}")
        # and the operands themselves are bit-exact
        assert torch.equal(bfp_ops.float_to_bfp_blocked(mod.weight.detach(), **args, identifier='w').cpu().view(torch.int32 if dtype == torch.float32 else torch.int16),
                           wq.view(torch.int32 if dtype == torch.float32 else torch.int16)), n
    for h in hooks:
        h.remove()


@pytest.mark.gpu
def test_prime_weight_caches_in_one_list_call():
    """patch.prime_weight_caches: every cached BFPLinear's weight through ONE list call; the first forward then hits every cache and gives
    what the un-primed model gives; a model in training mode is left alone (the cache is inference-only)."""
    args = pkg.BFPConfig.hbfp(4, 64, w_sparsity=True, N=2, M=4, sparsity_mode='structured').to_kwargs()
    ids = torch.randint(0, 1000, (2, 16), device="cuda")
    plain = _tiny_llama(torch.bfloat16).cuda()
    patch_linear_layers(plain, args)
    with torch.no_grad():
        want = plain(ids).logits
    model = _tiny_llama(torch.bfloat16).cuda()
    patch_linear_layers(model, args, cache_weights=True)
    mods = [m for m in model.modules() if isinstance(m, bfp_ops.BFPLinear)]
    assert prime_weight_caches(model) == len(mods) == 14
    with torch.no_grad():
        got = model(ids).logits
    assert all(m._weight_cache.hits == 1 and m._weight_cache.misses == 1 for m in mods)        # (the one miss is the priming store)
    assert torch.equal(got, want)
    model.train()
    for m in mods:
        m._weight_cache.invalidate()
    assert prime_weight_caches(model) == 0


@pytest.mark.gpu
def test_packed_inference_model_matches_fake_quantised_model():
    """deployment form: the same LLaMA with its Linear weights held as 4-bit codes (PackedBFPLinear) against the
    BFPLinear (fake-quantise + F.linear) model -- decode-sized and prefill-sized inputs"""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=768, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=1000, max_position_embeddings=64)
    base = LlamaForCausalLM(cfg).to(torch.float32).eval()
    args = pkg.BFPConfig.hbfp(4, 64, w_sparsity=True, N=2, M=4, sparsity_mode='structured', first='s').to_kwargs()
    args['rounding_mode'] = 'determ'
    fake, packed = copy.deepcopy(base), copy.deepcopy(base)
    patch_linear_layers(fake, args)
    fake.to("cuda:0"); packed.to("cuda:0")
    names, saved = pack_linear_layers(packed, args)
    assert len(names) == 2 * 7 and saved > 0
    assert isinstance(packed.model.layers[1].mlp.down_proj, PackedBFPLinear)
    assert type(packed.lm_head) is torch.nn.Linear
    for shape in ((1, 12), (2, 48)):                                      # 12 tokens: integer block dot products; 96: decode + GEMM
        tokens = torch.randint(0, 1000, shape, generator=torch.Generator().manual_seed(5)).to("cuda:0")
        with torch.no_grad():
            a, b = fake(tokens).logits, packed(tokens).logits
        err = float((a - b).abs().max() / a.abs().max())
        assert err < 2e-3, (shape, err)                                     # same quantised operands; accumulation order differs
    # the packed weights are buffers: they are in the state dict, survive a round trip through it, and follow .to()
    sd = packed.state_dict()
    key = "model.layers.1.mlp.down_proj.codes"
    assert key in sd and sd[key].dtype == torch.uint8 and "model.layers.1.mlp.down_proj.exps" in sd
    clone = copy.deepcopy(base).to("cuda:0")
    pack_linear_layers(clone, args)
    for m in clone.modules():
        if isinstance(m, PackedBFPLinear):
            m.codes.zero_()                                                    # wipe, then restore from the state dict
    clone.load_state_dict(sd)
    tokens = torch.randint(0, 1000, (1, 12), generator=torch.Generator().manual_seed(5)).to("cuda:0")
    with torch.no_grad():
        assert torch.equal(clone(tokens).logits, packed(tokens).logits)
    with pytest.raises(ValueError):                                           # pruning after quantization is not packed
        PackedBFPLinear.from_linear(torch.nn.Linear(256, 64).to("cuda:0"), dict(args, sparsity_mode='unstructured', sparsity_frac=0.5, first='q'))
    # unstructured pruning before quantization IS: the packed module equals the BFPLinear it stands for
    ua = dict(args, sparsity_mode='unstructured', sparsity_frac=0.5, first='s')
    lin = torch.nn.Linear(256, 64).to("cuda:0").to(torch.bfloat16)
    pl = PackedBFPLinear.from_linear(lin, ua)
    from quantization_sparsity_interplay_amd.bfp import bfp_ops
    wq = bfp_ops.float_to_bfp_blocked(lin.weight.detach(), **bfp_ops.unpack_bfp_args(dict(ua)), identifier='w')
    assert torch.equal(pl.packed.dequantize().abs(), wq.abs()) and int((wq == 0).sum()) >= 256 * 64 // 2


@pytest.mark.gpu
def test_layerwise_perplexity_harness_runs():
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import layerwise_eval
    model = _tiny_llama(torch.float16)
    args = pkg.BFPConfig.hbfp(8, 32, w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=0.5).to_kwargs()
    patch_linear_layers(model, args, cache_weights=True)
    tokens = torch.randint(0, 1000, (1, 64 * 4), generator=torch.Generator().manual_seed(1))
    ppl = layerwise_eval.layerwise_perplexity(model, tokens, 64, torch.device("cuda:0"), log=lambda *a: None)
    assert 100.0 < ppl < 1e5                                                # random weights: ~ vocabulary size
    dense = _tiny_llama(torch.float16)
    patch_linear_layers(dense, pkg.BFPConfig().to_kwargs())                 # num_format 'fp32': plain F.linear
    ppl_dense = layerwise_eval.layerwise_perplexity(dense, tokens, 64, torch.device("cuda:0"), log=lambda *a: None)
    assert abs(ppl - ppl_dense) / ppl_dense < 0.5                           # quantized + 50 % sparse stays in the same ballpark
