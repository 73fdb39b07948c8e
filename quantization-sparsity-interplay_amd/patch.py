"""Patch an existing torch model in place: every nn.Linear (optionally nn.Conv2d) that the reference's patched
model files build as BFPLinear / BFPConv2d becomes one, sharing the original parameters.

The reference does this by editing modeling_*.py (e.g. llama/modeling_llama.py:225-237,305-319: q/k/v/o and
gate/up/down become BFPLinear, lm_head stays nn.Linear, llama/modeling_llama.py:1166); for an unmodified upstream
model the same result is obtained by swapping the modules after construction."""
import torch

from .bfp import bfp_ops


def patch_linear_layers(model, bfp_args, skip=("lm_head", "classifier", "score"), convs=False, cache_weights=False, matrix_unit=False):
    """Replace nn.Linear children by bfp_ops.BFPLinear(**bfp_args) that reuse weight / bias.  Modules whose
    qualified name ends with an entry of `skip` stay as they are (the reference leaves the LM heads alone).
    Returns the list of patched module names."""
    patched = []
    for parent_name, parent in list(model.named_modules()):
        for child_name, child in list(parent.named_children()):
            full = f"{parent_name}.{child_name}" if parent_name else child_name
            if any(full.endswith(s) for s in skip):
                continue
            new = None
            if type(child) is torch.nn.Linear:
                new = bfp_ops.BFPLinear(child.in_features, child.out_features, child.bias is not None, **dict(bfp_args))
            elif convs and type(child) is torch.nn.Conv2d:
                new = bfp_ops.BFPConv2d(child.in_channels, child.out_channels, child.kernel_size, child.stride, child.padding,
                                        child.dilation, child.groups, child.bias is not None, **dict(bfp_args))
            if new is None:
                continue
            new.weight = child.weight                       # share the parameters: state_dict keys are unchanged
            new.bias = child.bias
            new.train(child.training)
            if cache_weights:
                new.enable_weight_cache(matrix_unit=matrix_unit and isinstance(new, bfp_ops.BFPLinear))
            setattr(parent, child_name, new)
            patched.append(full)
    return patched


def prime_weight_caches(model):
    """Fill the (opt-in, inference-only) weight caches of every BFPLinear / BFPConv2d of `model` with ONE list call per configuration
    (bfp_ops.float_to_bfp_blocked_many: large weights in launches of their own over several streams, small ones up to 64 per launch --
    LLaMA-7B's 224 weights in 4.1 ms) instead of one call per module on its first forward.  Modules whose cache is off or not usable
    right now (training mode, gradients being recorded, stochastic rounding) are left alone.  Returns the number of weights stored."""
    groups = {}
    for m in model.modules():
        cache = getattr(m, '_weight_cache', None)
        if cache is None or getattr(m, 'num_format', None) != 'bfp' or not isinstance(m, (bfp_ops.BFPLinear, bfp_ops.BFPConv2d)):
            continue
        with torch.no_grad():
            if not cache.usable(m.weight, m.bfp_args) or m.weight.device.type != 'cuda':
                continue
        key = (m.weight.dtype, m.weight.device, tuple(sorted((k, repr(v)) for k, v in m.bfp_args.items())))
        groups.setdefault(key, []).append(m)
    n = 0
    for mods in groups.values():
        with torch.no_grad():
            outs = bfp_ops.float_to_bfp_blocked_many([m.weight.detach() for m in mods], identifier='w', **dict(mods[0].bfp_args))
            for m, o in zip(mods, outs):
                m._weight_cache.store(m.weight, o, m.bfp_args)
                n += 1
    return n


class PackedBFPLinear(torch.nn.Module):
    """Inference-only Linear whose weight lives in packed HBFP form (4-bit codes + one int8 exponent per block of 64:
    0.516 B per weight instead of 2) -- what a BFPLinear with w_sparsity / HBFP4 computes in its forward, with the weight
    quantized ONCE.  Up to 64 tokens the product comes straight from the codes (integer block dot products on the int8
    matrix cores, PackedBFP.linear_decode); more tokens decode the weight and use the library GEMM.
    The codes and exponents are registered buffers (`codes`, `exps`): they follow .to() / .cuda(), appear in state_dict()
    and load back with load_state_dict(); shape / dtype / mantissa width travel as extra state."""

    def __init__(self, packed, bias=None, x_mant_bits=7, epsilon=1e-8):
        super().__init__()
        self.out_features, self.in_features = packed.shape
        self.x_mant_bits, self.epsilon = int(x_mant_bits), float(epsilon)
        self.register_buffer("codes", packed.codes)
        self.register_buffer("exps", packed.exps)
        self._meta = dict(shape=list(packed.shape), dtype=str(packed.dtype).replace("torch.", ""), mant_bits=packed.mant_bits,
                          block_size=packed.block_size, code_bits=packed.code_bits)
        self._packed = packed
        self.bias = None if bias is None else torch.nn.Parameter(bias.detach(), requires_grad=False)

    @property
    def packed(self):
        """the PackedBFP view of the buffers (rebuilt when .to() / load_state_dict() replaced them; the MFMA-tiled copy of the
        decode kernel is then re-made lazily on the new device)"""
        p = self._packed
        if p is None or p.codes is not self.codes or p.exps is not self.exps:
            m = self._meta
            p = self._packed = bfp_ops.PackedBFP(self.codes, self.exps, tuple(m['shape']), getattr(torch, m['dtype']), m['mant_bits'],
                                                 m['block_size'], m['code_bits'])
        return p

    def get_extra_state(self):
        return dict(self._meta, x_mant_bits=self.x_mant_bits, epsilon=self.epsilon)

    def set_extra_state(self, state):
        self._meta = {k: state[k] for k in ('shape', 'dtype', 'mant_bits', 'block_size', 'code_bits')}
        self.x_mant_bits, self.epsilon = int(state['x_mant_bits']), float(state['epsilon'])
        self._packed = None

    @classmethod
    def from_linear(cls, lin, bfp_args):
        """weight -> float_to_bfp_packed with the module's config (format 'bfp', block 64, mant_bits <= 3, round-half-even;
        N:M from the w_sparsity keys, order from `first`); activations use mant_bits of the same config.  Unstructured weight pruning is
        packed for first='s' (prune, then quantize: one fused launch writes the codes).  Configurations this module cannot reproduce
        (unstructured pruning after quantization, pruned activations, stochastic rounding) are refused."""
        a = bfp_ops.unpack_bfp_args(dict(bfp_args))
        assert a['num_format'] == 'bfp' and a['sparsity_num_format'] == 'bfp' and a['block_size'] == 64 and 1 <= a['mant_bits'] <= 3, \
            "PackedBFPLinear holds 4-bit codes: an HBFP config with block_size 64 and mant_bits <= 3 (HBFP4)"
        if a['w_sparsity'] and a['sparsity_mode'] == 'unstructured' and a['first'] != 's':
            raise ValueError("PackedBFPLinear: unstructured pruning AFTER quantization (first='q') is not packed here")
        if a['w_sparsity'] and a['sparsity_mode'] not in ('structured', 'unstructured'):
            raise ValueError('Sparsity mode not implemented')
        if a['in_sparsity']:
            raise ValueError("PackedBFPLinear quantizes activations densely: in_sparsity is not reproduced")
        if a['rounding_mode'] != 'determ':
            raise ValueError("PackedBFPLinear packs with round-half-even: rounding_mode must be 'determ'")
        sp = a['w_sparsity'] and a['sparsity_mode'] == 'structured'
        if a['w_sparsity'] and a['sparsity_mode'] == 'unstructured':
            pw = bfp_ops.PackedBFP.quantize_unstructured(lin.weight.detach(), a['mant_bits'], 64, a['sparsity_frac'], a['epsilon'])
            return cls(pw, lin.bias, a['mant_bits'], a['epsilon'])
        pw = bfp_ops.PackedBFP.quantize(lin.weight.detach(), a['mant_bits'], 64, a['epsilon'], a['N'] if sp else 0, a['M'] if sp else 0, a['first'])
        return cls(pw, lin.bias, a['mant_bits'], a['epsilon'])

    def forward(self, x):
        return self.packed.linear(x, self.bias, self.x_mant_bits, self.epsilon)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, packed_bytes={self.packed.nbytes()}"


def pack_linear_layers(model, bfp_args, skip=("lm_head", "classifier", "score")):
    """Inference deployment: replace nn.Linear / BFPLinear children by PackedBFPLinear (weights quantized once and
    held as 4-bit codes; the original bf16/fp16 weight is dropped).  Needs an HBFP config with block_size 64 and
    weight mantissa <= 3 bits (+ optional N:M).  Returns the patched names and the bytes saved."""
    patched, saved = [], 0
    for parent_name, parent in list(model.named_modules()):
        for child_name, child in list(parent.named_children()):
            full = f"{parent_name}.{child_name}" if parent_name else child_name
            if any(full.endswith(s) for s in skip) or not isinstance(child, torch.nn.Linear):
                continue
            if child.in_features % 256 or child.out_features % 16:
                continue                                    # shapes the decode kernel does not take stay as they are
            new = PackedBFPLinear.from_linear(child, bfp_args)
            saved += child.weight.numel() * child.weight.element_size() - new.packed.nbytes()
            setattr(parent, child_name, new)
            patched.append(full)
    return patched, saved
