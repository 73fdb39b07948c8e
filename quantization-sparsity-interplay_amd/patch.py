"""Patch an existing torch model in place: every nn.Linear (optionally nn.Conv2d) that the reference's patched
model files build as BFPLinear / BFPConv2d becomes one, sharing the original parameters.

The reference does this by editing modeling_*.py (e.g. llama/modeling_llama.py:225-237,305-319: q/k/v/o and
gate/up/down become BFPLinear, lm_head stays nn.Linear, llama/modeling_llama.py:1166); for an unmodified upstream
model the same result is obtained by swapping the modules after construction."""
import torch

from .bfp import bfp_ops


def patch_linear_layers(model, bfp_args, skip=("lm_head", "classifier", "score"), convs=False, cache_weights=False):
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
                new.enable_weight_cache()
            setattr(parent, child_name, new)
            patched.append(full)
    return patched
