"""torch.library registration of the engine's two tensor->tensor entry points, so that code calling them is
traceable (torch.compile / export see an opaque op with a shape-and-dtype rule instead of a ctypes call):

    torch.ops.bfpq.fake_quantize(t, block_size, mant_bits, epsilon, N, M, sparsify_first, seed)
        = bfpq_quantize_nm in drop-in mode (the reference's float_to_bfp_blocked for structured / no sparsity)
    torch.ops.bfpq.prune_threshold_quantize(t, k, block_size, mant_bits, epsilon, seed)
        = radix select of the k-th magnitude + bfpq_quantize_threshold (unstructured, first == 's')
"""
import torch

from . import native

def _workspace(device):
    from .bfp import bfp_ops
    return bfp_ops._workspace(device)


@torch.library.custom_op("bfpq::fake_quantize", mutates_args=())
def fake_quantize(t: torch.Tensor, block_size: int, mant_bits: int, epsilon: float, N: int, M: int,
                  sparsify_first: bool, seed: int) -> torch.Tensor:
    y, _, _ = native.quantize_nm(t, block_size, mant_bits, epsilon, N=N, M=M, sparsify_first=sparsify_first, stoch_seed=seed)
    return y.view(t.shape)


@fake_quantize.register_fake
def _(t, block_size, mant_bits, epsilon, N, M, sparsify_first, seed):
    return torch.empty_like(t, memory_format=torch.contiguous_format)


@torch.library.custom_op("bfpq::prune_threshold_quantize", mutates_args=())
def prune_threshold_quantize(t: torch.Tensor, k: int, block_size: int, mant_bits: int, epsilon: float, seed: int) -> torch.Tensor:
    ws = _workspace(t.device)
    native.select_threshold(t, k, ws)
    y, _, _ = native.quantize_threshold(t, ws, block_size, mant_bits, epsilon, stoch_seed=seed)
    return y.view(t.shape)


@prune_threshold_quantize.register_fake
def _(t, k, block_size, mant_bits, epsilon, seed):
    return torch.empty_like(t, memory_format=torch.contiguous_format)
