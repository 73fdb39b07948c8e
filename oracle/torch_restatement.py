"""TEST INFRASTRUCTURE -- never imported by the product (quantization-sparsity-interplay_amd/).

Pure-torch CPU restatement of the reference's hot path, ATen op for ATen op, so that timing it on the GPU
box's host cores is timing what the reference itself would spend there (bench.py's `cpu_baseline`; the
reference's own files do not travel to the GPU box).  Written from SURVEY.md Appendix A and the op list of
§8(a); pinned bit for bit against the fixtures the reference itself produced (tests/test_oracle_golden.py,
G2 / G3 / G4), which is what makes it a faithful stand-in.

What follows which reference lines (src/transformers/bfp/bfp_ops.py):
    shared_exponent   :29-33   abs, max(dim=1), + eps, log2, ceil -- all in the tensor's dtype
    snap_blocks       :35-44   pow, pow, sub, div, round (half-even), mul_, neg, max, min
    hbfp              :46-59   pad to the block width, view(-1, block), snap, un-view, narrow
    prune_groups      :73-91   pad to M, view(-1, M), topk(|.|, M-N, smallest), int64 ones mask, scatter_, where
    prune_global      :61-71   the same recipe on the tensor flattened to one row, k = int(numel * frac)
    fake_quantize     :124-149 sparsity flag by identifier, order by `first`
"""
import torch
import torch.nn.functional as F


def shared_exponent(blocks, eps):
    return (blocks.abs().max(dim=1, keepdim=True)[0] + eps).log2().ceil()


def snap_blocks(blocks, mant_bits, eps):
    e = shared_exponent(blocks, eps)
    step = torch.pow(2.0, e - mant_bits)
    top = torch.pow(2.0, e) - step
    q = (blocks / step).round()
    q *= step
    return torch.min(torch.max(q, -top), top)


def _rows_of(t, width):
    """last dim zero-padded to a multiple of `width`, then [-1, width]; also the padded shape"""
    shape = list(t.shape)
    tail = shape[-1] % width
    if tail:
        t = F.pad(t, (0, width - tail), 'constant')
        shape[-1] += width - tail
    return t.contiguous().view(-1, width), shape


def hbfp(t, block_size, mant_bits, eps=1e-8):
    cols = t.shape[-1]
    rows, shape = _rows_of(t, block_size)
    return snap_blocks(rows, mant_bits, eps).contiguous().view(shape).narrow(-1, 0, cols)


def _zero_smallest(rows, drop):
    idx = torch.topk(torch.abs(rows), k=drop, dim=1, largest=False)[1]
    mask = torch.full(rows.shape, 1)                       # int64, as in the reference
    mask.scatter_(index=idx, dim=1, value=0)
    return torch.where(mask == 0, 0, rows)


def prune_groups(t, N, M):
    assert N > 0 and M > 0 and N <= M
    cols = t.shape[-1]
    rows, shape = _rows_of(t, M)
    return _zero_smallest(rows, M - N).contiguous().view(shape).narrow(-1, 0, cols)


def prune_global(t, frac):
    assert frac > 0
    row = t.contiguous().view(1, -1)
    return _zero_smallest(row, int(row.shape[1] * frac)).contiguous().view(t.shape)


def fake_quantize(t, mant_bits, epsilon, rounding_mode, device, block_size, num_format, weight_mant_bits,
                  in_sparsity, w_sparsity, grad_sparsity, sparsity_frac, N, M, sparsity_num_format, first,
                  sparsity_mode, identifier='', sgd_update=False, **_unused):
    """float_to_bfp_blocked for the 'bfp' / 'fp32' formats and round-half-even ('determ')"""
    assert num_format == 'bfp' and rounding_mode == 'determ'
    on = {'in': in_sparsity, 'w': w_sparsity, 'grad': grad_sparsity}.get(identifier, False) == True  # noqa: E712

    def S(x):
        if not on:
            return x
        if sparsity_mode == 'structured':
            return prune_groups(x, N, M)
        if sparsity_mode == 'unstructured':
            return prune_global(x, sparsity_frac)
        raise ValueError(sparsity_mode)

    def Q(x):
        if sparsity_num_format == 'fp32':
            return x
        if sparsity_num_format == 'bfp':
            return hbfp(x, block_size, weight_mant_bits if sgd_update else mant_bits, epsilon)
        raise ValueError(sparsity_num_format)

    return Q(S(t)) if first == 's' else S(Q(t))
