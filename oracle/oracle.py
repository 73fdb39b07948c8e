"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end for oracle/libbfp_oracle.so, the plain-C CPU restatement of the reference's
hot path (reference: src/transformers/bfp/bfp_ops.py:16-149).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Parity pin: checked against outputs of the reference itself (tests/golden/*.npz, produced by
tests/golden/make_golden.py in the build container) by tests/test_oracle_golden.py.

The functions take and return CPU torch tensors and mirror the reference's names and argument
meaning (rounding_mode is always 'determ': the reference's 'stoc' path draws from torch's CPU
RNG stream, which nothing else can reproduce).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def build(force=False):
    so = os.path.join(_HERE, "libbfp_oracle.so")
    src = os.path.join(_HERE, "bfp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libbfp_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        i64, i32, dbl, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p
        L.oracle_bfp_quantize.argtypes = [vp, vp, i64, i64, i32, i32, i32, dbl, vp]
        L.oracle_nm_sparsify.argtypes = [vp, vp, i64, i64, i32, i32, i32]
        L.oracle_unstructured_sparsify.argtypes = [vp, vp, i64, i32, dbl, vp, vp]
        L.oracle_float_to_bfp_blocked.argtypes = [vp, vp, i64, i64, i32, i32, i32, i32, dbl, i32, i32, i32, dbl, i32]
        L.oracle_topk_smallest_mask.argtypes = [vp, i64, i64, vp]
        L.oracle_int_quantize.argtypes = [vp, vp, i64, i64, i64, i32, i32]
        L.oracle_int_quantize.restype = ctypes.c_int
        for f in (L.oracle_bfp_quantize, L.oracle_nm_sparsify, L.oracle_unstructured_sparsify,
                  L.oracle_float_to_bfp_blocked, L.oracle_topk_smallest_mask, L.oracle_version):
            f.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _as2d(t):
    assert t.device.type == "cpu" and t.dtype in _DT, (t.device, t.dtype)
    t = t.contiguous()
    cols = t.shape[-1] if t.dim() > 0 else 1
    rows = t.numel() // cols if cols else 0
    return t, rows, cols


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


def no_sparsity_float_to_bfp(t, block_size, mant_bits, epsilon=1e-8, return_exponents=False):
    """reference: _no_sparsity_float_to_bfp, bfp_ops.py:46-59 ('determ')"""
    t, rows, cols = _as2d(t)
    out = torch.empty_like(t)
    nblk = (cols + block_size - 1) // block_size
    exps = torch.empty(rows * nblk, dtype=torch.float32)
    _check(lib().oracle_bfp_quantize(t.data_ptr(), out.data_ptr(), rows, cols, _DT[t.dtype],
                                     block_size, mant_bits, float(epsilon), exps.data_ptr()), "bfp_quantize")
    return (out, exps.view(rows, nblk)) if return_exponents else out


def structured_N_M_sparsity(t, N, M):
    """reference: _structured_N_M_sparsity, bfp_ops.py:73-91"""
    assert N > 0 and M > 0 and N <= M
    t, rows, cols = _as2d(t)
    out = torch.empty_like(t)
    _check(lib().oracle_nm_sparsify(t.data_ptr(), out.data_ptr(), rows, cols, _DT[t.dtype], N, M), "nm_sparsify")
    return out


def unstructured_sparsity(t, sparsity_frac, return_stats=False):
    """reference: _unstructured_sparsity, bfp_ops.py:61-71"""
    assert sparsity_frac > 0
    t, _, _ = _as2d(t)
    out = torch.empty_like(t)
    tau = ctypes.c_float(0)
    k = ctypes.c_int64(0)
    _check(lib().oracle_unstructured_sparsify(t.data_ptr(), out.data_ptr(), t.numel(), _DT[t.dtype],
                                              float(sparsity_frac), ctypes.byref(tau), ctypes.byref(k)), "unstructured")
    return (out, tau.value, k.value) if return_stats else out


def topk_smallest_mask(absvals, k):
    """which k of the n (non-negative fp32) values ATen topk(largest=False) selects"""
    a = np.ascontiguousarray(absvals, dtype=np.float32)
    m = np.zeros(a.shape[0], dtype=np.uint8)
    _check(lib().oracle_topk_smallest_mask(a.ctypes.data, a.shape[0], k, m.ctypes.data), "topk")
    return m


def int_channel_view(shape, weight):
    """(outer, C, inner) of int_ops.Quantizer.find_params' per-channel view (int_ops.py:38-50)"""
    shape = tuple(shape)
    n = 1
    for d in shape:
        n *= d
    if weight:
        return 1, shape[0], n // max(shape[0], 1)
    if len(shape) == 4:
        return shape[0], shape[1], shape[2] * shape[3]
    if len(shape) in (2, 3):
        return n // max(shape[-1], 1), shape[-1], 1
    raise ValueError("int_ops.Quantizer handles 2-D, 3-D and 4-D activations only")


def int_quantize(t, bits, weight):
    """reference: _quantize 'int' branch, bfp_ops.py:111-120 (returns fp32 like the reference)"""
    t = t.contiguous()
    assert t.device.type == "cpu" and t.dtype in _DT
    outer, C, inner = int_channel_view(t.shape, weight)
    out = torch.empty(t.shape, dtype=torch.float32)
    _check(lib().oracle_int_quantize(t.data_ptr(), out.data_ptr(), outer, C, inner, _DT[t.dtype], int(bits)), "int_quantize")
    return out


def float_to_bfp_blocked(t, mant_bits, epsilon, rounding_mode, device, block_size, num_format,
                         weight_mant_bits, in_sparsity, w_sparsity, grad_sparsity, sparsity_frac,
                         N, M, sparsity_num_format, first, sparsity_mode, identifier='',
                         sgd_update=False, mx_w_elem_format='', mx_a_elem_format='', scale_bits=0, bfloat=0):
    """reference: float_to_bfp_blocked, bfp_ops.py:124-149 (same signature; 'determ' only)"""
    assert num_format == 'bfp'
    assert ((sparsity_num_format == 'bfp' and block_size > 0) or sparsity_num_format == 'fp32'
            or sparsity_num_format == 'int')
    if rounding_mode != 'determ':
        raise NotImplementedError("oracle restates rounding_mode='determ' only")
    sparsity = ((in_sparsity is True and identifier == 'in') or (w_sparsity is True and identifier == 'w')
                or (grad_sparsity is True and identifier == 'grad'))
    if sparsity:
        if sparsity_mode == 'structured':
            smode = 1
        elif sparsity_mode == 'unstructured':
            smode = 2
        else:
            raise ValueError(f'Unknown sparsity mode: {sparsity_mode} given as argument')
    else:
        smode = 0
    if sparsity_num_format == 'int':
        # composition done here in Python: the quantizer changes the dtype to fp32 (reference quirk)
        bits = weight_mant_bits if sgd_update else mant_bits

        def S(x):
            if not sparsity:
                return x
            return structured_N_M_sparsity(x, N, M).view(x.shape) if smode == 1 else unstructured_sparsity(x, sparsity_frac).view(x.shape)
        if first == 's':
            return int_quantize(S(t), bits, identifier == 'w')
        return S(int_quantize(t, bits, identifier == 'w'))
    if sparsity_num_format == 'fp32':
        q = 0
    elif sparsity_num_format == 'bfp':
        q = 1
    else:
        raise ValueError(f'oracle does not restate format {sparsity_num_format}')
    mb = weight_mant_bits if sgd_update else mant_bits
    t, rows, cols = _as2d(t)
    out = torch.empty_like(t)
    _check(lib().oracle_float_to_bfp_blocked(t.data_ptr(), out.data_ptr(), rows, cols, _DT[t.dtype], q,
                                             max(int(block_size), 1), int(mb), float(epsilon), smode, int(N), int(M),
                                             float(sparsity_frac), 1 if first == 's' else 0), "float_to_bfp_blocked")
    return out
