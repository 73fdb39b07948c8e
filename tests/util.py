"""Shared helpers for the tests: fixture loading and bit-pattern conversions."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def bits(t):
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.float32:
        return t.view(torch.int32).numpy().view(np.uint32).copy()
    return t.view(torch.int16).numpy().view(np.uint16).copy()


def from_bits(a, dtype):
    a = np.asarray(a)
    if dtype == torch.float32:
        return torch.from_numpy(a.astype(np.uint32).view(np.int32).copy()).view(torch.float32)
    return torch.from_numpy(a.astype(np.uint16).view(np.int16).copy()).view(dtype)


def is_nan_bits(a, dtype):
    a = np.asarray(a)
    if dtype == torch.float32:
        return (a & 0x7FFFFFFF) > 0x7F800000
    if dtype == torch.float16:
        return (a & 0x7FFF) > 0x7C00
    return (a & 0x7FFF) > 0x7F80


def assert_bits_equal(got, want, dtype, what=""):
    """bit-exact, except that any NaN matches any NaN (payload/sign of NaN is not part of the contract)"""
    got = np.asarray(got).reshape(-1)
    want = np.asarray(want).reshape(-1)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    gn, wn = is_nan_bits(got, dtype), is_nan_bits(want, dtype)
    bad = (gn != wn) | (~gn & (got != want))
    if bad.any():
        i = int(np.flatnonzero(bad)[0])
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.size} mismatches; first at {i}: got 0x{int(got[i]):x} want 0x{int(want[i]):x}")
