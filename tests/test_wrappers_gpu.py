"""GPU tests of the module / functional wrappers the patched models call (reference bfp_ops.py:233-268) against fixture G11,
which tests/golden/make_golden.py captured from the reference's own BFPConv2d / F_matmul_bfp / F_linear_bfp (forward AND
backward), and of the fp32 shared-exponent windows on the device against fixture G1 (reference bfp_ops.py:29-33).
Quantized operands: bit-exact.  Outputs and gradients pass through an ordinary convolution / matmul on other hardware:
tolerance (the G8 BFPLinear tolerances), stated where used."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from quantization_sparsity_interplay_amd.bfp import bfp_ops
from util import DT, load, from_bits, bits, assert_bits_equal
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cfg(**kw):
    base = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cuda', block_size=64,
                num_format='bfp', weight_mant_bits=15, in_sparsity=False, w_sparsity=False,
                grad_sparsity=False, sparsity_frac=0.5, N=2, M=4, sparsity_num_format='bfp',
                first='s', sparsity_mode='structured')
    base.update(kw)
    return base


KW = dict(mant_bits=7, block_size=16, N=1, M=4, w_sparsity=True, sparsity_mode='structured')      # as the fixture (BASELINE config 5's numerics)


def _tol(dt):
    # the op between the quantizers is a library convolution / GEMM on different hardware (fp32: summation order; bf16: one
    # bf16 rounding of an O(1) result): same tolerances as the G8 BFPLinear test
    return dict(rtol=2e-2, atol=2e-2) if dt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)


def _close(got, g, key, dt, shape):
    torch.testing.assert_close(got.detach().cpu().float(), from_bits(g[key], dt).view(shape).float(), **_tol(dt))


@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_g11_bfpconv2d_forward_backward(dname):
    """BFPConv2d(3, 64, 16, stride=16) in 'bfp' mode -- ViT-L's patch embedding (modeling_vit.py:168-173): a 4-D activation
    and a 4-D weight (blocks along kw) through the module, autograd included"""
    g = load("g11_wrappers.npz")
    dt = DT[dname]
    kw = cfg(**KW)
    conv = bfp_ops.BFPConv2d(3, 64, 16, stride=16, **dict(kw)).to(dt)
    assert conv.num_format == 'bfp' and sorted(conv.state_dict()) == ['bias', 'weight']
    with torch.no_grad():
        conv.weight.copy_(from_bits(g[f"conv_w_{dname}"], dt).view(64, 3, 16, 16))
        conv.bias.copy_(from_bits(g[f"conv_b_{dname}"], dt).view(64))
    conv = conv.to(DEV)
    x = from_bits(g[f"conv_x_{dname}"], dt).view(2, 3, 32, 32).to(DEV).requires_grad_(True)
    gy = from_bits(g[f"conv_gy_{dname}"], dt).view(2, 64, 2, 2).to(DEV)
    # the three quantized operands the module computes: bit-exact
    assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(x.detach(), **kw, identifier='in')), g[f"conv_xq_{dname}"], dt, "conv xq")
    assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(conv.weight.detach(), **kw, identifier='w')), g[f"conv_wq_{dname}"], dt, "conv wq")
    assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(gy, **kw, identifier='grad')), g[f"conv_gq_{dname}"], dt, "conv gq")
    y = conv(x)
    assert y.shape == (2, 64, 2, 2) and y.dtype == dt
    y.backward(gy)
    _close(y, g, f"conv_y_{dname}", dt, (2, 64, 2, 2))
    _close(x.grad, g, f"conv_gx_{dname}", dt, (2, 3, 32, 32))
    _close(conv.weight.grad, g, f"conv_gw_{dname}", dt, (64, 3, 16, 16))
    _close(conv.bias.grad, g, f"conv_gb_{dname}", dt, (64,))
    # the module's forward IS conv2d on the two golden operands (pins that nothing else happens in between); the library may
    # pick another algorithm per call, so: the library's own reproducibility, not bits
    xq = from_bits(g[f"conv_xq_{dname}"], dt).view(2, 3, 32, 32).to(DEV)
    wq = from_bits(g[f"conv_wq_{dname}"], dt).view(64, 3, 16, 16).to(DEV)
    torch.testing.assert_close(y.detach(), F.conv2d(xq, wq, conv.bias.detach(), stride=16), **_tol(dt))
    # inference path (no autograd nodes) gives the same tensor
    with torch.no_grad():
        torch.testing.assert_close(conv(x.detach()), y.detach(), **_tol(dt))


@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_g11_f_matmul_bfp_forward_backward(dname):
    """the callable of F_matmul_bfp: second operand quantized along its transposed last dim (transpose=True) inside an
    autograd graph, straight-through operand gradients, output gradient quantized with identifier 'grad'"""
    g = load("g11_wrappers.npz")
    dt = DT[dname]
    kw = cfg(**KW)
    mm = bfp_ops.F_matmul_bfp(**dict(kw))
    assert mm is not torch.matmul
    a = from_bits(g[f"mm_a_{dname}"], dt).view(2, 4, 16, 32).to(DEV).requires_grad_(True)
    b = from_bits(g[f"mm_b_{dname}"], dt).view(2, 4, 32, 16).to(DEV).requires_grad_(True)
    gy = from_bits(g[f"mm_gy_{dname}"], dt).view(2, 4, 16, 16).to(DEV)
    aq, bq = bfp_ops.MxM_pre_processing(a.detach(), b.detach(), True, **kw)
    assert_bits_equal(bits(aq), g[f"mm_aq_{dname}"], dt, "matmul aq")
    assert_bits_equal(bits(bq.contiguous()), g[f"mm_bq_{dname}"], dt, "matmul bq (transpose path)")
    assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(gy, **kw, identifier='grad')), g[f"mm_gq_{dname}"], dt, "matmul gq")
    y = mm(a, b)
    y.backward(gy)
    _close(y, g, f"mm_y_{dname}", dt, (2, 4, 16, 16))
    _close(a.grad, g, f"mm_ga_{dname}", dt, (2, 4, 16, 32))
    _close(b.grad, g, f"mm_gb_{dname}", dt, (2, 4, 32, 16))
    torch.testing.assert_close(y.detach(), torch.matmul(aq, bq), **_tol(dt))
    # gradients are those of matmul on the quantized operands with the quantized output gradient (straight-through)
    gq = from_bits(g[f"mm_gq_{dname}"], dt).view(2, 4, 16, 16).to(DEV)
    torch.testing.assert_close(a.grad, torch.matmul(gq, bq.transpose(-1, -2)), **_tol(dt))
    torch.testing.assert_close(b.grad, torch.matmul(aq.transpose(-1, -2), gq), **_tol(dt))
    with torch.no_grad():
        torch.testing.assert_close(mm(a.detach(), b.detach()), y.detach(), **_tol(dt))


@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_g11_f_linear_bfp_forward_backward(dname):
    """the callable of F_linear_bfp (what bfp_rnn-style callers use instead of the module)"""
    g = load("g11_wrappers.npz")
    dt = DT[dname]
    kw = cfg(**KW)
    fl = bfp_ops.F_linear_bfp(**dict(kw))
    assert fl is not F.linear
    x = from_bits(g[f"lin_x_{dname}"], dt).view(2, 5, 64).to(DEV).requires_grad_(True)
    w = from_bits(g[f"lin_w_{dname}"], dt).view(48, 64).to(DEV).requires_grad_(True)
    b = from_bits(g[f"lin_b_{dname}"], dt).view(48).to(DEV).requires_grad_(True)
    gy = from_bits(g[f"lin_gy_{dname}"], dt).view(2, 5, 48).to(DEV)
    y = fl(x, w, b)
    y.backward(gy)
    _close(y, g, f"lin_y_{dname}", dt, (2, 5, 48))
    _close(x.grad, g, f"lin_gx_{dname}", dt, (2, 5, 64))
    _close(w.grad, g, f"lin_gw_{dname}", dt, (48, 64))
    _close(b.grad, g, f"lin_gb_{dname}", dt, (48,))
    xq = bfp_ops.float_to_bfp_blocked(x.detach(), **kw, identifier='in')
    wq = bfp_ops.float_to_bfp_blocked(w.detach(), **kw, identifier='w')
    assert_bits_equal(bits(xq), bits(O.float_to_bfp_blocked(x.detach().cpu(), **kw, identifier='in')), dt, "F_linear xq")
    assert_bits_equal(bits(wq), bits(O.float_to_bfp_blocked(w.detach().cpu(), **kw, identifier='w')), dt, "F_linear wq")
    torch.testing.assert_close(y.detach(), F.linear(xq, wq, b.detach()), **_tol(dt))
    # keyword bias and the fp32 format's plain functions (bfp_ops.py:237-238, :244-245)
    with torch.no_grad():
        torch.testing.assert_close(fl(x.detach(), w.detach(), bias=b.detach()), y.detach(), **_tol(dt))
    assert bfp_ops.F_linear_bfp(num_format='fp32') is F.linear and bfp_ops.F_matmul_bfp(num_format='fp32') is torch.matmul


@pytest.mark.parametrize("blk,m", [(16, 3), (64, 7), (32, 15)])
def test_f32_block_max_windows_on_the_device(blk, m):
    """G1's 4 577 fp32 patterns -- every binade's 2^k (1 + j 2^-23), j <= 11, where fl32(k + log2(1 + f)) rounds back to k for
    the first few j, plus mid / top mantissas, zero and subnormals -- as fp32 BLOCK MAXIMA through the device
    quantizer: the shared exponent must equal the reference's own get_exponent (fixture) and the tensor the oracle's bits.
    (OPT and ViT, BASELINE configs 1 and 5, are fp32 models; the host table alone was pinned before.)"""
    g = load("g1_exponent.npz")
    pat = g["f32_bits"].astype(np.uint32)
    e_ref = g["f32_e"].astype(np.float64)
    n = pat.size
    rng = np.random.default_rng(17)
    x = np.zeros((n, blk), dtype=np.uint32)
    x[:, 0] = pat
    for j in range(1, blk):                                  # mates: magnitudes <= the max (random exponent drop, mantissa, sign)
        drop = rng.integers(0, 30, size=n).astype(np.int64) << 23
        mate = np.maximum(pat.astype(np.int64) - drop - rng.integers(0, 1 << 23, size=n), 0)
        x[:, j] = mate.astype(np.uint32) | (rng.integers(0, 2, size=n).astype(np.uint32) << 31)
    x[:, 0] |= rng.integers(0, 2, size=n).astype(np.uint32) << 31
    x[:, [0, blk - 1]] = x[:, [blk - 1, 0]]                   # the max sits in the last lane item of its block
    xc = from_bits(x.reshape(-1), torch.float32).view(n, blk)
    xd = xc.to(DEV)
    got = bfp_ops._no_sparsity_float_to_bfp(xd, blk, m, 1e-8, 'determ', 'cuda')
    assert_bits_equal(bits(got), bits(O.no_sparsity_float_to_bfp(xc, blk, m, 1e-8)), torch.float32, f"f32 windows b{blk} m{m}")
    # the exponent itself, against the reference's get_exponent on these very maxima
    code_bits = 4 if m <= 3 else (8 if m <= 7 else 16)
    _, exps = bfp_ops.float_to_bfp_packed(xd, m, blk, code_bits=code_bits)
    want_e = np.clip(e_ref, -127, 127)
    got_e = exps.cpu().numpy().reshape(-1).astype(np.float64)
    finite = np.isfinite(e_ref)
    assert finite.all()                                       # (fp32: epsilon 1e-8 is representable, no -inf exponents)
    bad = got_e != want_e
    assert not bad.any(), (int(bad.sum()), hex(int(pat[np.flatnonzero(bad)[0]])), got_e[bad][:4], want_e[bad][:4])
    # and through the composed entry point with 2:4 in both orders (the fused fp32 instantiations)
    for first in ('s', 'q'):
        c = cfg(mant_bits=min(m, 7), block_size=blk, w_sparsity=True, first=first)
        assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(xd, **c, identifier='w')),
                          bits(O.float_to_bfp_blocked(xc, **c, identifier='w')), torch.float32, f"f32 windows composed {first}")
