"""GPU parity tests (run with `-m gpu` on an MI355X).  Every call goes Python boundary -> C ABI
(libbfpq.so) -> HIP kernels.  Checks, bit-exact unless stated:
  * against the committed golden vectors (outputs of the reference itself, G2..G8)
  * against the CPU oracle on seeded inputs at the BASELINE.json shapes
  * size-independent properties at full size (idempotence, N:M group counts, packed round trip)
"""
import numpy as np
import pytest
import torch

import quantization_sparsity_interplay_amd as pkg
from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops
from util import DT, load, from_bits, bits, assert_bits_equal
from gen import int_bits_tensor
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


def synth(rows, cols, dtype, scale=0.02, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(rows, cols, generator=g) * scale).to(dtype)


def test_native_library_is_the_one_running():
    assert torch.cuda.is_available()
    assert pkg.load_library().bfpq_version() == 4
    import os
    maps = open(f"/proc/{os.getpid()}/maps").read()
    assert "libbfpq.so" in maps


# ---- golden vectors from the reference ---------------------------------------------------------
def test_g2_quantize_golden():
    g = load("g2_quantize.npz")
    for sname in ("s0.02", "s1", "s30"):
        for dname, dt in DT.items():
            x = from_bits(g[f"in_{sname}_{dname}"], dt).view(16, 192).to(DEV)
            for blk in (16, 32, 64):
                for m in (3, 5, 7, 15):
                    y = bfp_ops._no_sparsity_float_to_bfp(x, blk, m, 1e-8, 'determ', 'cuda')
                    assert y.dtype == dt and y.shape == x.shape
                    assert_bits_equal(bits(y), g[f"out_{sname}_{dname}_b{blk}_m{m}"], dt, f"{sname} {dname} b{blk} m{m}")


def test_g3_nm_golden():
    g = load("g3_nm.npz")
    rows = torch.from_numpy(g["m4_rows"].astype(np.float32) + 1.0).to(DEV)
    for N in (1, 2, 3):
        y = bfp_ops._structured_N_M_sparsity(rows, 'cuda', N, 4)
        assert np.array_equal((y != 0).cpu().numpy().astype(np.uint8), g[f"m4_keep_N{N}"]), N
        for dt in (torch.bfloat16, torch.float16):                     # same table through the 16-bit kernels
            y = bfp_ops._structured_N_M_sparsity(rows.to(dt), 'cuda', N, 4)
            assert np.array_equal((y != 0).cpu().numpy().astype(np.uint8), g[f"m4_keep_N{N}"]), (N, dt)
    for (N, M) in ((2, 8), (4, 8), (1, 8), (7, 8), (4, 16), (8, 16), (2, 16), (16, 32), (8, 32), (1, 2), (3, 6), (2, 5)):
        r = g[f"rows_{N}_{M}"].astype(np.float32) + 1.0
        sign = np.where(g[f"sign_{N}_{M}"] != 0, -1.0, 1.0).astype(np.float32)
        y = bfp_ops._structured_N_M_sparsity(torch.from_numpy(r * sign).to(DEV), 'cuda', N, M)
        keep = np.packbits((y != 0).cpu().numpy().astype(np.uint8), axis=1)
        assert np.array_equal(keep, g[f"keep_{N}_{M}"]), (N, M)
    for M in (16, 32, 64):
        r = g[f"killer_rows_{M}"].astype(np.float32) + 1.0
        y = bfp_ops._structured_N_M_sparsity(torch.from_numpy(r).to(DEV), 'cuda', M // 2, M)
        assert np.array_equal(np.packbits((y != 0).cpu().numpy().astype(np.uint8), axis=1), g[f"killer_keep_{M}"]), M
    for dname, dt in DT.items():
        x = from_bits(g[f"real_in_{dname}"], dt).view(512, 16).to(DEV)
        for (N, M) in ((2, 4), (1, 4), (3, 4), (4, 8), (2, 16)):
            assert_bits_equal(bits(bfp_ops._structured_N_M_sparsity(x, 'cuda', N, M)), g[f"real_out_{dname}_{N}_{M}"], dt, f"{dname} {N}:{M}")


def _tie_class_check(x_cpu, y_gpu, frac, dt, what):
    """unstructured contract: same threshold and count as the oracle, identical outside the tie class"""
    want, tau, k = O.unstructured_sparsity(x_cpu, frac, return_stats=True)
    got = y_gpu.cpu()
    mag = x_cpu.float().abs()
    tau_t = torch.tensor(tau)
    outside = ~((mag == tau_t) | (torch.isnan(mag) & torch.isnan(tau_t)))      # all NaNs form one tie class (ATen's comparator)
    sel = outside.view(-1).numpy()
    assert_bits_equal(bits(got).reshape(-1)[sel], bits(want).reshape(-1)[sel], dt, what + " (outside tie class)")
    tie = ~outside
    pruned_w = int(((want == 0) & tie).sum())
    pruned_g = int(((got == 0) & tie).sum())
    if tau != 0 and tau == tau:
        assert pruned_g == pruned_w, (what, pruned_g, pruned_w)
    kept = (tie.view(-1) & (got.view(-1) != 0)).numpy()                     # kept ties are bit-unchanged (NaN == NaN here)
    assert_bits_equal(bits(got).reshape(-1)[kept], bits(x_cpu).reshape(-1)[kept], dt, what + " (kept ties)")
    # among ties the engine prunes the lowest flat indices
    idx = torch.nonzero(tie.view(-1)).view(-1)
    if tau != 0 and idx.numel():
        z = (got.view(-1)[idx] == 0)
        n = int(z.sum())
        assert bool(z[:n].all()) and not bool(z[n:].any()), what


def test_g4_composed_golden():
    g = load("g4_composed.npz")
    for dname, dt in DT.items():
        xc = from_bits(g[f"in_{dname}"], dt).view(32, 256)
        x = xc.to(DEV)
        for first in ('s', 'q'):
            for mode, extra in (("structured", dict(N=2, M=4)), ("structured", dict(N=1, M=4)), ("structured", dict(N=4, M=8))):
                for m, blk in ((3, 64), (7, 32), (7, 16)):
                    c = cfg(mant_bits=m, block_size=blk, first=first, sparsity_mode=mode, w_sparsity=True, **extra)
                    y = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
                    tag = f"{dname}_{first}_{mode[:1]}_{extra.get('N', 0)}_{extra.get('M', 0)}_{extra.get('sparsity_frac', 0)}_m{m}_b{blk}"
                    assert_bits_equal(bits(y), g[f"out_{tag}"], dt, tag)
        for ident in ('w', 'in', 'grad', ''):
            for flag in ('in_sparsity', 'w_sparsity', 'grad_sparsity'):
                y = bfp_ops.float_to_bfp_blocked(x, **cfg(**{flag: True}), identifier=ident)
                assert_bits_equal(bits(y), g[f"ident_{dname}_{ident or 'none'}_{flag}"], dt, f"ident {ident} {flag}")
        y = bfp_ops.float_to_bfp_blocked(x, **cfg(sparsity_num_format='fp32', w_sparsity=True), identifier='w')
        assert_bits_equal(bits(y), g[f"fp32fmt_{dname}"], dt, "fp32fmt")
        y = bfp_ops.float_to_bfp_blocked(x, **cfg(weight_mant_bits=15), identifier='', sgd_update=True)
        assert_bits_equal(bits(y), g[f"sgd_{dname}"], dt, "sgd")
        # unstructured in both orders: the tie-class contract, checked step by step against the oracle
        for frac in (0.5, 0.3):
            for m, blk in ((3, 64), (7, 16)):
                ys = bfp_ops._unstructured_sparsity(x, 'cuda', frac)
                _tie_class_check(xc, ys, frac, dt, f"unstructured s {dname} {frac}")
                xq = O.no_sparsity_float_to_bfp(xc, blk, m)
                yq = bfp_ops._unstructured_sparsity(xq.to(DEV), 'cuda', frac)
                _tie_class_check(xq, yq, frac, dt, f"unstructured q {dname} {frac} m{m}")
                # full composition through the public entry point runs (values checked piecewise above)
                c = cfg(mant_bits=m, block_size=blk, first='q', sparsity_mode='unstructured', w_sparsity=True, sparsity_frac=frac)
                yc = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
                assert torch.equal(yc, yq)


def test_g5_unstructured_golden():
    g = load("g5_unstructured.npz")
    xc = from_bits(g["small_in"], torch.bfloat16).view(64, 256)
    for frac in (0.5, 0.25, 0.9, 0.001):
        y = bfp_ops._unstructured_sparsity(xc.to(DEV), 'cuda', frac)
        want = from_bits(g[f"small_out_{frac}"], torch.bfloat16).view(64, 256)
        assert int((y == 0).sum()) == int((want == 0).sum())
        _tie_class_check(xc, y, frac, torch.bfloat16, f"g5 {frac}")
    for dname, shape, seed in (("bf16", (512, 1024), 11), ("f16", (256, 512), 12), ("f32", (256, 512), 13)):
        xc = from_bits(int_bits_tensor(shape, dname, seed), DT[dname]).view(shape)
        y = bfp_ops._unstructured_sparsity(xc.to(DEV), 'cuda', 0.5)
        ref_zero = np.unpackbits(g[f"big_zero_{dname}"])[: xc.numel()]
        assert int((y == 0).sum()) == int(ref_zero.sum()), dname            # same count as the reference
        _tie_class_check(xc, y, 0.5, DT[dname], f"g5 big {dname}")


def test_g6_padding_golden():
    g = load("g6_padding.npz")
    for C in (100, 6, 65, 1, 63, 129):
        for dname, dt in DT.items():
            x = from_bits(g[f"in_{C}_{dname}"], dt).view(5, C).to(DEV)
            assert_bits_equal(bits(bfp_ops._no_sparsity_float_to_bfp(x, 64, 3, 1e-8, 'determ', 'cuda')), g[f"q_{C}_{dname}"], dt, f"q {C} {dname}")
            assert_bits_equal(bits(bfp_ops._structured_N_M_sparsity(x, 'cuda', 2, 4)), g[f"nm_{C}_{dname}"], dt, f"nm {C} {dname}")
            for first in ('s', 'q'):
                y = bfp_ops.float_to_bfp_blocked(x, **cfg(first=first, w_sparsity=True), identifier='w')
                assert_bits_equal(bits(y), g[f"comp_{first}_{C}_{dname}"], dt, f"comp {first} {C} {dname}")
                y = bfp_ops.float_to_bfp_blocked(x, **cfg(first=first, w_sparsity=True, N=3, M=8, block_size=16, mant_bits=7), identifier='w')
                assert_bits_equal(bits(y), g[f"comp38_{first}_{C}_{dname}"], dt, f"comp38 {first} {C} {dname}")


def test_g7_edges_golden():
    g = load("g7_edges.npz")
    for dname, dt in DT.items():
        x = from_bits(g[f"in_{dname}"], dt).view(-1, 16).to(DEV)
        for m in (3, 7, 15):
            y = bfp_ops._no_sparsity_float_to_bfp(x, 16, m, 1e-8, 'determ', 'cuda')
            assert_bits_equal(bits(y), g[f"out_{dname}_m{m}"], dt, f"edges {dname} m{m}")
        assert_bits_equal(bits(bfp_ops._structured_N_M_sparsity(x, 'cuda', 2, 4)), g[f"nm_{dname}"], dt, f"edges nm {dname}")


def test_g8_nd_and_linear_golden():
    g = load("g8_nd_linear.npz")
    c = cfg(mant_bits=7, block_size=16, N=1, M=4, in_sparsity=True, w_sparsity=True)
    for dname, dt in DT.items():
        a = from_bits(g[f"act_in_{dname}"], dt).view(2, 7, 128).to(DEV)
        y = bfp_ops.float_to_bfp_blocked(a, **c, identifier='in')
        assert y.shape == a.shape
        assert_bits_equal(bits(y), g[f"act_out_{dname}"], dt, "act")
        w = from_bits(g[f"conv_in_{dname}"], dt).view(8, 3, 16, 16).to(DEV)
        assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(w, **c, identifier='w')), g[f"conv_out_{dname}"], dt, "conv")
        a4 = from_bits(g[f"mm_a_in_{dname}"], dt).view(2, 4, 16, 32).to(DEV)
        b4 = from_bits(g[f"mm_b_in_{dname}"], dt).view(2, 4, 32, 16).to(DEV)
        xa, xb = bfp_ops.MxM_pre_processing(a4, b4, True, **c)
        assert_bits_equal(bits(xa), g[f"mm_a_out_{dname}"], dt, "matmul a")
        assert xb.shape == b4.shape
        assert_bits_equal(bits(xb.contiguous()), g[f"mm_b_out_{dname}"], dt, "matmul b (transpose path)")
    for dname in ("f32", "bf16"):
        dt = DT[dname]
        kw = cfg(mant_bits=7, block_size=32, N=2, M=4, w_sparsity=True, sparsity_mode='structured')
        lin = bfp_ops.BFPLinear(64, 128, True, **dict(kw)).to(dt)
        with torch.no_grad():
            lin.weight.copy_(from_bits(g[f"lin_w_{dname}"], dt).view(128, 64))
            lin.bias.copy_(from_bits(g[f"lin_b_{dname}"], dt).view(128))
        lin = lin.to(DEV)
        x = from_bits(g[f"lin_x_{dname}"], dt).view(2, 5, 64).to(DEV).requires_grad_(True)
        gy = from_bits(g[f"lin_gy_{dname}"], dt).view(2, 5, 128).to(DEV)
        # the three quantized operands: bit-exact
        assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(x.detach(), **kw, identifier='in')), g[f"lin_xq_{dname}"], dt, "lin xq")
        assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(lin.weight.detach(), **kw, identifier='w')), g[f"lin_wq_{dname}"], dt, "lin wq")
        assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(gy, **kw, identifier='grad')), g[f"lin_gq_{dname}"], dt, "lin gq")
        # the GEMM itself is an ordinary fp matmul on different hardware: tolerance, not bits
        y = lin(x)
        y.backward(gy)
        tol = dict(rtol=2e-2, atol=2e-2) if dt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(y.detach().cpu().float(), from_bits(g[f"lin_y_{dname}"], dt).view(2, 5, 128).float(), **tol)
        torch.testing.assert_close(x.grad.cpu().float(), from_bits(g[f"lin_gx_{dname}"], dt).view(2, 5, 64).float(), **tol)
        torch.testing.assert_close(lin.weight.grad.cpu().float(), from_bits(g[f"lin_gw_{dname}"], dt).view(128, 64).float(), **tol)
        torch.testing.assert_close(lin.bias.grad.cpu().float(), from_bits(g[f"lin_gb_{dname}"], dt).view(128).float(), **tol)


# ---- oracle at the BASELINE.json shapes --------------------------------------------------------
@pytest.mark.parametrize("case", [
    ("cfg1 OPT-125m [768,768] f32 HBFP8 b32 dense", 768, 768, "f32", dict(mant_bits=7, block_size=32)),
    ("cfg2 q_proj [4096,4096] bf16 HBFP4 b64 dense", 4096, 4096, "bf16", dict(mant_bits=3, block_size=64)),
    ("cfg3 down_proj [4096,11008] bf16 HBFP4 b64 2:4 s", 4096, 11008, "bf16", dict(w_sparsity=True)),
    ("cfg3 gate_proj [11008,4096] bf16 HBFP4 b64 2:4 s", 11008, 4096, "bf16", dict(w_sparsity=True)),
    ("cfg3 down_proj [4096,11008] f16 HBFP4 b64 2:4 q", 4096, 11008, "f16", dict(w_sparsity=True, first='q')),
    ("cfg3 [4096,4096] bf16 HBFP4 b64 2:4 q (tie heavy)", 4096, 4096, "bf16", dict(w_sparsity=True, first='q')),
    ("cfg5 ViT fc1 [4096,1024] f32 HBFP8 b16 1:4", 4096, 1024, "f32", dict(mant_bits=7, block_size=16, N=1, M=4, w_sparsity=True)),
    ("cfg5 ViT act [8*197,1024] f32 HBFP8 b16 dense in", 8 * 197, 1024, "f32", dict(mant_bits=7, block_size=16, N=1, M=4, w_sparsity=True, ident='in', scale=1.0)),
    ("general path [1000,1000] bf16 b48 3:6 s", 1000, 1000, "bf16", dict(block_size=48, N=3, M=6, w_sparsity=True)),
    ("general path [333,777] f16 b64 4:8 q", 333, 777, "f16", dict(mant_bits=7, N=4, M=8, w_sparsity=True, first='q')),
])
def test_oracle_parity_at_baseline_shapes(case):
    name, rows, cols, dname, kw = case
    kw = dict(kw)
    ident = kw.pop('ident', 'w')
    scale = kw.pop('scale', 0.02)
    dt = DT[dname]
    xc = synth(rows, cols, dt, scale)
    c = cfg(**kw)
    got = bfp_ops.float_to_bfp_blocked(xc.to(DEV), **c, identifier=ident)
    want = O.float_to_bfp_blocked(xc, **c, identifier=ident)
    assert_bits_equal(bits(got), bits(want), dt, name)


def test_oracle_parity_cfg4_unstructured_13b():
    """cfg 4: [5120,5120] bf16, HBFP4 + 50 % unstructured, both orders (tie-class contract)"""
    xc = synth(5120, 5120, torch.bfloat16)
    x = xc.to(DEV)
    ys = bfp_ops._unstructured_sparsity(x, 'cuda', 0.5)
    _tie_class_check(xc, ys, 0.5, torch.bfloat16, "cfg4 s")
    assert int((ys == 0).sum()) == 5120 * 5120 // 2                # exactly k zeroed (no zeros in the input)
    c = cfg(w_sparsity=True, sparsity_mode='unstructured', first='s')
    yq = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
    want = O.no_sparsity_float_to_bfp(ys.cpu(), 64, 3)            # quantize step on the engine's own pruned tensor
    assert_bits_equal(bits(yq), bits(want), torch.bfloat16, "cfg4 s then q")


@pytest.mark.parametrize("shape,dname", [((512, 1024), "bf16"), ((333, 777), "f16"), ((257, 129), "f32"), ((64, 4096), "f32"), ((1, 5), "bf16")])
def test_unstructured_fused_and_fallback(shape, dname):
    """sparsify->quantize with a global threshold: the single-pass kernel on flat shapes, the two-launch
    fallback on ragged ones (numel not a multiple of the vector width, cols not a multiple of the block)"""
    dt = DT[dname]
    xc = synth(shape[0], shape[1], dt, seed=77)
    x = xc.to(DEV)
    for frac in (0.5, 0.37, 0.99):
        ys = bfp_ops._unstructured_sparsity(x, 'cuda', frac)
        _tie_class_check(xc, ys, frac, dt, f"unstructured {shape} {dname} {frac}")
        for m, blk in ((3, 64), (7, 16)):
            c = cfg(mant_bits=m, block_size=blk, w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=frac, first='s')
            got = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
            want = O.no_sparsity_float_to_bfp(ys.cpu(), blk, m)
            assert_bits_equal(bits(got), bits(want), dt, f"fused unstructured {shape} {dname} {frac} m{m} b{blk}")
    # tie-heavy integer-generated input (thousands of elements equal to the threshold)
    xb = from_bits(int_bits_tensor(shape, dname, 5), dt).view(shape)
    ys = bfp_ops._unstructured_sparsity(xb.to(DEV), 'cuda', 0.5)
    _tie_class_check(xb, ys, 0.5, dt, f"tie-heavy {shape} {dname}")
    c = cfg(w_sparsity=True, sparsity_mode='unstructured', first='s')
    got = bfp_ops.float_to_bfp_blocked(xb.to(DEV), **c, identifier='w')
    assert_bits_equal(bits(got), bits(O.no_sparsity_float_to_bfp(ys.cpu(), 64, 3)), dt, f"tie-heavy fused {shape} {dname}")


# ---- packed output -----------------------------------------------------------------------------
@pytest.mark.parametrize("dname,m,blk,code_bits", [("bf16", 3, 64, 4), ("f16", 3, 32, 4), ("f32", 7, 16, 8), ("bf16", 7, 64, 8), ("f32", 15, 32, 16),
                                                   ("f16", 3, 64, 4), ("bf16", 2, 64, 4), ("f16", 1, 64, 4), ("bf16", 1, 64, 4)])
def test_packed_roundtrip(dname, m, blk, code_bits):
    dt = DT[dname]
    xc = synth(512, 1024, dt)
    x = xc.to(DEV)
    for first in ('s', 'q'):
        codes, exps, deq = bfp_ops.float_to_bfp_packed(x, m, blk, N=2, M=4, first=first, code_bits=code_bits, with_dequant=True)
        want = O.float_to_bfp_blocked(xc, **cfg(mant_bits=m, block_size=blk, w_sparsity=True, first=first), identifier='w')
        assert_bits_equal(bits(deq), bits(want), dt, "deq")
        c = codes.cpu()
        if code_bits == 4:
            lo = (c & 0xF).to(torch.int16)
            hi = (c >> 4).to(torch.int16)
            q = torch.stack([lo, hi], dim=-1).view(512, 1024)
            q = torch.where(q > 7, q - 16, q)
        else:
            q = c.to(torch.int32)
        e = exps.cpu().to(torch.float64).repeat_interleave(blk, dim=1)
        val = q.to(torch.float64) * torch.pow(torch.tensor(2.0, dtype=torch.float64), e - m)
        assert torch.equal(val, want.to(torch.float64)), (dname, first)
        assert int((q.abs() > (1 << m) - 1).sum()) == 0
        codes2, exps2 = bfp_ops.float_to_bfp_packed(x, m, blk, N=2, M=4, first=first, code_bits=code_bits)
        assert torch.equal(codes2, codes) and torch.equal(exps2, exps)


@pytest.mark.parametrize("T", [1, 16, 40, 200])
def test_packed_linear_any_token_count(T):
    """PackedBFP.linear: decode kernel in chunks of 16 up to 64 tokens, decode-once + library GEMM beyond; both against
    the fp64 product of the fake-quantised operands"""
    w = synth(256, 512, torch.bfloat16).to(DEV)
    x = synth(T, 512, torch.bfloat16, 1.0, seed=4).to(DEV).view(1, T, 512)
    b = synth(1, 256, torch.bfloat16, 1.0, seed=6).to(DEV).view(256)
    pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
    got = pw.linear(x, b)
    assert got.shape == (1, T, 256) and got.dtype == torch.bfloat16
    xq = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=7, block_size=64), identifier='in').double().cpu()
    want = xq @ pw.dequantize().double().cpu().t() + b.double().cpu()
    err = (got.double().cpu() - want).abs().max() / want.abs().max()
    assert float(err) < 1.2e-2, float(err)                              # bf16 output (+ bf16 bias add) rounding


@pytest.mark.parametrize("dname", ["bf16", "f16", "f32"])
def test_nm8_in_the_flat_kernel(dname):
    """N:8 on 16-bit dtypes runs inside the fused kernel (one lane item = one group): counted decision in registers,
    nth_element replay only when ties straddle the cut.  Real-valued rows (no straddling ties), coarse-grid rows (ties
    everywhere -> the replay), both orders, drop-in and packed outputs, sparsify-only, vs the oracle."""
    dt = DT[dname]
    assert native.is_fused(torch.empty(64, 1024, dtype=dt, device=DEV), 64, 4, 8)
    real = synth(96, 1024, dt)
    coarse = (synth(96, 1024, dt, 1.0, seed=5).float() * 2).round().div(2).to(dt)        # ~9 distinct magnitudes
    for xc, tag in ((real, "real"), (coarse, "coarse")):
        for N in (1, 2, 4, 5, 7):
            for first in ('s', 'q'):
                for blk, m in ((64, 3), (32, 7)):
                    c = cfg(mant_bits=m, block_size=blk, w_sparsity=True, N=N, M=8, first=first)
                    got = bfp_ops.float_to_bfp_blocked(xc.to(DEV), **c, identifier='w')
                    want = O.float_to_bfp_blocked(xc, **c, identifier='w')
                    assert_bits_equal(bits(got), bits(want), dt, f"{tag} {N}:8 first={first} b{blk} m{m}")
            got = bfp_ops._structured_N_M_sparsity(xc.to(DEV), DEV, N, 8)
            assert_bits_equal(bits(got), bits(O.structured_N_M_sparsity(xc, N, 8).view(xc.shape)), dt, f"{tag} {N}:8 sparsify only")
        # packed output (codes + exponents) of the same kernel
        codes, exps, deq = bfp_ops.float_to_bfp_packed(xc.to(DEV), 3, 64, N=4, M=8, first='s', code_bits=4, with_dequant=True)
        want = O.float_to_bfp_blocked(xc, **cfg(mant_bits=3, block_size=64, w_sparsity=True, N=4, M=8), identifier='w')
        assert_bits_equal(bits(deq), bits(want), dt, f"{tag} packed deq")
        back = bfp_ops.PackedBFP(codes, exps, xc.shape, dt, 3, 64, 4).dequantize()
        assert torch.equal(back.cpu().double(), want.double()), f"{tag} packed roundtrip"      # (a code has no -0)
    # special values: NaNs (one largest key), infinities, signed zeros, equal magnitudes of opposite sign -- sparsify only
    sp = synth(8, 64, dt, 1.0, seed=11)
    sp[0, 0:8] = torch.tensor([float('nan'), 1, -1, 1, float('nan'), -1, 0.0, -0.0], dtype=torch.float32).to(dt)
    sp[1, 8:16] = torch.tensor([float('inf'), -float('inf'), 2, -2, 2, 0.5, -0.5, 0.5], dtype=torch.float32).to(dt)
    sp[2, :] = 0
    sp[3, 0:8] = torch.tensor([-0.0, 0.0, -0.0, 0.0, 3, -3, 3, -3], dtype=torch.float32).to(dt)
    for N in (1, 3, 4, 6):
        got = bfp_ops._structured_N_M_sparsity(sp.to(DEV), DEV, N, 8)
        assert_bits_equal(bits(got), bits(O.structured_N_M_sparsity(sp, N, 8).view(sp.shape)), dt, f"special values {N}:8")
    # without the 16 MiB rank table the kernel replays nth_element for the ambiguous groups: same bits
    try:
        native.USE_NM8_TABLE = False
        c = cfg(mant_bits=3, block_size=64, w_sparsity=True, N=4, M=8, first='q')
        got = bfp_ops.float_to_bfp_blocked(coarse.to(DEV), **c, identifier='w')
        assert_bits_equal(bits(got), bits(O.float_to_bfp_blocked(coarse, **c, identifier='w')), dt, "4:8 replay path")
    finally:
        native.USE_NM8_TABLE = True
    # fp32 through the same kernel (a group spans two adjacent lane items), odd row count x ragged last sweep
    x32 = synth(67, 512, torch.float32)
    c = cfg(mant_bits=3, block_size=64, w_sparsity=True, N=4, M=8)
    assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(x32.to(DEV), **c, identifier='w')),
                      bits(O.float_to_bfp_blocked(x32, **c, identifier='w')), torch.float32, "f32 4:8")


def test_packed_general_path_matches_fused():
    """same tensor through the ragged-row kernels, forced by a storage offset that breaks 16-B alignment"""
    xc = synth(64, 512, torch.bfloat16)
    base = xc.to(DEV)
    buf = torch.empty(64 * 512 + 8, dtype=torch.bfloat16, device=DEV)
    xs = buf[1:1 + 64 * 512].view(64, 512)
    xs.copy_(base)
    assert xs.is_contiguous() and xs.data_ptr() % 16 != 0
    for first in ('s', 'q'):
        c1, e1, d1 = bfp_ops.float_to_bfp_packed(base, 3, 64, N=2, M=4, first=first, with_dequant=True)
        c2, e2, d2 = bfp_ops.float_to_bfp_packed(xs, 3, 64, N=2, M=4, first=first, with_dequant=True)
        assert torch.equal(c1, c2) and torch.equal(e1, e2) and torch.equal(d1, d2)
    # M = 8 goes through k_nm_rows + k_quant_rows; compare packed codes with the dequantised tensor
    codes, exps, deq = bfp_ops.float_to_bfp_packed(base, 3, 64, N=4, M=8, first='q', with_dequant=True)
    want = O.float_to_bfp_blocked(base.cpu(), **cfg(N=4, M=8, w_sparsity=True, first='q'), identifier='w')
    assert_bits_equal(bits(deq), bits(want), torch.bfloat16, "4:8 q")
    c = codes.cpu()
    q = torch.stack([(c & 0xF).to(torch.int16), (c >> 4).to(torch.int16)], dim=-1).view(64, 512)
    q = torch.where(q > 7, q - 16, q)
    e = exps.cpu().to(torch.float64).repeat_interleave(64, dim=1)
    assert torch.equal(q.to(torch.float64) * torch.pow(torch.tensor(2.0, dtype=torch.float64), e - 3), want.to(torch.float64))


# ---- size-independent properties at full size ----------------------------------------------------
def test_properties_full_size_headline():
    x = synth(4096, 11008, torch.bfloat16).to(DEV)
    c = cfg(w_sparsity=True)
    y = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
    # idempotence of S: pruning an already 2:4-sparse tensor changes nothing
    s1 = bfp_ops._structured_N_M_sparsity(x, 'cuda', 2, 4)
    assert torch.equal(bfp_ops._structured_N_M_sparsity(s1, 'cuda', 2, 4), s1)
    # idempotence of Q(S(.)) on every block whose largest mantissa is 5, 6 or 7: such a block keeps its
    # exponent (a block whose largest mantissa is <= 4 gets a smaller exponent the second time and its
    # maximum saturates -- that is the reference's behaviour too)
    codes, exps = bfp_ops.float_to_bfp_packed(x, 3, 64, N=2, M=4, first='s')
    lo, hi = (codes & 0xF).to(torch.int16), (codes >> 4).to(torch.int16)
    q = torch.stack([lo, hi], dim=-1).view(4096, 11008)
    q = torch.where(q > 7, q - 16, q).abs().view(-1, 64).max(dim=1)[0]
    stable = (q >= 5).repeat_interleave(64).view(4096, 11008)
    y2 = bfp_ops.float_to_bfp_blocked(y, **c, identifier='w')
    assert torch.equal(y[stable], y2[stable]) and int(stable.sum()) > y.numel() // 2
    # every group of 4 has at least 2 zeros; every block has at most 15 distinct magnitudes x sign
    assert int(((y.view(-1, 4) != 0).sum(dim=1) > 2).sum()) == 0
    # sign symmetry
    yn = bfp_ops.float_to_bfp_blocked(-x, **c, identifier='w')
    assert torch.equal(yn.view(torch.int16) & 0x7FFF, y.view(torch.int16) & 0x7FFF)
    # does not write its input
    x0 = x.clone()
    bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
    assert torch.equal(x, x0)


def test_empty_and_degenerate_inputs():
    c = cfg(w_sparsity=True)
    for shape in ((0, 64), (4, 0), (0,)):
        x = torch.empty(shape, dtype=torch.bfloat16, device=DEV)
        y = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
        assert y.shape == x.shape
    x = torch.zeros(8, 64, dtype=torch.bfloat16, device=DEV)
    assert int((bfp_ops.float_to_bfp_blocked(x, **c, identifier='w') != 0).sum()) == 0
    xh = torch.zeros(8, 64, dtype=torch.float16, device=DEV)                 # fp16 zero block -> NaN
    assert bool(torch.isnan(bfp_ops.float_to_bfp_blocked(xh, **cfg(), identifier='w')).all())
    # N == M keeps everything
    x = synth(8, 64, torch.float32).to(DEV)
    assert torch.equal(bfp_ops._structured_N_M_sparsity(x, 'cuda', 4, 4), x)
    # non-contiguous input
    xt = synth(64, 128, torch.float32).to(DEV).t()
    want = O.float_to_bfp_blocked(xt.cpu().contiguous(), **c, identifier='w')
    assert_bits_equal(bits(bfp_ops.float_to_bfp_blocked(xt, **c, identifier='w')), bits(want), torch.float32, "non-contiguous")


def test_stochastic_rounding_statistics():
    torch.manual_seed(0)
    x = synth(256, 1024, torch.bfloat16, scale=1.0).to(DEV)
    c = cfg(mant_bits=3, block_size=64, rounding_mode='stoc')
    ys = [bfp_ops.float_to_bfp_blocked(x, **c, identifier='w') for _ in range(64)]
    assert ys[0].dtype == torch.float32                                   # reference quirk: half in -> fp32 out
    assert not torch.equal(ys[0], ys[1])
    mean = torch.stack(ys).mean(0)
    det = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=3, block_size=64), identifier='w').float()
    step = (det - x.float()).abs().max()
    # unbiased away from the clamp: the mean of 64 draws is much closer to x than one rounding step
    inner = (x.float().abs() < 0.8 * x.float().abs().view(-1, 64).max(dim=1, keepdim=True)[0].repeat_interleave(64, dim=1).view_as(x))
    assert float(((mean - x.float()).abs()[inner]).mean()) < 0.2 * float(step)
    torch.manual_seed(5)
    a = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
    torch.manual_seed(5)
    b = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
    assert torch.equal(a, b)                                              # seeded from torch's generator


def test_graph_capture_and_side_stream():
    x = synth(1024, 4096, torch.bfloat16).to(DEV)
    c = cfg(w_sparsity=True)
    want = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
    s.synchronize()
    assert torch.equal(y, want)
    out = torch.empty_like(x)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        native.quantize_nm(x, 64, 3, 1e-8, N=2, M=4, sparsify_first=True, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_weight_cache_opt_in():
    """§8f next #1: the cached quantized weight == the re-quantized weight.  The cache is inference-only: it is bypassed
    while gradients are recorded for the weight and in training mode (optimizers write p.data in place without bumping the
    version counter), keyed on the configuration as well as on the parameter, and explicitly invalidatable."""
    kw = cfg(mant_bits=7, block_size=32, N=2, M=4, w_sparsity=True)
    lin = bfp_ops.BFPLinear(64, 128, True, **dict(kw)).to(DEV)
    ref = bfp_ops.BFPLinear(64, 128, True, **dict(kw)).to(DEV)
    ref.load_state_dict(lin.state_dict())
    lin.enable_weight_cache()
    cache = lin.linear_op.weight_cache
    x = synth(10, 64, torch.float32, 1.0).to(DEV).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    # training-style forward/backward: cache bypassed, gradients as without it
    y, yr = lin(x), ref(x2)
    assert torch.equal(y, yr) and cache.hits == 0 and cache.misses == 0
    y.sum().backward(); yr.sum().backward()
    assert torch.equal(x.grad, x2.grad) and torch.equal(lin.weight.grad, ref.weight.grad)     # straight-through to w
    # an optimizer-style update through .data does not bump the version counter; in training mode nothing is cached, so
    # the next forward sees the new weight
    lin.weight.data.add_(0.01); ref.weight.data.add_(0.01)
    assert torch.equal(lin(x), ref(x2))
    lin.eval(); ref.eval()
    with torch.no_grad():
        for it in range(3):
            assert torch.equal(lin(x), ref(x2))
        assert cache.misses == 1 and cache.hits == 2
        first = lin(x)
        assert first.grad_fn is None
        lin.weight.add_(0.01); ref.weight.add_(0.01)                                           # bumps weight._version
        assert torch.equal(lin(x), ref(x2)) and cache.misses == 2
        lin.weight.data.mul_(1.5); ref.weight.data.mul_(1.5)                                   # silent update: stale until invalidated
        cache.invalidate()
        assert torch.equal(lin(x), ref(x2)) and cache.misses == 3
        lin.bfp_args['mant_bits'] = 3; ref.bfp_args['mant_bits'] = 3                            # the configuration is part of the key
        assert torch.equal(lin(x), ref(x2)) and cache.misses == 4
        lin.train()
        h = cache.hits
        assert torch.equal(lin(x), ref(x2)) and cache.hits == h                                 # training mode: bypassed
    lin.enable_weight_cache(False)
    assert lin.linear_op.weight_cache is None and torch.equal(lin(x), ref(x2))


def test_g9_int_format_golden():
    """'int' per-channel format against the reference's own outputs (fp32 result for every dtype)"""
    g = load("g9_int.npz")
    shapes = {"w2": ((48, 200), 'w'), "a2": ((33, 96), 'in'), "a3": ((2, 7, 96), 'in'), "w4": ((8, 3, 5, 5), 'w'),
              "a4": ((2, 6, 4, 4), 'in'), "g2": ((16, 64), 'grad')}
    for name, (shape, ident) in shapes.items():
        for dname, dt in DT.items():
            x = from_bits(g[f"in_{name}_{dname}"], dt).view(shape).to(DEV)
            for nbits in (8, 4):
                c = cfg(sparsity_num_format='int', mant_bits=nbits, block_size=32)
                y = bfp_ops.float_to_bfp_blocked(x, **c, identifier=ident)
                assert y.dtype == torch.float32 and y.shape == x.shape
                assert_bits_equal(bits(y), g[f"out_{name}_{dname}_b{nbits}"], torch.float32, f"int {name} {dname} b{nbits}")
            if name in ("w2", "a3"):
                flag = 'w_sparsity' if ident == 'w' else 'in_sparsity'
                for first in ('s', 'q'):
                    c = cfg(sparsity_num_format='int', mant_bits=8, block_size=32, first=first, sparsity_mode='structured', N=2, M=4, **{flag: True})
                    y = bfp_ops.float_to_bfp_blocked(x, **c, identifier=ident)
                    assert_bits_equal(bits(y), g[f"comp_{name}_{dname}_{first}_s"], torch.float32, f"int comp {name} {dname} {first}")


def test_int_format_oracle_at_model_shapes():
    for rows, cols, dname, ident in ((4096, 4096, "bf16", 'w'), (11008, 4096, "f16", 'w'), (1024, 20000, "f32", 'w'), (8 * 197, 1024, "f32", 'in'),
                                     (520, 11008, "bf16", 'w'), (130, 16384, "f16", 'w'), (300, 6144, "f32", 'w'), (70, 3000, "bf16", 'w'),
                                     (33, 1024, "f32", 'w')):          # the register-resident row kernel, 2..8 items per thread
        dt = DT[dname]
        xc = synth(rows, cols, dt, 0.05 if ident == 'w' else 1.0)
        if rows == 70:
            xc[3, 17] = float('nan'); xc[5] = 0                       # a NaN row and an all-zero row
        c = cfg(sparsity_num_format='int', mant_bits=8, block_size=32)
        got = bfp_ops.float_to_bfp_blocked(xc.to(DEV), **c, identifier=ident)
        want = O.float_to_bfp_blocked(xc, **c, identifier=ident)
        assert_bits_equal(bits(got), bits(want), torch.float32, f"int {rows}x{cols} {dname} {ident}")


def test_int_format_activations_grid_rule_and_capture():
    """the 'int' ACTIVATION path (per-column min/max over all rows, int_ops.py:44-50): the wide min/max launch over row tiles, the flat
    quantize launch whose grid is a multiple of (C/4) / gcd(C/4, 256) (one column unit per thread on every trip), the row-batch launch
    behind it for column counts without such a grid, a NaN column, an all-zero column, a one-signed column, repeated calls, and a
    captured call -- all bit for bit against the oracle"""
    c = cfg(sparsity_num_format='int', mant_bits=8, block_size=32)
    cases = ((4096, 4096, "bf16"), (300, 11008, "bf16"), (1031, 768, "f16"), (8 * 197, 1024, "f32"), (5, 4096, "bf16"), (3, 64, "f32"), (129, 8 * 2053, "bf16"),
             (2, 4 * 521, "f32"), (700, 1000, "f16"), (64, 6, "f32"), (2500, 512, "bf16"))   # (C/4) / gcd: 4, 43, 3, 1, 4, 1, 2053 (row-batch launch), 521, 125, scalar kernels, 1
    for rows, cols, dname in cases:
        dt = DT[dname]
        xc = synth(rows, cols, dt, 1.0, seed=rows + cols)
        if rows == 300:
            xc[7, 40] = float('nan'); xc[:, 41] = 0; xc[:, 42] = xc[:, 42].abs()       # a NaN column, a zero column, a one-signed column
        want = bits(O.float_to_bfp_blocked(xc, **c, identifier='in'))
        x = xc.to(DEV)
        for rep in range(2):
            got = bfp_ops.float_to_bfp_blocked(x, **c, identifier='in')
            assert got.dtype == torch.float32
            assert_bits_equal(bits(got), want, torch.float32, f"int act [{rows},{cols}] {dname} call {rep}")
    x = synth(512, 4096, torch.bfloat16, 1.0, seed=5).to(DEV)
    want = bits(O.float_to_bfp_blocked(x.cpu(), **c, identifier='in'))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        bfp_ops.float_to_bfp_blocked(x, **c, identifier='in')
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            y = bfp_ops.float_to_bfp_blocked(x, **c, identifier='in')
        g.replay(); g.replay()
    torch.cuda.synchronize()
    assert_bits_equal(bits(y), want, torch.float32, "int act captured")


def test_int_format_weights_on_rounding_boundaries():
    """the per-row 'int' kernel rounds with x * (1 / scale) where that provably equals x / scale and redoes a wave's unit with the true division
    otherwise: rows built so that MANY elements sit exactly on half-integers of the grid (x = (n + 0.5) * scale with scale a power of two or
    scale = 2 max / 255 from a row maximum chosen to make it so), one ulp beside them, on +-max, zeros, inf / NaN rows -- bit for bit against the oracle"""
    c = cfg(sparsity_num_format='int', mant_bits=8, block_size=32)
    g = torch.Generator().manual_seed(77)
    for dname, dt in DT.items():
        rows, cols = 64, 2048
        x = torch.zeros(rows, cols, dtype=torch.float32)
        for r in range(rows):
            mx = float(2.0 ** (r % 7 - 3)) * (255.0 if r % 2 == 0 else 1.0 + (r % 5) / 8.0)       # even rows: scale = 2 mx / 255 is a power of two
            n = torch.randint(-127, 127, (cols,), generator=g).float()
            scale = 2.0 * mx / 255.0
            x[r] = (n + 0.5) * scale                                                                # half-integers of the row's grid (before the dtype rounds them)
            x[r, :8] = torch.tensor([mx, -mx, 0.0, -0.0, mx * 0.5, -mx * 0.5, scale * 0.5, -scale * 1.5])
        xc = x.to(dt)
        b = bits(xc).reshape(rows, cols)
        absmask = 0x7FFFFFFF if dt == torch.float32 else 0x7FFF
        b[:, 1024:1536] += 1                                       # one ulp above, and
        b[:, 1536:] = np.where((b[:, 1536:] & absmask) > 0, b[:, 1536:] - 1, b[:, 1536:])   # one ulp below the pattern (magnitude-wise)
        xc = from_bits(b, dt).view(rows, cols).clone()
        xc[60, 100] = float('inf'); xc[61, 7] = float('nan'); xc[62] = 0; xc[63] = -xc[63].abs()
        got = bfp_ops.float_to_bfp_blocked(xc.to(DEV), **c, identifier='w')
        want = O.float_to_bfp_blocked(xc, **c, identifier='w')
        assert_bits_equal(bits(got), bits(want), torch.float32, f"int weights on rounding boundaries {dname}")


def test_random_configs_vs_oracle():
    """120 seeded random (shape, dtype, block, mantissa, N:M / unstructured, order) cases: exercises the dispatch
    between the fused kernel, the threshold kernel and the ragged-row fallbacks against the oracle"""
    import random
    rnd = random.Random(2024)
    dts = list(DT.items())
    for case in range(120):
        dname, dt = rnd.choice(dts)
        rows = rnd.choice([1, 2, 3, 7, 16, 33, 64])
        cols = rnd.choice([1, 5, 8, 16, 24, 63, 64, 65, 96, 128, 200, 256, 520])
        blk = rnd.choice([4, 8, 16, 32, 48, 64, 128])
        m = rnd.choice([1, 2, 3, 4, 5, 7])
        mode = rnd.choice(['none', 'structured', 'structured', 'unstructured'])
        first = rnd.choice(['s', 'q'])
        N, M = rnd.choice([(1, 2), (2, 4), (1, 4), (3, 4), (2, 8), (4, 8), (3, 6), (5, 7), (1, 16)])
        frac = rnd.choice([0.1, 0.5, 0.75])
        scale = rnd.choice([0.02, 1.0, 37.0])
        xc = synth(rows, cols, dt, scale, seed=1000 + case)
        if rnd.random() < 0.3:
            xc = (xc.float() * 8).round().div(8).to(dt)                       # coarse grid: many ties
        c = cfg(mant_bits=m, block_size=blk, first=first, N=N, M=M, sparsity_frac=frac,
                w_sparsity=(mode != 'none'), sparsity_mode=('unstructured' if mode == 'unstructured' else 'structured'))
        what = f"case {case}: [{rows},{cols}] {dname} b{blk} m{m} {mode} {N}:{M} {frac} first={first}"
        got = bfp_ops.float_to_bfp_blocked(xc.to(DEV), **c, identifier='w')
        assert got.shape == xc.shape and got.dtype == dt, what
        if mode != 'unstructured':
            assert_bits_equal(bits(got), bits(O.float_to_bfp_blocked(xc, **c, identifier='w')), dt, what)
        else:
            # tie-class contract: compare piecewise with the engine's own pruning step
            if first == 's':
                ys = bfp_ops._unstructured_sparsity(xc.to(DEV), 'cuda', frac)
                _tie_class_check(xc, ys, frac, dt, what)
                assert_bits_equal(bits(got), bits(O.no_sparsity_float_to_bfp(ys.cpu(), blk, m)), dt, what)
            else:
                xq = O.no_sparsity_float_to_bfp(xc, blk, m)
                _tie_class_check(xq, got, frac, dt, what)


def test_packed_format_dequantize_and_safetensors(tmp_path):
    """§8f next #3: packed HBFP -> tensor decode, and the on-disk form (safetensors) round trip"""
    for dname, m, blk, cols in (("bf16", 3, 64, 1024), ("f16", 7, 32, 1024), ("f32", 3, 16, 520), ("bf16", 3, 64, 1000)):
        dt = DT[dname]
        xc = synth(128, cols, dt)
        x = xc.to(DEV)
        for first in ('s', 'q'):
            p = bfp_ops.PackedBFP.quantize(x, m, blk, N=2, M=4, first=first)
            want = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=m, block_size=blk, w_sparsity=True, first=first), identifier='w')
            got = p.dequantize()
            assert got.dtype == dt and got.shape == x.shape
            assert torch.equal(got.float(), want.float())                    # value-equal (-0.0 == +0.0)
            nz = want != 0
            assert_bits_equal(bits(got).reshape(-1)[nz.cpu().view(-1).numpy()], bits(want).reshape(-1)[nz.cpu().view(-1).numpy()], dt, "packed decode")
            f = tmp_path / f"w_{dname}_{first}.safetensors"
            p.save(str(f))
            q = bfp_ops.PackedBFP.load(str(f), DEV)
            assert q.shape == p.shape and q.dtype == dt and q.mant_bits == m and q.block_size == blk
            assert torch.equal(q.dequantize(), got)
            assert p.nbytes() < x.numel() * x.element_size() * (0.27 if m <= 3 else 0.53) * (1 if dt != torch.float32 else 0.5) + 4096
    # NaN blocks survive the packed form
    xh = torch.zeros(4, 64, dtype=torch.float16, device=DEV)
    assert bool(torch.isnan(bfp_ops.PackedBFP.quantize(xh, 3, 64).dequantize()).all())


def test_packed_compact24(tmp_path):
    """§8f next #3 (optional part): 2:4 compaction of the 4-bit codes -- 2 nibbles + 2 x 2-bit positions per group of 4 --
    is an exact round trip for both orders, survives the safetensors file, and refuses a tensor that is not 2:4 sparse"""
    import os
    xc = synth(256, 1024, torch.bfloat16)
    x = xc.to(DEV)
    for first in ('s', 'q'):
        p = bfp_ops.PackedBFP.quantize(x, 3, 64, N=2, M=4, first=first)
        vals, idx = native.compact24(p.codes)
        assert vals.numel() == p.codes.numel() // 2 and idx.numel() == p.codes.numel() // 4
        back = native.expand24(vals, idx).view(p.codes.shape)
        assert torch.equal(back, p.codes), first
        f, g = tmp_path / f"c24_{first}.safetensors", tmp_path / f"plain_{first}.safetensors"
        p.save(str(f), compact24=True)
        p.save(str(g))
        q = bfp_ops.PackedBFP.load(str(f), DEV)
        assert torch.equal(q.codes, p.codes) and torch.equal(q.exps, p.exps) and q.shape == p.shape
        assert os.path.getsize(f) < 0.8 * os.path.getsize(g)                 # (0.375 + 1/64) / (0.5 + 1/64) = 0.76
    # hand-made groups: 0, 1, 2 non-zeros at every position pair, negative codes
    codes = torch.tensor([0x00, 0x00, 0x0f, 0x00, 0x00, 0x90, 0x71, 0x00, 0x07, 0x0a, 0x30, 0x0c, 0x00, 0xf1, 0x50, 0x00],
                         dtype=torch.uint8, device=DEV)
    v, i = native.compact24(codes)
    assert torch.equal(native.expand24(v, i), codes)
    dense = bfp_ops.PackedBFP.quantize(x, 3, 64)                              # no pruning: most groups have 3-4 non-zeros
    with pytest.raises(ValueError):
        native.compact24(dense.codes)


def test_large_tensor_parity():
    """0.5 GiB bf16 weight (268 M elements, ~200 sweeps per workgroup): full-tensor oracle comparison for the fused
    kernel, and the unstructured path's counts at that size"""
    rows, cols = 16384, 16384
    g = torch.Generator().manual_seed(4321)
    xc = (torch.randn(rows, cols, generator=g, dtype=torch.float32) * 0.02).to(torch.bfloat16)
    x = xc.to(DEV)
    c = cfg(w_sparsity=True)
    got = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
    want = O.float_to_bfp_blocked(xc, **c, identifier='w')
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))
    del want
    ys = bfp_ops._unstructured_sparsity(x, 'cuda', 0.5)
    assert int((ys == 0).sum()) == rows * cols // 2                       # exactly k pruned (no zeros in the input)
    kept_min = float(ys.float().abs()[ys != 0].min())
    pruned_max = float(x.float().abs()[ys == 0].max())
    assert pruned_max <= kept_min


def test_torch_library_ops_and_compile():
    """the engine's entry points are registered torch ops: opcheck's fake-tensor rule holds and a patched module
    compiles (aot_eager backend: no code generation involved) to the same result as eager"""
    x = synth(64, 256, torch.bfloat16).to(DEV)
    y = torch.ops.bfpq.fake_quantize(x, 64, 3, 1e-8, 2, 4, True, 0)
    want = O.float_to_bfp_blocked(x.cpu(), **cfg(w_sparsity=True), identifier='w')
    assert_bits_equal(bits(y), bits(want), torch.bfloat16, "registered op")
    torch.library.opcheck(torch.ops.bfpq.fake_quantize, (x, 64, 3, 1e-8, 2, 4, True, 0), test_utils=("test_schema", "test_faketensor"))
    lin = bfp_ops.BFPLinear(256, 128, True, **cfg(mant_bits=7, block_size=32, w_sparsity=True)).to(DEV).to(torch.bfloat16)
    eager = lin(x)
    compiled = torch.compile(lin, backend="aot_eager", fullgraph=True)(x)
    assert torch.equal(eager, compiled)


@pytest.mark.parametrize("N,K,T,dname", [(64, 256, 1, "bf16"), (256, 512, 5, "bf16"), (128, 1024, 16, "f16"), (4096, 11008, 16, "bf16"), (11008, 4096, 3, "bf16"), (4096, 4096, 1, "f32"),
                                           (11008, 4096, 16, "bf16"), (8192, 512, 9, "f16"), (4096, 384, 16, "f32"),
                                           (4096, 11008, 64, "bf16"), (512, 1024, 33, "f32"), (11008, 4096, 17, "f16")])
def test_packed_consumer_decode_linear(N, K, T, dname):
    """§8f next #3: out = x @ W^T from the PACKED weight with integer block dot products (int8 MFMA), against the
    same product of the fake-quantised tensors in fp64"""
    dt = DT[dname]
    w = synth(N, K, dt).to(DEV)
    x = synth(T, K, dt, 1.0, seed=9).to(DEV)
    pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
    got = pw.linear_decode(x)
    assert got.shape == (T, N) and got.dtype == dt
    wq = pw.dequantize().double().cpu()
    xq = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=7, block_size=64), identifier='in').double().cpu()
    want = xq @ wq.t()
    err = (got.double().cpu() - want).abs().max() / want.abs().max()
    tol = {"f32": 2e-6, "bf16": 6e-3, "f16": 8e-4}[dname]          # output rounding of the dtype; the sums themselves are fp32 of exact block sums
    assert float(err) < tol, (float(err), tol)
    # fp32 output: only the cross-block fp32 accumulation order separates it from the fp64 reference
    if K % 256 == 0 and T <= 16:
        got32 = native.hbfp_linear_decode(x, pw.codes, pw.exps, 3, 7, out_dtype=torch.float32)
        err32 = (got32.double().cpu() - want).abs().max() / want.abs().max()
        assert float(err32) < 2e-6, float(err32)
    if K % 128 == 0 and K >= 256:                                   # HBFP4 activations (the W4A4 pairing of an HBFP4 config)
        g4 = pw.linear_decode(x, x_mant_bits=3)
        xq4 = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=3, block_size=64), identifier='in').double().cpu()
        want4 = xq4 @ wq.t()
        e4 = (g4.double().cpu() - want4).abs().max() / want4.abs().max()
        assert float(e4) < tol, ("W4A4", float(e4), tol)
    if K % 128 == 0:                                                # the MFMA-tiled weight layout: same sums, other slice order
        tiles, expt = native.mfma_tiles(pw.codes, pw.exps)
        got32t = native.hbfp_linear_decode_tiled(x, tiles, expt, N, 3, 7, out_dtype=torch.float32)
        err32t = (got32t.double().cpu() - want).abs().max() / want.abs().max()
        assert float(err32t) < 2e-6, float(err32t)
        try:                                                        # every tiles-per-wave variant gives the same sums
            for rt in (1, 2, 4):
                native.load_library().bfpq_tune(1, rt)
                g = native.hbfp_linear_decode_tiled(x, tiles, expt, N, 3, 7, out_dtype=torch.float32)
                assert torch.equal(g, got32t) or float((g.double().cpu() - want).abs().max() / want.abs().max()) < 2e-6, rt
        finally:
            native.load_library().bfpq_tune(1, 0)


@pytest.mark.parametrize("N,K,T,dname", [(256, 512, 200, "bf16"), (384, 768, 129, "bf16"), (1024, 1024, 300, "f16"), (200, 256, 1, "f32"),
                                          (4096, 4096, 512, "bf16"), (100, 256, 130, "bf16"), (1500, 1280, 700, "bf16")])
def test_packed_consumer_prefill_mx8(N, K, T, dname):
    """§8f next #3, prefill: out = Q(x) @ Q(W)^T for many tokens from the packed weight on the block-scaled matrix
    instruction (e4m3 mantissas + E8M0 block scales: exact block dot products, fp32 across blocks), against the same product of
    the two fake-quantised tensors in fp64.  Ragged T and N (tiles past the edge), bias, HBFP4 and HBFP5 activations."""
    dt = DT[dname]
    w = synth(N, K, dt).to(DEV)
    x = synth(T, K, dt, 1.0, seed=9).to(DEV)
    x[0, :64] = 0                                                     # a zero block (exponent -26, all-zero mantissas)
    x[T - 1, 64:128] *= 4096                                          # and a block on a very different scale
    bias = synth(1, N, dt, 1.0, seed=3).to(DEV).view(N)
    pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
    wq = pw.dequantize().double().cpu()
    w8, wsc = native.mx8_from_hbfp(pw.codes, pw.exps, K, 3, 4)
    for xm in (3, 4):
        assert native.hbfp_linear_mx8_ok(T, N, K, 3, xm)
        xq = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=xm, block_size=64), identifier='in').double().cpu()
        want = xq @ wq.t()
        got32 = native.hbfp_linear_mx8(x, w8, wsc, xm, out_dtype=torch.float32)
        assert got32.shape == (T, N)
        g = got32.double().cpu()
        nan = torch.isnan(want)                                      # fp16: the zero block is a NaN block in the reference (A.2) -> NaN row
        assert torch.equal(torch.isnan(g), nan) and bool(nan.any()) == (dname == "f16")
        err32 = (g - want)[~nan].abs().max() / want[~nan].abs().max()
        assert float(err32) < 2e-6, (xm, float(err32))              # only the fp32 accumulation order across blocks differs
    # through the module-level entry point, with bias, in the tensor's dtype
    got = pw.linear(x, bias, x_mant_bits=3, decode_tokens=0)
    assert got.shape == (T, N) and got.dtype == dt
    xq = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=3, block_size=64), identifier='in').double().cpu()
    want = xq @ wq.t() + bias.double().cpu()
    nan = torch.isnan(want)
    err = (got.double().cpu() - want)[~nan].abs().max() / want[~nan].abs().max()
    tol = {"f32": 2e-6, "bf16": 6e-3, "f16": 8e-4}[dname]
    assert float(err) < tol and torch.equal(torch.isnan(got.cpu()), nan), (float(err), tol)
    # 8-bit activations do not fit e4m3: the entry point must take the decode-and-library route, not a wrong product
    assert not native.hbfp_linear_mx8_ok(T, N, K, 3, 7)
    # projections that share an input quantize it once (one-entry memo keyed on the tensor and its version counter)
    native._last_image.clear()
    g1 = native.hbfp_linear_mx8(x, w8, wsc, 3, out_dtype=torch.float32)
    img = native._last_image[x.device][2][0]
    g2 = native.hbfp_linear_mx8(x, w8, wsc, 3, out_dtype=torch.float32)
    assert native._last_image[x.device][2][0] is img and torch.equal(torch.nan_to_num(g1), torch.nan_to_num(g2))
    x.mul_(2)                                                          # in-place update: the memo must miss
    g3 = native.hbfp_linear_mx8(x, w8, wsc, 3, out_dtype=torch.float32)
    assert native._last_image[x.device][2][0] is not img
    assert torch.equal(torch.nan_to_num(g3), torch.nan_to_num(g1 * 2))   # (a power-of-two scale moves every block exponent by one)
    x.div_(2)
    # every tile variant walks K in the same order: bit-identical results
    try:
        ref = None
        for v in (0, 1, 2, 3, 4, 5, 6, -1):
            assert native.load_library().bfpq_tune(2, v) == 0
            g = native.hbfp_linear_mx8(x, w8, wsc, 3, out_dtype=torch.float32)
            ref = g if ref is None else ref
            assert torch.equal(torch.nan_to_num(g), torch.nan_to_num(ref)), v
    finally:
        native.load_library().bfpq_tune(2, -1)


@pytest.mark.parametrize("dname", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("xm", [3, 4, 2])
def test_quantize_mx8_equals_codes_then_image(dname, xm):
    """the one-pass activation image (bfpq_quantize_mx8: lean arithmetic, e4m3 bytes by table look-up) against the two-step
    route (int8 codes + exponents, which the oracle pins, then bfpq_mx8_from_hbfp), byte for byte -- ordinary blocks, a zero
    block, blocks on extreme scales (cold tier), an inf block"""
    dt = DT[dname]
    x = synth(300, 1024, dt, 1.0, seed=21)
    x[0, :64] = 0
    x[1, 64:128] *= 3e4 if dname == "f16" else 1e30
    x[2, :64] *= 1e-7 if dname == "f16" else 1e-35
    x[3, 5] = float("inf")
    x[4, 64:128] = 0.75                                               # ties at the rounding boundary of every element
    x = x.to(DEV)
    x8, xs = native.quantize_mx8(x, xm)
    xc = torch.empty((300, 1024), dtype=torch.int8, device=DEV)
    xe = torch.empty((300, 16), dtype=torch.int8, device=DEV)
    native.quantize_nm(x, 64, xm, 1e-8, want_deq=False, code_bits=8, want_exp=True, codes_out=xc, exps_out=xe)
    w8, ws = native.mx8_from_hbfp(xc, xe, 1024, xm, 8)
    assert torch.equal(xs, ws)
    nanblk = (ws == 255).repeat_interleave(64, dim=1)                  # mantissas of a NaN block are "don't care"
    assert bool(nanblk.any())
    a, b = x8.clone(), w8.clone()
    a[nanblk] = 0; b[nanblk] = 0
    # a two's-complement int8 code has no -0: compare magnitudes and the signs of the non-zero mantissas
    assert torch.equal(a & 0x7f, b & 0x7f)
    nz = (b & 0x7f) != 0
    assert torch.equal(a[nz], b[nz])
    # and against the oracle's dequantised values: mantissa * 2^(scale - 127)
    import numpy as np
    e4 = np.zeros(256, dtype=np.float64)
    for m, byte in enumerate([0, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e] + [0x50 + i for i in range(8)]):
        e4[byte] = m; e4[byte | 0x80] = -m
    val = torch.from_numpy(e4[x8.cpu().numpy()]) * torch.exp2(xs.cpu().double() - 127).repeat_interleave(64, dim=1)
    want = O.float_to_bfp_blocked(x.cpu(), **cfg(mant_bits=xm, block_size=64, device='cpu'), identifier='in').double()
    ok = ~nanblk.cpu() & torch.isfinite(want) & (xs.cpu().repeat_interleave(64, dim=1) > 0)
    assert torch.equal(val[ok], want[ok])


def test_packed_unstructured_weight_on_the_matrix_unit():
    """cfg 4's weight form (HBFP4 + 50 % unstructured, prune then quantize) packed by the fused prune + quantize launch: the codes decode to
    the drop-in result (pinned to the oracle's tie-class contract elsewhere), and a prefill from them equals the fp64 product"""
    dt = torch.bfloat16
    w = synth(384, 1024, dt).to(DEV)
    x = synth(200, 1024, dt, 1.0, seed=9).to(DEV)
    c = cfg(mant_bits=3, block_size=64, w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=0.5, first='s')
    pw = bfp_ops.PackedBFP.quantize_unstructured(w, 3, 64, 0.5)
    wq = bfp_ops.float_to_bfp_blocked(w, **c, identifier='w')
    assert torch.equal(pw.dequantize().abs(), wq.abs())                   # (a packed code has no -0)
    assert int((wq == 0).sum()) >= w.numel() // 2
    got = pw.linear(x, x_mant_bits=3, decode_tokens=0).double().cpu()
    xq = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=3, block_size=64), identifier='in').double().cpu()
    want = xq @ wq.double().cpu().t()
    assert float((got - want).abs().max() / want.abs().max()) < 6e-3
    lin = bfp_ops.BFPLinear(1024, 384, False, **dict(c)).to(DEV).to(dt).eval().enable_weight_cache(matrix_unit=True)
    with torch.no_grad():
        lin.weight.copy_(w)
        got2 = lin(x).double().cpu()
    assert lin.linear_op.weight_cache.mx_calls == 1
    assert float((got2 - want).abs().max() / want.abs().max()) < 6e-3


def test_packed_consumer_prefill_random_shapes():
    """24 random problems (tokens 1-900, features 8-1300, K a multiple of 256 up to 2304; dtypes, N:M patterns, mantissa widths, bias or
    not) through PackedBFP.linear on the matrix unit -- whatever tile plan / K split the shape picks -- against the fp64 product of the
    oracle-pinned fake-quantised operands"""
    import random
    rng = random.Random(20260)
    for case in range(24):
        T, N, K = rng.randint(1, 900), rng.randint(1, 160) * 8 + rng.choice([0, 0, 0, 4]), 256 * rng.randint(1, 9)
        dname = rng.choice(["bf16", "bf16", "f16", "f32"])
        dt = DT[dname]
        wm, xm = rng.choice([3, 3, 2, 1, 4]), rng.choice([3, 3, 4, 2])
        nm = rng.choice([(0, 0), (2, 4), (1, 4), (4, 8)])
        w = synth(N, K, dt, 0.02, seed=100 + case).to(DEV)
        x = synth(T, K, dt, rng.choice([1.0, 0.05, 30.0]), seed=200 + case).to(DEV)
        bias = synth(1, N, dt, 1.0, seed=300 + case).to(DEV).view(N) if rng.random() < 0.5 else None
        pw = bfp_ops.PackedBFP.quantize(w, wm, 64, N=nm[0], M=nm[1])
        got = pw.linear(x, bias, x_mant_bits=xm, decode_tokens=0)
        assert got.shape == (T, N) and got.dtype == dt
        xq = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=xm, block_size=64), identifier='in').double().cpu()
        want = xq @ pw.dequantize().double().cpu().t()
        if bias is not None:
            want = want + bias.double().cpu()
        nan = torch.isnan(want)
        assert torch.equal(torch.isnan(got.cpu()), nan), (case, T, N, K, dname)
        if bool((~nan).any()):
            err = float((got.double().cpu() - want)[~nan].abs().max() / want[~nan].abs().max().clamp_min(1e-30))
            assert err < {"f32": 2e-6, "bf16": 6e-3, "f16": 8e-4}[dname], (case, T, N, K, dname, wm, xm, nm, err)


@pytest.mark.parametrize("N,K,T", [(512, 4096, 100), (256, 11008, 128), (384, 1024, 33)])
def test_packed_consumer_prefill_split_k(N, K, T):
    """short token counts: K split over several workgroups per output tile, fp32 slabs added in part order -- against the fp64 product,
    equal to the unsplit kernel up to the fp32 summation order, and reproducible"""
    dt = torch.bfloat16
    w = synth(N, K, dt).to(DEV)
    x = synth(T, K, dt, 1.0, seed=9).to(DEV)
    bias = synth(1, N, dt, 1.0, seed=3).to(DEV).view(N)
    pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
    w8, wsc = native.mx8_from_hbfp(pw.codes, pw.exps, K, 3, 4)
    parts = native.load_library().bfpq_hbfp_linear_mx8_parts(T, N, K)
    assert parts > 1
    xq = bfp_ops.float_to_bfp_blocked(x, **cfg(mant_bits=3, block_size=64), identifier='in').double().cpu()
    want = xq @ pw.dequantize().double().cpu().t() + bias.double().cpu()
    got = native.hbfp_linear_mx8(x, w8, wsc, 3, bias=bias, out_dtype=torch.float32)
    assert float((got.double().cpu() - want).abs().max() / want.abs().max()) < 2e-6
    assert torch.equal(got, native.hbfp_linear_mx8(x, w8, wsc, 3, bias=bias, out_dtype=torch.float32))
    try:
        native.SPLIT_K = False
        one = native.hbfp_linear_mx8(x, w8, wsc, 3, bias=bias, out_dtype=torch.float32)
    finally:
        native.SPLIT_K = True
    assert float((got - one).abs().max() / want.abs().max()) < 2e-6
    gb = native.hbfp_linear_mx8(x, w8, wsc, 3, bias=bias)                 # bf16 output through the reduce launch
    assert gb.dtype == dt and float((gb.double().cpu() - want).abs().max() / want.abs().max()) < 6e-3


@pytest.mark.parametrize("dname", ["bf16", "f32"])
def test_bfplinear_cached_on_the_matrix_unit(dname):
    """opt-in enable_weight_cache(matrix_unit=True): the reference's module (bfp_ops.py:270-287) with its Linear on the block-scaled
    matrix instruction -- against the fp64 product of the oracle-pinned fake-quantised operands, and against the module's ordinary
    forward; short inputs, training mode and unsupported configurations keep the ordinary route"""
    dt = DT[dname]
    c = cfg(mant_bits=3, block_size=64, w_sparsity=True, N=2, M=4)
    lin = bfp_ops.BFPLinear(512, 384, True, **dict(c)).to(DEV).to(dt).eval()
    with torch.no_grad():
        lin.weight.copy_(synth(384, 512, dt).to(DEV)); lin.bias.copy_(synth(1, 384, dt, 1.0, seed=5).view(384).to(DEV))
    x = synth(3 * 50, 512, dt, 1.0, seed=11).view(3, 50, 512).to(DEV)
    with torch.no_grad():
        ref = lin(x)                                                      # F.linear on the two fake-quantised operands
        lin.enable_weight_cache(matrix_unit=True)
        got = lin(x)
        cache = lin.linear_op.weight_cache
        assert cache.mx_calls == 1 and got.shape == ref.shape and got.dtype == dt
        xq = bfp_ops.float_to_bfp_blocked(x, **c, identifier='in').double().cpu().view(-1, 512)
        wq = bfp_ops.float_to_bfp_blocked(lin.weight, **c, identifier='w').double().cpu()
        want = (xq @ wq.t() + lin.bias.double().cpu()).view(3, 50, 384)
        tol = {"f32": 2e-6, "bf16": 6e-3}[dname]
        assert float((got.double().cpu() - want).abs().max() / want.abs().max()) < tol
        assert float((ref.double().cpu() - want).abs().max() / want.abs().max()) < max(tol, 2e-5)   # (the library GEMM: its own summation order)
        got2 = lin(x)                                                     # weight image cached
        assert cache.mx_calls == 2 and torch.equal(got, got2)
        short = lin(x[:1, :8])                                            # 8 tokens: the ordinary route
        assert cache.mx_calls == 2 and torch.equal(short, ref[:1, :8])
        lin.weight.mul_(0.5)                                              # in-place update: new image
        got3 = lin(x)
        assert cache.mx_calls == 3 and not torch.equal(got3, got)
        lin.train()
        lin(x)
        assert cache.mx_calls == 3                                        # training mode bypasses the cache altogether
    lin8 = bfp_ops.BFPLinear(512, 384, False, **dict(cfg(mant_bits=7, block_size=64))).to(DEV).to(dt).eval().enable_weight_cache(matrix_unit=True)
    with torch.no_grad():
        lin8(x)
    assert lin8.linear_op.weight_cache.mx_calls == 0                      # HBFP8 does not fit e4m3: ordinary route
    c5 = cfg(mant_bits=4, block_size=64)                                  # HBFP5 (|mantissa| <= 15, int8 codes) still fits e4m3
    lin5 = bfp_ops.BFPLinear(512, 384, False, **dict(c5)).to(DEV).to(dt).eval().enable_weight_cache(matrix_unit=True)
    with torch.no_grad():
        lin5.weight.copy_(synth(384, 512, dt).to(DEV))
        got5 = lin5(x)
        want5 = (bfp_ops.float_to_bfp_blocked(x, **c5, identifier='in').double().cpu().view(-1, 512)
                 @ bfp_ops.float_to_bfp_blocked(lin5.weight, **c5, identifier='w').double().cpu().t()).view(3, 50, 384)
    assert lin5.linear_op.weight_cache.mx_calls == 1
    assert float((got5.double().cpu() - want5).abs().max() / want5.abs().max()) < {"f32": 2e-6, "bf16": 6e-3}[dname]


def test_dist_paths_with_the_native_engine_on_rccl():
    """dist.py end to end on the device with the HIP engine and RCCL (backend "nccl"), one rank: the gloo tests cover the
    multi-rank protocol with a stand-in engine, this covers the real kernels, streams and collectives behind the same
    functions -- sharded quantize (+ gather), the unstructured exchange (histogram all-gather), the
    side-stream overlapped gather and the packed-bytes gather."""
    import torch.distributed as dist
    from quantization_sparsity_interplay_amd import dist as D
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    import socket
    with socket.socket() as sk:                                         # a free port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        xc = synth(256, 1024, torch.bfloat16)
        x = xc.to(DEV)
        c = cfg(w_sparsity=True)
        single = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
        assert torch.equal(D.float_to_bfp_blocked_sharded(x, 256, gather=True, identifier='w', **c), single)
        for first in ('s', 'q'):
            cu = cfg(w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=0.5, first=first)
            a = D.float_to_bfp_blocked_sharded(x, 256, gather=True, identifier='w', **cu)
            assert_bits_equal(bits(a), bits(bfp_ops.float_to_bfp_blocked(x, **cu, identifier='w')), torch.bfloat16, f"sharded unstructured {first}")
        g = D.gather_overlapped(x, 256, lambda p: bfp_ops.float_to_bfp_blocked(p, **c, identifier='w'), chunks=4)
        assert torch.equal(g, single)
        pk = D.all_gather_packed(x, 256, 3, 64, N=2, M=4)
        assert torch.equal(pk.float(), single.float())
        codes, exps = D.float_to_bfp_packed_sharded(x, 256, 3, 64, gather=True, N=2, M=4)
        c1, e1 = bfp_ops.float_to_bfp_packed(x, 3, 64, N=2, M=4)
        assert torch.equal(codes, c1) and torch.equal(exps, e1)
    finally:
        dist.destroy_process_group()


def _flat_order_prune(xc, frac):
    """the engine's own rule, restated in numpy: everything below the k-th smallest magnitude, plus the first `need`
    elements EQUAL to it in flat index order (NaNs form one largest class, as in ATen's comparator)"""
    dt = xc.dtype
    if dt == torch.float32:
        keys = xc.contiguous().view(torch.int32).numpy().reshape(-1).astype(np.int64) & 0x7FFFFFFF
        inf = 0x7F800000
    else:
        keys = xc.contiguous().view(torch.int16).numpy().view(np.uint16).reshape(-1).astype(np.int64) & 0x7FFF
        inf = 0x7F80 if dt == torch.bfloat16 else 0x7C00
    keys = np.minimum(keys, inf + 1)
    k = int(xc.numel() * frac)
    out = xc.contiguous().view(-1).clone()
    if k == 0:
        return out.view(xc.shape)
    tau = np.partition(keys, k - 1)[k - 1]
    below = keys < tau
    eq = keys == tau
    need = k - int(below.sum())
    rank = np.cumsum(eq) - 1
    out[torch.from_numpy(below | (eq & (rank < need)))] = 0
    return out.view(xc.shape)


@pytest.mark.parametrize("dname", ["bf16", "f16", "f32"])
def test_unstructured_exact_flat_order(dname):
    """the three-launch unstructured path against the engine's tie rule restated on the host, bit for bit: small and large
    tensors, ragged element counts (element loads), tie-heavy inputs (huge tie classes: the cut falls inside a segment and
    a piece), every sparsity fraction, and the quantizer fused behind it."""
    dt = DT[dname]
    shapes = [(4, 64), (64, 256), (257, 1024), (333, 777), (2048, 4096), (5120, 5120)]
    for rows, cols in shapes:
        for tag in ("real", "coarse"):
            xc = synth(rows, cols, dt, 1.0, seed=rows + cols)
            if tag == "coarse":
                xc = (xc.float() * 4).round().div(4).to(dt)                     # ~20 distinct magnitudes
            x = xc.to(DEV)
            for frac in ((0.5,) if rows > 1000 and tag == "real" else (0.1, 0.5, 0.9)):
                want = _flat_order_prune(xc, frac)
                got = bfp_ops._unstructured_sparsity(x, 'cuda', frac)
                assert_bits_equal(bits(got), bits(want), dt, f"prune [{rows},{cols}] {tag} frac={frac}")
                if cols % 64 == 0:
                    c = cfg(w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=frac, first='s')
                    q = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
                    assert_bits_equal(bits(q), bits(O.no_sparsity_float_to_bfp(want, 64, 3)), dt, f"prune+quantize [{rows},{cols}] {tag} frac={frac}")


def test_unstructured_window_miss_falls_back_to_recount():
    """A tensor whose segments live on wildly different scales: the per-segment histogram windows (2048 bins around the
    segment's own quantile) do not cover the global threshold everywhere, so the resolve launch recounts those segments and
    the apply launch ranks the whole cut segment -- slower, same bits.  Also all-zero and constant tensors."""
    rows, cols = 2048, 4096
    g = torch.Generator().manual_seed(7)
    base = torch.randn(rows, cols, generator=g)
    scale = torch.logspace(-30, 30, rows, base=2.0).view(rows, 1)               # 60 binades across the rows
    for dt in (torch.bfloat16, torch.float16):
        xc = (base * (scale if dt == torch.bfloat16 else scale.clamp(2.0 ** -12, 2.0 ** 12))).to(dt)
        xc[100:140] = xc[5]                                                      # repeated rows: ties far apart in flat order
        xc[::7, ::3] = 0
        x = xc.to(DEV)
        for frac in (0.3, 0.5, 0.77):
            got = bfp_ops._unstructured_sparsity(x, 'cuda', frac)
            assert_bits_equal(bits(got), bits(_flat_order_prune(xc, frac)), dt, f"heterogeneous rows {dt} frac={frac}")
        ws = bfp_ops._workspace(x.device)
        st = ws.read_state()
        assert st["done"] == 1
    for val in (0.0, 0.37):
        xc = torch.full((512, 1024), val, dtype=torch.bfloat16)
        got = bfp_ops._unstructured_sparsity(xc.to(DEV), 'cuda', 0.5)
        assert_bits_equal(bits(got), bits(_flat_order_prune(xc, 0.5)), torch.bfloat16, f"constant {val}")


@pytest.mark.parametrize("dname", ["bf16", "f32"])
def test_unstructured_on_already_sparse_tensors(dname):
    """A heavy atom next to the threshold: a tensor that is already (about) half zeros, pruned by half again -- a pre-pruned checkpoint
    going through the path on every forward.  The segment's own quantile then lies in the zero bin in one segment and thousands of
    bins higher in the next; every segment publishes a SECOND window for that case (without it half the segments were recounted by the
    last workgroup: 5.3 ms instead of 41 us on [5120,5120]).  Bit for bit against the engine's tie rule on the host, and the time of
    one call is held to a generous bound so that the recount path cannot come back unnoticed."""
    dt = DT[dname]
    g = torch.Generator().manual_seed(11)
    rows, cols = 2048, 4096
    base = torch.randn(rows, cols, generator=g) * 0.02
    for tag, keep in (("49.9 % zeros", 0.501), ("50.1 % zeros", 0.499), ("30 % zeros", 0.7), ("70 % zeros", 0.3)):
        xc = (base * (torch.rand(rows, cols, generator=g) < keep)).to(dt)
        x = xc.to(DEV)
        for frac in (0.5, 0.3):
            got = bfp_ops._unstructured_sparsity(x, 'cuda', frac)
            assert_bits_equal(bits(got), bits(_flat_order_prune(xc, frac)), dt, f"{tag} frac={frac}")
    # pruned by this engine, then pruned again at the same fraction (the threshold sits exactly at the zero / non-zero boundary)
    x = (torch.randn(5120, 5120, generator=torch.Generator(device=DEV).manual_seed(3), device=DEV) * 0.02).to(dt)
    once = bfp_ops._unstructured_sparsity(x, 'cuda', 0.5)
    twice = bfp_ops._unstructured_sparsity(once, 'cuda', 0.5)
    assert np.array_equal(bits(twice), bits(once)) and int((once == 0).sum()) >= x.numel() // 2
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        bfp_ops._unstructured_sparsity(once, 'cuda', 0.5)
    e1.record()
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) / 10 < 1.0, f"{e0.elapsed_time(e1) / 10:.3f} ms per call: the serial recount path is back"


@pytest.mark.parametrize("dname", ["bf16", "f16", "f32"])
def test_unstructured_on_already_quantized_tensors(dname):
    """A tensor that went through the quantizer BEFORE it is pruned (first = 'q', or a checkpoint stored in HBFP values): a hundred-odd distinct
    magnitudes whose low mantissa bits are all zero.  The selection's LDS histogram maps its bins through a bank swizzle (unmixed, every atomic
    of a wave fell into bank 0) and counts a segment whose keys all have five zero low bits into 32 replicas per bin; windows, flush and recount
    read through the same map.  Bit for bit against the engine's tie rule
    on the host, for HBFP4 and HBFP8 values, and the composed q-first call against the oracle's contract."""
    dt = DT[dname]
    xc = synth(1536, 4096, dt, 0.02, seed=21)
    for m in (3, 7):
        q = O.no_sparsity_float_to_bfp(xc, 64, m)
        for frac in (0.5, 0.25):
            got = bfp_ops._unstructured_sparsity(q.to(DEV), 'cuda', frac)
            assert_bits_equal(bits(got), bits(_flat_order_prune(q, frac)), dt, f"already HBFP{m + 1} {dname} frac={frac}")
        if m == 3:
            # coarse-keyed at the start of every segment, a few full-precision elements further in: the segments that meet one are counted
            # into replicas first and again the plain way; segments without one keep their replicas -- the two kinds publish the same counts
            mixed = q.clone().view(-1)
            mixed[700001::900001] = xc.view(-1)[700001::900001] * 1.37
            mixed = mixed.view(q.shape)
            got = bfp_ops._unstructured_sparsity(mixed.to(DEV), 'cuda', 0.5)
            assert_bits_equal(bits(got), bits(_flat_order_prune(mixed, 0.5)), dt, f"HBFP4 with stray full-precision elements {dname}")
        c = cfg(w_sparsity=True, sparsity_mode='unstructured', first='q', mant_bits=m)
        yq = bfp_ops.float_to_bfp_blocked(xc.to(DEV), **c, identifier='w')
        _tie_class_check(q, yq, 0.5, dt, f"q-first unstructured HBFP{m + 1} {dname}")
        assert int((yq == 0).sum()) >= xc.numel() // 2


@pytest.mark.parametrize("shape", [(13824, 5120), (5120, 13824)])
def test_oracle_parity_cfg4_13b_mlp_shapes(shape):
    """cfg 4 at the LLaMA-13B MLP shapes (gate/up [13824,5120], down [5120,13824]) bf16, HBFP4 + 50 % unstructured,
    sparsify -> quantize: the oracle's contract (same threshold and count, identical outside the tie class; reference
    bfp_ops.py:61-71) and the quantizer on the engine's own pruned tensor (reference :35-59)."""
    xc = synth(shape[0], shape[1], torch.bfloat16)
    x = xc.to(DEV)
    ys = bfp_ops._unstructured_sparsity(x, 'cuda', 0.5)
    _tie_class_check(xc, ys, 0.5, torch.bfloat16, f"cfg4 {shape} s")
    assert int((ys == 0).sum()) == shape[0] * shape[1] // 2
    c = cfg(w_sparsity=True, sparsity_mode='unstructured', first='s')
    yq = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
    want = O.no_sparsity_float_to_bfp(ys.cpu(), 64, 3)
    assert_bits_equal(bits(yq), bits(want), torch.bfloat16, f"cfg4 {shape} s then q")


def test_unstructured_graph_replay_odd_call_count():
    """the histogram buffers alternate on the device, so a captured graph holding an ODD number of calls replays correctly"""
    xs = [synth(512, 1024, torch.bfloat16, seed=s).to(DEV) for s in (1, 2, 3)]
    c = cfg(w_sparsity=True, sparsity_mode='unstructured', first='s')
    want = [bfp_ops.float_to_bfp_blocked(x, **c, identifier='w').clone() for x in xs]
    outs = None
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        bfp_ops.float_to_bfp_blocked(xs[0], **c, identifier='w')              # (creates this stream's workspace outside the capture)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            outs = [bfp_ops.float_to_bfp_blocked(x, **c, identifier='w') for x in xs]
        for _ in range(3):
            graph.replay()
    torch.cuda.synchronize()
    for o, w in zip(outs, want):
        assert torch.equal(o, w)


@pytest.mark.parametrize("dname", ["bf16", "f16"])
def test_every_block_max_pattern_dropin(dname):
    """drop-in quantizer over EVERY finite 16-bit magnitude as the block max (all exponent windows, the epsilon-dominated
    tiny maxima, subnormals, the largest finite values) with full-range and tiny block mates, both signs, for several
    mantissa widths and epsilons: the lean hot path, the general fast path and the step-by-step replay must all give the
    oracle's bits (reference bfp_ops.py:29-44)."""
    dt = DT[dname]
    hi = 0x7F80 if dname == "bf16" else 0x7C00
    pat = np.arange(0, hi, dtype=np.uint16)
    n = pat.size
    rng = np.random.default_rng(5)
    blk = np.zeros((n, 64), dtype=np.uint16)
    blk[:, 0] = pat
    for j in range(1, 64):                                   # mates: magnitudes <= the max, random exponent drop, random sign
        drop = rng.integers(0, 40, size=n).astype(np.int64) * (1 << (7 if dname == "bf16" else 10))
        m = np.maximum(pat.astype(np.int64) - drop - rng.integers(0, 128, size=n), 0)
        blk[:, j] = (m.astype(np.uint16)) | (rng.integers(0, 2, size=n).astype(np.uint16) << 15)
    blk[:, 0] |= (rng.integers(0, 2, size=n).astype(np.uint16) << 15)
    xc = from_bits(blk.reshape(-1), dt).view(n, 64)
    x = xc.to(DEV)
    for m, eps in ((3, 1e-8), (7, 1e-8), (1, 1e-8), (8, 1e-8), (3, 0.0), (5, 1e-3)):
        got = bfp_ops._no_sparsity_float_to_bfp(x, 64, m, eps, 'determ', 'cuda')
        want = O.no_sparsity_float_to_bfp(xc, 64, m, eps)
        assert_bits_equal(bits(got), bits(want), dt, f"{dname} m={m} eps={eps}")
    got = bfp_ops._no_sparsity_float_to_bfp(x.view(-1, 16), 16, 3, 1e-8, 'determ', 'cuda')          # other lane-group widths
    assert_bits_equal(bits(got), bits(O.no_sparsity_float_to_bfp(xc.view(-1, 16), 16, 3, 1e-8)), dt, f"{dname} block 16")


@pytest.mark.parametrize("dname", ["bf16", "f16"])
def test_every_block_max_pattern_packed_only(dname):
    """the packed-only instantiation (4-bit codes + int8 exponents, no dequantised tensor: north_star's "packed int4 stores")
    over EVERY finite 16-bit magnitude as the block max: codes straight from the signed lean arithmetic (clamp, magic-constant
    add, low nibble), exponent bytes by the one-lane-per-block byte store, general tiers packed inside their branch -- decoded
    as code * 2^(e - m), they must equal the oracle's values; dense, 2:4 sparsify-first and 2:4 quantize-first."""
    dt = DT[dname]
    hi = 0x7F80 if dname == "bf16" else 0x7C00
    pat = np.arange(0, hi, dtype=np.uint16)
    n = pat.size
    rng = np.random.default_rng(9)
    blk = np.zeros((n, 64), dtype=np.uint16)
    blk[:, 0] = pat
    for j in range(1, 64):
        drop = rng.integers(0, 40, size=n).astype(np.int64) * (1 << (7 if dname == "bf16" else 10))
        m_ = np.maximum(pat.astype(np.int64) - drop - rng.integers(0, 128, size=n), 0)
        blk[:, j] = (m_.astype(np.uint16)) | (rng.integers(0, 2, size=n).astype(np.uint16) << 15)
    blk[:, 0] |= (rng.integers(0, 2, size=n).astype(np.uint16) << 15)
    blk[:, [0, 37]] = blk[:, [37, 0]]
    xc = from_bits(blk.reshape(-1), dt).view(n, 64)
    x = xc.to(DEV)
    for m in (3, 2, 1):
        for N, M, first in ((0, 0, 's'), (2, 4, 's'), (2, 4, 'q')):
            codes, exps = bfp_ops.float_to_bfp_packed(x, m, 64, N=N, M=M, first=first, code_bits=4)
            c = cfg(mant_bits=m, block_size=64, w_sparsity=N > 0, N=N or 2, M=M or 4, first=first)
            want = O.float_to_bfp_blocked(xc, **c, identifier='w').to(torch.float64)
            cc = codes.cpu()
            q = torch.stack([(cc & 0xF).to(torch.int16), (cc >> 4).to(torch.int16)], dim=-1).view(n, 64)
            q = torch.where(q > 7, q - 16, q).to(torch.float64)
            e = exps.cpu().to(torch.float64).view(n, 1)
            val = q * torch.pow(torch.tensor(2.0, dtype=torch.float64), e - m)
            nanblk = exps.cpu().view(n) == -128                               # the reference's NaN blocks (fp16: zero block)
            ok = torch.isnan(want).any(dim=1) == nanblk                        # (quantize-first: pruning after it zeroes part of a NaN block)
            assert bool(ok.all()), (dname, m, N, first, "NaN-block markers")
            good = ~nanblk
            sat = (exps.cpu().view(n).abs() == 127) & good                     # exponents beyond int8: the byte saturates, values are lost by design
            good &= ~sat
            good &= ~torch.isinf(want).any(dim=1)                              # fp16 block max >= 61440: the reference's 2^16 overflows to inf and so
                                                                               # do its top values (code 8 x 2^13); a 4-bit code cannot say "inf"
            assert torch.equal(val[good], want[good]), (dname, m, N, first, int((val[good] != want[good]).sum()))
            assert int((q.abs() > (1 << m) - 1).sum()) == 0


@pytest.mark.parametrize("dname", ["bf16", "f16", "f32"])
def test_batched_list_equals_per_tensor_and_oracle(dname):
    """bfpq_fake_quantize_batched (a list of tensors per launch) == the per-tensor call == the oracle: mixed shapes incl.
    ragged ones (handled one by one inside the call), an empty tensor, more than 64 tensors (two launches), 3-D inputs,
    and the N:M switch per tensor (a Linear's activation next to its weight)."""
    dt = DT[dname]
    shapes = [(64, 256), (3, 64), (257, 1024), (5, 100), (0, 128), (2, 7, 128), (768, 768), (33, 192)] + [(4 + i, 64 * (1 + i % 3)) for i in range(66)]
    xs_c = [synth(int(np.prod(sh[:-1])), sh[-1], dt, 0.5, seed=10 + i).view(sh) for i, sh in enumerate(shapes)]
    xs = [x.to(DEV) for x in xs_c]
    for c in (cfg(w_sparsity=True), cfg(w_sparsity=True, first='q', mant_bits=7, block_size=32), cfg()):
        got = bfp_ops.float_to_bfp_blocked_many(xs, identifier='w', **c)
        assert len(got) == len(xs)
        for i, (g, x, xc) in enumerate(zip(got, xs, xs_c)):
            assert g.shape == x.shape and g.dtype == dt
            assert_bits_equal(bits(g), bits(bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')), dt, f"batched vs single #{i} {tuple(x.shape)}")
            if i < 8:
                assert_bits_equal(bits(g), bits(O.float_to_bfp_blocked(xc, **c, identifier='w')), dt, f"batched vs oracle #{i} {tuple(x.shape)}")
    # per-tensor N:M switch: activation dense, weight 2:4 -- what one BFPLinear forward issues
    f = native.FastQuant(64, 3, 1e-8, 2, 4, True)
    a, w = xs[2], xs[0]
    qa, qw = f.many([a, w], [False, True])
    assert_bits_equal(bits(qa), bits(O.float_to_bfp_blocked(xs_c[2], **cfg(), identifier='in')), dt, "pair: activation")
    assert_bits_equal(bits(qw), bits(O.float_to_bfp_blocked(xs_c[0], **cfg(w_sparsity=True), identifier='w')), dt, "pair: weight")
    lin = bfp_ops.BFPLinear(256, 64, True, **cfg(w_sparsity=True)).to(DEV).to(dt)
    x = synth(10, 256, dt, 1.0).to(DEV)
    with torch.no_grad():
        y = lin(x)
        bfp_ops.FUSE_OPERAND_PAIR = False
        try:
            y2 = lin(x)
        finally:
            bfp_ops.FUSE_OPERAND_PAIR = True
    assert torch.equal(y, y2)


def test_list_with_large_tensors_over_two_streams():
    """bfpq_fake_quantize_list: tensors of 24 MB or more get launches of their own, alternating between the caller's stream and
    the side stream; small ones share list launches.  A mixed list (large, small, ragged, large) == the per-tensor call, eagerly,
    captured into a hipGraph and replayed on new data, from a non-default stream, and through the one-stream C entry point."""
    dt = torch.bfloat16
    shapes = [(4096, 4096), (64, 256), (3072, 4096), (5, 100), (2048, 8192), (768, 768), (4096, 3072)]
    c = cfg(w_sparsity=True)
    g = torch.Generator(device=DEV).manual_seed(5)
    xs = [(torch.randn(r, k, generator=g, device=DEV) * 0.02).to(dt) for r, k in shapes]
    want = [bfp_ops.float_to_bfp_blocked(x, **c, identifier='w').clone() for x in xs]
    for i in (0, 2):                                                     # (the large ones against the oracle on a slab)
        assert_bits_equal(bits(want[i][:64].cpu()), bits(O.float_to_bfp_blocked(xs[i][:64].cpu(), **c, identifier='w')), dt, f"large #{i} vs oracle")
    got = bfp_ops.float_to_bfp_blocked_many(xs, identifier='w', **c)
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"eager list #{i}"
    prep = bfp_ops.PreparedMany(xs, identifier='w', **c)
    bound = prep.run()                                                   # (the list's own output tensors, rewritten by every run)
    for y in bound:
        y.zero_()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        prep.run()
    with torch.no_grad():
        for x in xs:
            x.mul_(1.3)
    want2 = [bfp_ops.float_to_bfp_blocked(x, **c, identifier='w').clone() for x in xs]
    gr.replay()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(bound, want2)):
        assert torch.equal(a, b), f"graph replay #{i}"
    s1 = torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        outs = [y.clone() for y in prep.run()]
    torch.cuda.current_stream().wait_stream(s1)
    for i, (a, b) in enumerate(zip(outs, want2)):
        assert torch.equal(a, b), f"side-stream caller #{i}"
    # the one-stream entry point (aux_stream NULL): same results
    f = native.FastQuant(64, 3, 1e-8, 2, 4, True)
    pl = native.PreparedList(f, xs)
    pl._aux = None
    for i, (a, b) in enumerate(zip(pl.run(), want2)):
        assert torch.equal(a, b), f"one stream #{i}"


def test_list_without_a_list_kernel_goes_over_streams():
    """float_to_bfp_blocked_many for configurations the library has no list form for (quantize-first unstructured pruning, the 'int'
    format, 4:8): large tensors are dealt to a few streams (bfp_ops._many_over_streams) -- same bytes as the per-tensor calls, also
    from a non-default stream and with a small tensor in between."""
    dt = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(9)
    xs = [(torch.randn(r, k, generator=g, device=DEV) * 0.02).to(dt) for r, k in ((4096, 4096), (3072, 4096), (64, 256), (2048, 8192), (4096, 3072))]
    for c in (cfg(w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=0.4, first='q'),
              cfg(sparsity_num_format='int', mant_bits=8),
              cfg(w_sparsity=True, N=4, M=8)):
        want = [bfp_ops.float_to_bfp_blocked(x, **c, identifier='w').clone() for x in xs]
        got = bfp_ops.float_to_bfp_blocked_many(xs, identifier='w', **c)
        for i, (a, b) in enumerate(zip(got, want)):
            assert a.dtype == b.dtype and torch.equal(a, b), (c['sparsity_mode'], c['sparsity_num_format'], c['M'], i)
        s1 = torch.cuda.Stream()
        s1.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            got = bfp_ops.float_to_bfp_blocked_many(xs, identifier='w', **c)
        torch.cuda.current_stream().wait_stream(s1)
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), ("side-stream caller", c['sparsity_mode'], c['sparsity_num_format'], c['M'], i)


def test_prepared_list_reruns_in_place():
    """PreparedMany: descriptors and outputs bound once; run() after an in-place weight update gives the new results"""
    ws = [synth(64 + 8 * i, 256, torch.bfloat16, 0.5, seed=i).to(DEV) for i in range(5)]
    c = cfg(w_sparsity=True)
    prep = bfp_ops.PreparedMany(ws, identifier='w', **c)
    a = [y.clone() for y in prep.run()]
    for w, y in zip(ws, a):
        assert torch.equal(y, bfp_ops.float_to_bfp_blocked(w, **c, identifier='w'))
    with torch.no_grad():
        for w in ws:
            w.mul_(1.7)
    b = prep.run()
    for w, y in zip(ws, b):
        assert torch.equal(y, bfp_ops.float_to_bfp_blocked(w, **c, identifier='w'))


def test_tensor_above_4gb_is_cut_into_pieces():
    """A 5 GiB bf16 tensor (the drop-in kernels address through 32-bit buffer offsets, so the launcher cuts it into
    2 GiB pieces): rows are independent, so every row slab must equal the same rows quantized on their own -- dense, 2:4 and
    4:8 -- and a sample of rows is held to the oracle."""
    rows, cols = 40960, 65536                                           # 2.68 G elements = 5 GiB
    g = torch.Generator(device=DEV).manual_seed(77)
    x = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
    for r0 in range(0, rows, 4096):
        x[r0:r0 + 4096] = (torch.randn(4096, cols, generator=g, device=DEV) * 0.02).to(torch.bfloat16)
    for c in (cfg(w_sparsity=True), cfg(), cfg(w_sparsity=True, N=4, M=8)):
        y = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
        for r0 in (0, 12288, 16384 - 64, 32768 - 8, rows - 4096):          # slabs inside pieces and across the piece boundaries
            part = bfp_ops.float_to_bfp_blocked(x[r0:r0 + 4096].contiguous(), **c, identifier='w')
            assert torch.equal(y[r0:r0 + 4096], part), (c['N'], c['M'], r0)
        sample = torch.cat([x[16383:16385], x[-2:]]).cpu()
        want = O.float_to_bfp_blocked(sample, **c, identifier='w')
        got = torch.cat([y[16383:16385], y[-2:]]).cpu()
        assert_bits_equal(bits(got), bits(want), torch.bfloat16, f"rows around a piece boundary {c['N']}:{c['M']}")
        del y


def test_unstructured_on_two_streams_concurrently():
    """the select workspace is per (device, stream): two unstructured calls in flight on two streams must not see each
    other's threshold state"""
    xs = [synth(2048, 4096, torch.bfloat16, 0.02, seed=s).to(DEV) for s in (21, 22)]
    cs = [cfg(w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=f, first='s') for f in (0.3, 0.7)]
    want = [bfp_ops.float_to_bfp_blocked(x, **c, identifier='w').clone() for x, c in zip(xs, cs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [None, None]
    for rep in range(5):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                got[i] = bfp_ops.float_to_bfp_blocked(xs[i], **cs[i], identifier='w')
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), rep
    assert bfp_ops._workspace.__doc__ and len({id(ws) for ws in bfp_ops._select_ws.values()}) >= 2
