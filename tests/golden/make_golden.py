#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

Run once here (never on the GPU box -- /root/reference does not exist there):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It loads the reference's own src/transformers/bfp/bfp_ops.py (+ its sibling int_ops.py) under a
synthetic package name -- nothing else of the fork is imported -- calls the reference functions on
CPU with rounding_mode='determ', and stores inputs and outputs as raw bit patterns in .npz files
(numpy arrays only, loadable with allow_pickle=False).  Fixture groups follow SURVEY.md §8c G1..G8.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen import int_bits_tensor  # noqa: E402  (shared integer-only input generator)

REF = "/root/reference/src/transformers/bfp"


def load_ref():
    pkg = types.ModuleType("refbfp")
    pkg.__path__ = [REF]
    sys.modules["refbfp"] = pkg

    def load(name):
        spec = importlib.util.spec_from_file_location(f"refbfp.{name}", f"{REF}/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refbfp.{name}"] = m
        spec.loader.exec_module(m)
        return m
    return load("bfp_ops")


DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def bits(t):
    """raw bit pattern of a float tensor as a numpy unsigned array"""
    t = t.contiguous()
    if t.dtype == torch.float32:
        return t.view(torch.int32).numpy().view(np.uint32).copy()
    return t.view(torch.int16).numpy().view(np.uint16).copy()


def from_bits(a, dtype):
    if dtype == torch.float32:
        return torch.from_numpy(a.astype(np.uint32).view(np.int32).copy()).view(torch.float32)
    return torch.from_numpy(a.astype(np.uint16).view(np.int16).copy()).view(dtype)


def cfg(**kw):
    base = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cpu', block_size=64,
                num_format='bfp', weight_mant_bits=15, in_sparsity=False, w_sparsity=False,
                grad_sparsity=False, sparsity_frac=0.5, N=2, M=4, sparsity_num_format='bfp',
                first='s', sparsity_mode='structured')
    base.update(kw)
    return base


def main():
    ops = load_ref()
    torch.set_num_threads(4)
    out = {}

    # ---- G1: shared-exponent tables -------------------------------------------------------
    g1 = {}
    # every non-negative finite bf16 / fp16 pattern as a 1-element block max
    for name, hi in (("bf16", 0x7F80), ("f16", 0x7C00)):
        pat = np.arange(0, hi, dtype=np.uint16)
        t = from_bits(pat, DT[name]).view(-1, 1)
        e = ops.get_exponent(t, 1e-8).float().view(-1).numpy()
        g1[f"{name}_e"] = e.astype(np.float32)          # index = bit pattern; may hold -inf
    # fp32: around every power of two, the first 12 mantissa steps, plus random mantissas
    ks = np.arange(1, 255, dtype=np.uint32)              # biased exponents of normals
    js = np.concatenate([np.arange(0, 12, dtype=np.uint32),
                         np.array([0x400000, 0x7FFFFF, 0x7FFFFE, 0x123456, 0x2AAAAA, 0x555555], dtype=np.uint32)])
    pat = ((ks[:, None] << 23) | js[None, :]).reshape(-1)
    pat = np.concatenate([pat, np.array([0, 1, 2, 0x7FFFFF, 0x400000], dtype=np.uint32)])   # zero + subnormals
    t = from_bits(pat, torch.float32).view(-1, 1)
    g1["f32_bits"] = pat
    g1["f32_e"] = ops.get_exponent(t, 1e-8).view(-1).numpy().astype(np.float32)
    # other epsilons (the kernel takes epsilon as a runtime argument)
    for eps_name, eps in (("1e-6", 1e-6), ("0", 0.0)):
        pat16 = np.arange(0, 0x7F80, 37, dtype=np.uint16)
        g1[f"bf16_eps{eps_name}_bits"] = pat16
        g1[f"bf16_eps{eps_name}_e"] = ops.get_exponent(from_bits(pat16, torch.bfloat16).view(-1, 1), eps).float().view(-1).numpy()
    # torch.pow(2.0, e) exactness probe: stored so the oracle's pow restatement is pinned too
    ee = torch.arange(-150, 129, dtype=torch.float32)
    g1["pow2_e"] = ee.numpy()
    g1["pow2_f32"] = bits(torch.pow(2.0, ee))
    g1["pow2_bf16"] = bits(torch.pow(2.0, ee.to(torch.bfloat16)))
    g1["pow2_f16"] = bits(torch.pow(2.0, ee.clamp(-40, 20).to(torch.float16)))
    np.savez_compressed(os.path.join(HERE, "g1_exponent.npz"), **g1)

    # ---- G2: dense HBFP quantizer ---------------------------------------------------------
    g2 = {}
    gen = torch.Generator().manual_seed(1234)
    base = torch.randn(16, 192, generator=gen)
    for sname, scale in (("s0.02", 0.02), ("s1", 1.0), ("s30", 30.0)):
        for dname, dt in DT.items():
            x = (base * scale).to(dt)
            g2[f"in_{sname}_{dname}"] = bits(x)
            for blk in (16, 32, 64):
                for m in (3, 5, 7, 15):
                    y = ops._no_sparsity_float_to_bfp(x, blk, m, 1e-8, 'determ', 'cpu')
                    assert y.dtype == dt and y.shape == x.shape
                    g2[f"out_{sname}_{dname}_b{blk}_m{m}"] = bits(y)
    np.savez_compressed(os.path.join(HERE, "g2_quantize.npz"), **g2)

    # ---- G3: N:M masks (tie rule) ---------------------------------------------------------
    g3 = {}
    # exhaustive rows over alphabet {0,1,2,3}, M = 4
    rows4 = np.array(np.meshgrid(*[np.arange(4)] * 4, indexing="ij")).reshape(4, -1).T.astype(np.float32)
    g3["m4_rows"] = rows4.astype(np.uint8)
    for N in (1, 2, 3):
        y = ops._structured_N_M_sparsity(torch.from_numpy(rows4 + 1.0), 'cpu', N, 4)   # +1 so kept values are non-zero
        g3[f"m4_keep_N{N}"] = (y != 0).numpy().astype(np.uint8)
    rng = np.random.RandomState(7)
    for (N, M) in ((2, 8), (4, 8), (1, 8), (7, 8), (4, 16), (8, 16), (2, 16), (16, 32), (8, 32), (1, 2), (3, 6), (2, 5)):
        r = rng.randint(0, 4, size=(4096, M)).astype(np.uint8)
        r[:64] = rng.randint(0, 2, size=(64, M))           # even tie-heavier
        r[64:128] = np.sort(rng.randint(0, 64, size=(64, M)), axis=1)          # ascending
        r[128:192] = np.sort(rng.randint(0, 64, size=(64, M)), axis=1)[:, ::-1]  # descending
        sign = rng.randint(0, 2, size=r.shape) * 2 - 1
        x = torch.from_numpy((r.astype(np.float32) + 1.0) * sign)
        y = ops._structured_N_M_sparsity(x, 'cpu', N, M)
        g3[f"rows_{N}_{M}"] = r
        g3[f"sign_{N}_{M}"] = (sign < 0).astype(np.uint8)
        g3[f"keep_{N}_{M}"] = np.packbits((y != 0).numpy().astype(np.uint8), axis=1)
    # median-of-3 killer style rows: push introselect toward its depth limit (heap_select branch)
    for M in (16, 32, 64):
        rows = []
        for rep in range(64):
            k = M // 2
            a = np.zeros(M, dtype=np.int64)
            for i in range(1, k + 1):                      # classic anti-median-of-3 permutation
                if i % 2 == 1:
                    a[i - 1] = i
                    a[i] = k + i
                a[k + i - 1] = 2 * i
            a = np.roll(a, rep % M) if rep else a
            rows.append(a)
        r = np.array(rows).astype(np.float32)
        N = M // 2
        y = ops._structured_N_M_sparsity(torch.from_numpy(r + 1.0), 'cpu', N, M)
        g3[f"killer_rows_{M}"] = r.astype(np.uint8)
        g3[f"killer_keep_{M}"] = np.packbits((y != 0).numpy().astype(np.uint8), axis=1)
    # real-valued rows incl. negative values, +-0 and a NaN-free inf
    xr = torch.randn(512, 16, generator=gen)
    xr[0, :4] = torch.tensor([0.0, -0.0, 0.0, -0.0])
    xr[1, :4] = torch.tensor([float('inf'), 1.0, -float('inf'), 2.0])
    for dname, dt in DT.items():
        x = xr.to(dt)
        g3[f"real_in_{dname}"] = bits(x)
        for (N, M) in ((2, 4), (1, 4), (3, 4), (4, 8), (2, 16)):
            g3[f"real_out_{dname}_{N}_{M}"] = bits(ops._structured_N_M_sparsity(x, 'cpu', N, M))
    np.savez_compressed(os.path.join(HERE, "g3_nm.npz"), **g3)

    # ---- G4: composed float_to_bfp_blocked ------------------------------------------------
    g4 = {}
    base = torch.randn(32, 256, generator=gen) * 0.02
    for dname, dt in DT.items():
        x = base.to(dt)
        g4[f"in_{dname}"] = bits(x)
        for first in ('s', 'q'):
            for mode, extra in (("structured", dict(N=2, M=4)), ("structured", dict(N=1, M=4)),
                                ("structured", dict(N=4, M=8)), ("unstructured", dict(sparsity_frac=0.5)),
                                ("unstructured", dict(sparsity_frac=0.3))):
                for m, blk in ((3, 64), (7, 32), (7, 16)):
                    c = cfg(mant_bits=m, block_size=blk, first=first, sparsity_mode=mode, w_sparsity=True, **extra)
                    y = ops.float_to_bfp_blocked(x, **c, identifier='w')
                    tag = f"{dname}_{first}_{mode[:1]}_{extra.get('N', 0)}_{extra.get('M', 0)}_{extra.get('sparsity_frac', 0)}_m{m}_b{blk}"
                    g4[f"out_{tag}"] = bits(y)
        # identifier / flag selection (bfp_ops.py:132-139) and the fp32 'format' (sparsify only)
        for ident in ('w', 'in', 'grad', ''):
            for flag in ('in_sparsity', 'w_sparsity', 'grad_sparsity'):
                c = cfg(**{flag: True})
                g4[f"ident_{dname}_{ident or 'none'}_{flag}"] = bits(ops.float_to_bfp_blocked(x, **c, identifier=ident))
        c = cfg(sparsity_num_format='fp32', w_sparsity=True)
        g4[f"fp32fmt_{dname}"] = bits(ops.float_to_bfp_blocked(x, **c, identifier='w'))
        c = cfg(w_sparsity=False, weight_mant_bits=15)
        g4[f"sgd_{dname}"] = bits(ops.float_to_bfp_blocked(x, **c, identifier='', sgd_update=True))
    np.savez_compressed(os.path.join(HERE, "g4_composed.npz"), **g4)

    # ---- G5: unstructured (threshold, count, tie class) ------------------------------------
    g5 = {}
    x = (torch.randn(64, 256, generator=gen) * 0.02).to(torch.bfloat16)
    g5["small_in"] = bits(x)
    for frac in (0.5, 0.25, 0.9, 0.001):
        y = ops._unstructured_sparsity(x, 'cpu', frac)
        g5[f"small_out_{frac}"] = bits(y)
    # big, tie-heavy, integer-generated input (regenerated by the tests from tests/golden/gen.py)
    for dname, shape, seed in (("bf16", (512, 1024), 11), ("f16", (256, 512), 12), ("f32", (256, 512), 13)):
        xb = int_bits_tensor(shape, dname, seed)
        x = from_bits(xb, DT[dname]).view(shape)
        y = ops._unstructured_sparsity(x, 'cpu', 0.5)
        g5[f"big_zero_{dname}"] = np.packbits((y == 0).numpy().reshape(-1).astype(np.uint8))
    np.savez_compressed(os.path.join(HERE, "g5_unstructured.npz"), **g5)

    # ---- G6: padding (cols not multiples of block / M) --------------------------------------
    g6 = {}
    for C in (100, 6, 65, 1, 63, 129):
        x = torch.randn(5, C, generator=gen)
        for dname, dt in DT.items():
            xd = x.to(dt)
            g6[f"in_{C}_{dname}"] = bits(xd)
            g6[f"q_{C}_{dname}"] = bits(ops._no_sparsity_float_to_bfp(xd, 64, 3, 1e-8, 'determ', 'cpu'))
            g6[f"nm_{C}_{dname}"] = bits(ops._structured_N_M_sparsity(xd, 'cpu', 2, 4))
            for first in ('s', 'q'):
                c = cfg(first=first, w_sparsity=True)
                g6[f"comp_{first}_{C}_{dname}"] = bits(ops.float_to_bfp_blocked(xd, **c, identifier='w'))
                c = cfg(first=first, w_sparsity=True, N=3, M=8, block_size=16, mant_bits=7)
                g6[f"comp38_{first}_{C}_{dname}"] = bits(ops.float_to_bfp_blocked(xd, **c, identifier='w'))
    np.savez_compressed(os.path.join(HERE, "g6_padding.npz"), **g6)

    # ---- G7: edge cases ---------------------------------------------------------------------
    g7 = {}
    for dname, dt in DT.items():
        rows = []
        rows.append(torch.zeros(16))                                             # all-zero block
        for k in (-20, -9, -6, -3, -2, 0, 1, 4, 8, 10):
            r = torch.linspace(-1, 1, 16) * (2.0 ** k) * 0.3
            r[3] = 2.0 ** k                                                      # max exactly 2^k
            rows.append(r.clone())
            ulp = 2.0 ** (k - (23 if dt == torch.float32 else 10 if dt == torch.float16 else 7))
            r[3] = 2.0 ** k + ulp                                                # 2^k (1 + ulp)
            rows.append(r.clone())
            r[3] = 2.0 ** k + 2 * ulp
            rows.append(r.clone())
            r[3] = -(2.0 ** k) * 1.999                                           # saturation
            rows.append(r.clone())
        r = torch.tensor([0.5, 1.5, 2.5, 3.5, -0.5, -1.5, -2.5, -3.5, 0.25, -0.25, 7.0, -7.0, 6.5, -6.5, 0.0, -0.0])
        rows.append(r)                                                           # half-way cases, -0.0
        r = torch.linspace(-1, 1, 16); r[5] = float('inf'); rows.append(r)       # inf element
        r = torch.linspace(-1, 1, 16); r[5] = float('nan'); rows.append(r)       # nan element
        r = torch.full((16,), 1e-30); rows.append(r)                             # tiny
        r = torch.linspace(-1, 1, 16) * 3e38; rows.append(r)                     # huge (inf in fp16)
        r = torch.linspace(-1, 1, 16) * 6e-8; rows.append(r)                     # fp16 subnormals
        r = torch.linspace(-1, 1, 16) * 60000; rows.append(r)                    # near fp16 max
        x = torch.stack(rows).to(dt)
        g7[f"in_{dname}"] = bits(x)
        for m in (3, 7, 15):
            g7[f"out_{dname}_m{m}"] = bits(ops._no_sparsity_float_to_bfp(x, 16, m, 1e-8, 'determ', 'cpu'))
        g7[f"nm_{dname}"] = bits(ops._structured_N_M_sparsity(x, 'cpu', 2, 4))
    np.savez_compressed(os.path.join(HERE, "g7_edges.npz"), **g7)

    # ---- G8: N-d inputs, transpose path, BFPLinear fwd/bwd ----------------------------------
    g8 = {}
    act = torch.randn(2, 7, 128, generator=gen)
    convw = torch.randn(8, 3, 16, 16, generator=gen) * 0.1
    a4 = torch.randn(2, 4, 16, 32, generator=gen)
    b4 = torch.randn(2, 4, 32, 16, generator=gen)
    for dname, dt in DT.items():
        c = cfg(mant_bits=7, block_size=16, N=1, M=4, in_sparsity=True, w_sparsity=True)
        g8[f"act_in_{dname}"] = bits(act.to(dt))
        g8[f"act_out_{dname}"] = bits(ops.float_to_bfp_blocked(act.to(dt), **c, identifier='in'))
        g8[f"conv_in_{dname}"] = bits(convw.to(dt))
        g8[f"conv_out_{dname}"] = bits(ops.float_to_bfp_blocked(convw.to(dt), **c, identifier='w'))
        xa, xb = ops.MxM_pre_processing(a4.to(dt), b4.to(dt), True, **c)
        g8[f"mm_a_in_{dname}"] = bits(a4.to(dt)); g8[f"mm_b_in_{dname}"] = bits(b4.to(dt))
        g8[f"mm_a_out_{dname}"] = bits(xa); g8[f"mm_b_out_{dname}"] = bits(xb.contiguous())
    # BFPLinear forward/backward, fp32 and bf16 (captured from the reference module class itself)
    for dname in ("f32", "bf16"):
        dt = DT[dname]
        kw = cfg(mant_bits=7, block_size=32, N=2, M=4, w_sparsity=True, sparsity_mode='structured')
        lin = ops.BFPLinear(64, 128, True, **dict(kw))
        with torch.no_grad():
            lin.weight.copy_(torch.randn(128, 64, generator=gen) * 0.05)
            lin.bias.copy_(torch.randn(128, generator=gen) * 0.01)
        lin = lin.to(dt)
        x = torch.randn(2, 5, 64, generator=gen).to(dt).requires_grad_(True)
        y = lin(x)
        gy = torch.randn(2, 5, 128, generator=gen).to(dt)
        y.backward(gy)
        g8[f"lin_w_{dname}"] = bits(lin.weight.detach()); g8[f"lin_b_{dname}"] = bits(lin.bias.detach())
        g8[f"lin_x_{dname}"] = bits(x.detach()); g8[f"lin_y_{dname}"] = bits(y.detach())
        g8[f"lin_gy_{dname}"] = bits(gy)
        g8[f"lin_gx_{dname}"] = bits(x.grad); g8[f"lin_gw_{dname}"] = bits(lin.weight.grad); g8[f"lin_gb_{dname}"] = bits(lin.bias.grad)
        # the three quantized operands the module computes internally (pins the call order)
        g8[f"lin_xq_{dname}"] = bits(ops.float_to_bfp_blocked(x.detach(), **kw, identifier='in'))
        g8[f"lin_wq_{dname}"] = bits(ops.float_to_bfp_blocked(lin.weight.detach(), **kw, identifier='w'))
        g8[f"lin_gq_{dname}"] = bits(ops.float_to_bfp_blocked(gy, **kw, identifier='grad'))
    np.savez_compressed(os.path.join(HERE, "g8_nd_linear.npz"), **g8)

    write_g9(ops)
    write_g10()
    write_g11(ops)
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print("golden fixtures written, total bytes:", tot, "torch", torch.__version__)


def write_g9(ops):
    """G9: the 'int' per-channel format (bfp_ops.py:111-120 -> int_ops.Quantizer), weights and activations,
    2-D / 3-D / 4-D, alone and composed with sparsity in both orders"""
    g9 = {}
    gen = torch.Generator().manual_seed(99)
    shapes = {"w2": ((48, 200), 'w'), "a2": ((33, 96), 'in'), "a3": ((2, 7, 96), 'in'), "w4": ((8, 3, 5, 5), 'w'),
              "a4": ((2, 6, 4, 4), 'in'), "g2": ((16, 64), 'grad')}
    for name, (shape, ident) in shapes.items():
        base = torch.randn(*shape, generator=gen) * 0.05
        if name == "a2":
            base[:, 3] = 0.0                     # an all-zero channel (xmin = xmax = 0 -> [-1, 1])
            base[:, 5] = base[:, 5].abs()        # an all-positive channel (xmin stays 0)
        for dname, dt in DT.items():
            x = base.to(dt)
            g9[f"in_{name}_{dname}"] = bits(x)
            for nbits in (8, 4):
                c = cfg(sparsity_num_format='int', mant_bits=nbits, block_size=32)
                y = ops.float_to_bfp_blocked(x, **c, identifier=ident)
                assert y.dtype == torch.float32
                g9[f"out_{name}_{dname}_b{nbits}"] = bits(y)
            if name in ("w2", "a3"):
                flag = 'w_sparsity' if ident == 'w' else 'in_sparsity'
                for first in ('s', 'q'):
                    for mode, extra in (("structured", dict(N=2, M=4)), ("unstructured", dict(sparsity_frac=0.5))):
                        c = cfg(sparsity_num_format='int', mant_bits=8, block_size=32, first=first, sparsity_mode=mode, **{flag: True}, **extra)
                        y = ops.float_to_bfp_blocked(x, **c, identifier=ident)
                        g9[f"comp_{name}_{dname}_{first}_{mode[:1]}"] = bits(y)
    np.savez_compressed(os.path.join(HERE, "g9_int.npz"), **g9)


def write_g10():
    """G10: the reference's BFPAdam (bfp_optim_lstm.py:12-93) for three steps on CPU, 'determ' rounding,
    weight_mant_bits = 15 (the wide-mantissa weight grid of HBFP training)"""
    load_ref()
    import importlib.util
    spec = importlib.util.spec_from_file_location("refbfp.bfp_util", f"{REF}/bfp_util.py")
    util = importlib.util.module_from_spec(spec); sys.modules["refbfp.bfp_util"] = util; spec.loader.exec_module(util)
    spec = importlib.util.spec_from_file_location("refbfp.bfp_optim_lstm", f"{REF}/bfp_optim_lstm.py")
    optm = importlib.util.module_from_spec(spec); sys.modules["refbfp.bfp_optim_lstm"] = optm; spec.loader.exec_module(optm)
    g10 = {}
    gen = torch.Generator().manual_seed(2025)
    for tag, amsgrad in (("plain", False), ("ams", True)):
        params = [torch.nn.Parameter(torch.randn(64, 128, generator=gen) * 0.05), torch.nn.Parameter(torch.randn(96, generator=gen) * 0.05)]
        opt = optm.BFPAdam(params, lr=1e-2, amsgrad=amsgrad)
        opt.bfp_args = cfg(mant_bits=7, weight_mant_bits=15, block_size=32)          # the YAML on disk says device 'cuda' / 'stoc'
        # the parameters as they enter the snap (bfp_optim_lstm.py:85: the argument of float_to_bfp_blocked), recorded by a
        # pass-through wrapper around the reference's own function: lets a test pin the snap alone, bit for bit
        presnap = []
        real_q = optm.float_to_bfp_blocked

        def recording_q(t, *a, **kw):
            presnap.append(t.detach().clone())
            return real_q(t, *a, **kw)
        optm.float_to_bfp_blocked = recording_q
        for i, p_ in enumerate(params):
            g10[f"{tag}_p{i}_init"] = bits(p_.data)
        for step in range(3):
            for i, p_ in enumerate(params):
                gr = torch.randn(p_.shape, generator=gen) * 0.1
                g10[f"{tag}_g{i}_s{step}"] = bits(gr)
                p_.grad = gr
            del presnap[:]
            opt.step()
            assert len(presnap) == len(params)
            for i, p_ in enumerate(params):
                g10[f"{tag}_p{i}_s{step}"] = bits(p_.data)
                g10[f"{tag}_pre{i}_s{step}"] = bits(presnap[i])
        optm.float_to_bfp_blocked = real_q
    np.savez_compressed(os.path.join(HERE, "g10_bfpadam.npz"), **g10)


def write_g11(ops):
    """G11: the module / functional wrappers that the patched models call (bfp_ops.py:233-268), forward AND backward, captured
    from the reference's own classes: BFPConv2d in 'bfp' mode (ViT's patch embedding, modeling_vit.py:168-173), the callable
    of F_matmul_bfp (transpose path inside an autograd graph) and the callable of F_linear_bfp"""
    g11 = {}
    gen = torch.Generator().manual_seed(4711)
    kw = cfg(mant_bits=7, block_size=16, N=1, M=4, w_sparsity=True, sparsity_mode='structured')     # BASELINE config 5's numerics
    for dname in ("f32", "bf16"):
        dt = DT[dname]
        # (a) BFPConv2d(3, 64, 16, stride=16) on [2,3,32,32]
        conv = ops.BFPConv2d(3, 64, 16, stride=16, **dict(kw))
        with torch.no_grad():
            conv.weight.copy_(torch.randn(64, 3, 16, 16, generator=gen) * 0.05)
            conv.bias.copy_(torch.randn(64, generator=gen) * 0.01)
        conv = conv.to(dt)
        x = torch.randn(2, 3, 32, 32, generator=gen).to(dt).requires_grad_(True)
        y = conv(x)
        gy = torch.randn(y.shape, generator=gen).to(dt)
        y.backward(gy)
        for name, t in (("w", conv.weight.detach()), ("b", conv.bias.detach()), ("x", x.detach()), ("y", y.detach()), ("gy", gy),
                        ("gx", x.grad), ("gw", conv.weight.grad), ("gb", conv.bias.grad),
                        ("xq", ops.float_to_bfp_blocked(x.detach(), **kw, identifier='in')),
                        ("wq", ops.float_to_bfp_blocked(conv.weight.detach(), **kw, identifier='w')),
                        ("gq", ops.float_to_bfp_blocked(gy, **kw, identifier='grad'))):
            g11[f"conv_{name}_{dname}"] = bits(t)
        # (b) the callable of F_matmul_bfp on [2,4,16,32] x [2,4,32,16]
        mm = ops.F_matmul_bfp(**dict(kw))
        a = torch.randn(2, 4, 16, 32, generator=gen).to(dt).requires_grad_(True)
        b = torch.randn(2, 4, 32, 16, generator=gen).to(dt).requires_grad_(True)
        y = mm(a, b)
        gy = torch.randn(y.shape, generator=gen).to(dt)
        y.backward(gy)
        aq, bq = ops.MxM_pre_processing(a.detach(), b.detach(), True, **kw)
        for name, t in (("a", a.detach()), ("b", b.detach()), ("y", y.detach()), ("gy", gy), ("ga", a.grad), ("gb", b.grad),
                        ("aq", aq), ("bq", bq.contiguous()), ("gq", ops.float_to_bfp_blocked(gy, **kw, identifier='grad'))):
            g11[f"mm_{name}_{dname}"] = bits(t)
        # (c) the callable of F_linear_bfp
        fl = ops.F_linear_bfp(**dict(kw))
        x = torch.randn(2, 5, 64, generator=gen).to(dt).requires_grad_(True)
        w = (torch.randn(48, 64, generator=gen) * 0.05).to(dt).requires_grad_(True)
        bias = (torch.randn(48, generator=gen) * 0.01).to(dt).requires_grad_(True)
        y = fl(x, w, bias)
        gy = torch.randn(y.shape, generator=gen).to(dt)
        y.backward(gy)
        for name, t in (("x", x.detach()), ("w", w.detach()), ("b", bias.detach()), ("y", y.detach()), ("gy", gy),
                        ("gx", x.grad), ("gw", w.grad), ("gb", bias.grad)):
            g11[f"lin_{name}_{dname}"] = bits(t)
    # the non-'bfp' format returns the plain functions themselves (bfp_ops.py:237-238, :244-245)
    assert ops.F_linear_bfp(num_format='fp32') is torch.nn.functional.linear and ops.F_matmul_bfp(num_format='fp32') is torch.matmul
    np.savez_compressed(os.path.join(HERE, "g11_wrappers.npz"), **g11)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g11":
        write_g11(load_ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "g9":
        write_g9(load_ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "g10":
        write_g10()
    else:
        main()
