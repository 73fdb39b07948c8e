"""CPU-only tests of the host side: the C-ABI library loads and exports what include/bfpq.h declares,
the host-built tables reproduce the reference (golden G1 / G3) and the oracle, config plumbing mirrors
the reference's dict semantics, and the product refuses CPU tensors (no fallback)."""
import os
import re

import numpy as np
import pytest
import torch

import quantization_sparsity_interplay_amd as pkg
from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops, bfp_util
from util import DT, load, from_bits
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = pkg.load_library()
    hdr = open(os.path.join(ROOT, "include", "bfpq.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bfpq_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(native.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.bfpq_version() == 4
    assert b"invalid" in L.bfpq_error_string(-1)


def _e_from_window(sbits_f32, table, mbits):
    """python restatement of block_scale()'s exponent step, for checking the table"""
    sbits_f32 = sbits_f32.astype(np.int64)
    k = (sbits_f32 >> 23) - 127
    mant = sbits_f32 & 0x7FFFFF
    sub = (sbits_f32 >> 23) == 0                                  # fp32 subnormal: normalise like the kernel does
    if sub.any():
        lz = 22 - np.floor(np.log2(mant[sub])).astype(np.int64)   # leading zeros inside the 23-bit field
        mant[sub] = (mant[sub] << (lz + 1)) & 0x7FFFFF
        k[sub] = -127 - lz
    mant = mant >> (23 - mbits)
    win = np.frombuffer(table, dtype=np.uint8)[k + 160]
    return k + (mant > win)


@pytest.mark.parametrize("dname", ["bf16", "f16"])
def test_exp_window_matches_reference_g1(dname):
    """every finite positive 16-bit pattern: window table + in-kernel arithmetic == reference get_exponent"""
    dt = DT[dname]
    g = load("g1_exponent.npz")
    want = g[f"{dname}_e"]
    hi = 0x7F80 if dname == "bf16" else 0x7C00
    x = from_bits(np.arange(0, hi, dtype=np.uint16), dt)
    eps_dt = torch.tensor(1e-8, dtype=torch.float32).to(dt).float()
    s = (x.float() + eps_dt).to(dt).float()                      # max + eps, rounded to dtype
    sb = s.view(torch.int32).numpy().astype(np.int64)
    table = native.exp_window_host(dt)
    ok = sb > 0
    got = _e_from_window(sb[ok], table, 7 if dname == "bf16" else 10).astype(np.float32)
    assert np.array_equal(got, want[ok])
    assert np.all(np.isinf(want[~ok]))                           # fp16 zero: log2(0) = -inf -> NaN block


def test_exp_window_matches_reference_g1_f32():
    g = load("g1_exponent.npz")
    bits_in = g["f32_bits"].astype(np.uint32)
    x = from_bits(bits_in, torch.float32)
    s = x + torch.tensor(1e-8, dtype=torch.float32)
    sb = s.view(torch.int32).numpy().astype(np.int64)
    norm = (sb >> 23) > 0
    got = _e_from_window(sb[norm], native.exp_window_host(torch.float32), 23).astype(np.float32)
    assert np.array_equal(got, g["f32_e"][norm])


def test_exp_window_matches_local_torch():
    """the same table against THIS machine's torch CPU log2 (the box's torch may differ from the build container's)"""
    for dname, mbits in (("bf16", 7), ("f16", 10)):
        dt = DT[dname]
        hi = 0x7F80 if dname == "bf16" else 0x7C00
        x = from_bits(np.arange(1, hi, dtype=np.uint16), dt)
        want = x.log2().ceil().float().numpy()
        got = _e_from_window(x.float().view(torch.int32).numpy().astype(np.int64), native.exp_window_host(dt), mbits)
        assert np.array_equal(got.astype(np.float32), want), dname


def test_nm4_lut_matches_reference_g3():
    g = load("g3_nm.npz")
    rows = g["m4_rows"].astype(np.int64)

    def c3(a, b):
        return (a > b).astype(np.int64) + (a >= b).astype(np.int64)
    idx = (c3(rows[:, 0], rows[:, 1]) + 3 * c3(rows[:, 0], rows[:, 2]) + 9 * c3(rows[:, 0], rows[:, 3])
           + 27 * c3(rows[:, 1], rows[:, 2]) + 81 * c3(rows[:, 1], rows[:, 3]) + 243 * c3(rows[:, 2], rows[:, 3]))
    for N in (1, 2, 3):
        lut = np.frombuffer(native.nm4_lut_host(N), dtype=np.uint8)
        keep = lut[idx]
        want = g[f"m4_keep_N{N}"]
        got = np.stack([(keep >> j) & 1 for j in range(4)], axis=1).astype(np.uint8)
        assert np.array_equal(got, want), N


@pytest.mark.parametrize("NM", [(2, 8), (4, 8), (1, 8), (7, 8), (4, 16), (8, 16), (2, 16), (16, 32), (8, 32), (1, 2), (3, 6), (2, 5)])
def test_nm_prune_mask_host_matches_reference_g3(NM):
    """the host/device introselect replay (csrc/nm_select.h) against the reference's masks"""
    N, M = NM
    g = load("g3_nm.npz")
    rows = g[f"rows_{N}_{M}"]
    want = np.unpackbits(g[f"keep_{N}_{M}"], axis=1)[:, :M]
    for r in range(0, rows.shape[0], 5):
        prune = native.nm_prune_mask_host(rows[r].astype(np.uint32) + 1, N, M)
        got = np.array([0 if (prune >> i) & 1 else 1 for i in range(M)], dtype=np.uint8)
        assert np.array_equal(got, want[r]), (N, M, r)


@pytest.mark.parametrize("M", [16, 32, 64])
def test_nm_prune_mask_host_killer_rows(M):
    g = load("g3_nm.npz")
    rows = g[f"killer_rows_{M}"]
    want = np.unpackbits(g[f"killer_keep_{M}"], axis=1)[:, :M]
    for r in range(rows.shape[0]):
        prune = native.nm_prune_mask_host(rows[r].astype(np.uint32) + 1, M // 2, M)
        got = np.array([0 if (prune >> i) & 1 else 1 for i in range(M)], dtype=np.uint8)
        assert np.array_equal(got, want[r]), (M, r)


def test_nm_prune_mask_host_vs_oracle_random():
    rng = np.random.RandomState(3)
    for M in (3, 4, 7, 12, 33, 64):
        for N in (1, max(1, M // 2), M - 1):
            for _ in range(200):
                keys = rng.randint(0, 5, size=M).astype(np.uint32)
                want = O.topk_smallest_mask(keys.astype(np.float32), M - N)
                prune = native.nm_prune_mask_host(keys, N, M)
                got = np.array([(prune >> i) & 1 for i in range(M)], dtype=np.uint8)
                assert np.array_equal(got, want), (N, M, keys)


def test_nm8_rank_table_matches_the_replay():
    """the 16 MiB N:8 table (index = 3-bit 'number of smaller keys' per element) against the host nth_element replay"""
    import random
    rnd = random.Random(7)
    for N in (1, 4, 6):
        lut = np.frombuffer(native.nm8_lut_host(N), dtype=np.uint8)
        assert int((lut != 0).sum()) == 545835                               # one entry per weak ordering of 8 elements
        for _ in range(400):
            keys = [rnd.choice([0, 1, 2, 3, 5, 9, 0x7f80]) for _ in range(8)]
            less = [sum(1 for j in range(8) if keys[j] < keys[i]) for i in range(8)]
            idx = sum(l << (3 * i) for i, l in enumerate(less))
            assert int(lut[idx]) == native.nm_prune_mask_host(keys, N, 8), (N, keys)


def test_is_fused_dispatch():
    L = pkg.load_library()
    assert L.bfpq_is_fused(4096, 11008, native.BF16, 64, 2, 4) == 1        # headline
    assert L.bfpq_is_fused(768, 768, native.F32, 32, 0, 0) == 1            # cfg 1
    assert L.bfpq_is_fused(4096, 1024, native.F32, 16, 1, 4) == 1          # cfg 5
    assert L.bfpq_is_fused(5, 100, native.BF16, 64, 2, 4) == 0             # ragged rows
    assert L.bfpq_is_fused(64, 256, native.BF16, 64, 4, 8) == 1            # N:8 on a 16-bit dtype: one lane item per group
    assert L.bfpq_is_fused(64, 256, native.F32, 64, 4, 8) == 1             # fp32: a group of 8 = two adjacent lane items (DPP exchange)
    assert L.bfpq_is_fused(64, 256, native.BF16, 64, 4, 16) == 0           # M = 16 -> general path
    assert L.bfpq_is_fused(64, 256, native.BF16, 48, 0, 0) == 0            # 6 lanes per block


def test_unpack_bfp_args_semantics():
    kw = dict(num_format='bfp', mant_bits=7, block_size=32, bfp_tile_size=8, unconstrained=True)
    args = bfp_ops.unpack_bfp_args(kw)
    assert kw == dict(bfp_tile_size=8, unconstrained=True)          # known keys popped, unknown left (bfp_ops.py:225-230)
    assert list(args.keys()) == ['num_format', 'sparsity_num_format', 'rounding_mode', 'epsilon', 'mant_bits', 'block_size',
                                 'weight_mant_bits', 'in_sparsity', 'w_sparsity', 'grad_sparsity', 'N', 'M', 'first',
                                 'sparsity_mode', 'sparsity_frac', 'mx_w_elem_format', 'mx_a_elem_format', 'bfloat',
                                 'scale_bits', 'device']
    assert args['rounding_mode'] == 'stoc' and args['epsilon'] == 1e-8 and args['device'] == 'cpu' and args['first'] == 's'
    assert args['sparsity_mode'] == 'unstructured' and args['bfloat'] == 16 and args['scale_bits'] == 8


def test_bfpconfig_roundtrip(tmp_path):
    c = pkg.BFPConfig.hbfp(4, 64, w_sparsity=True, N=2, M=4, sparsity_mode='structured')
    assert c.mant_bits == 3 and c.block_size == 64
    kw = c.to_kwargs()
    assert set(kw) == set(bfp_ops.unpack_bfp_args({}).keys())
    p = tmp_path / "c.yaml"
    c.to_yaml(p)
    assert pkg.BFPConfig.from_yaml(p) == c
    assert pkg.BFPConfig.from_dict(dict(kw, bfp_tile_size=3)) == c   # unknown keys tolerated
    d = bfp_util.get_bfp_args()
    # the shipped default file carries the reference's 21 keys AND values (src/transformers/bfp/bfp_config.yaml:1-21)
    assert d == dict(num_format='bfp', sparsity_num_format='fp32', rounding_mode='stoc', epsilon=1e-8, mant_bits=7,
                     weight_mant_bits=15, block_size=32, in_sparsity=False, w_sparsity=False, grad_sparsity=False,
                     sparsity_frac=0.5, N=2, M=4, first='s', sparsity_mode='structured', mx_w_elem_format='fp8_e4m3',
                     mx_a_elem_format='fp8_e4m3', bfloat=16, scale_bits=8, device='cuda')
    assert set(d) <= set(pkg.BFPConfig.keys())
    assert bfp_util.extract_sparsity_args(d) == dict(sparsity=False, device='cuda', sparsity_mode='structured', sparsity_frac=0.5, N=2, M=4)
    assert bfp_util.extract_mx_args(d)['block_size'] == 32
    import os
    h = bfp_util.get_bfp_args(os.path.join(os.path.dirname(bfp_util.__file__), 'bfp_config_headline.yaml'))   # the BASELINE.json workload
    assert (h['mant_bits'], h['block_size'], h['w_sparsity'], h['N'], h['M'], h['rounding_mode'], h['sparsity_num_format']) == \
        (3, 64, True, 2, 4, 'determ', 'bfp')


def test_modules_keep_stock_state_dict_and_identity_cases():
    lin = bfp_ops.BFPLinear(8, 4, True, **pkg.BFPConfig.hbfp(8, 32).to_kwargs())
    assert sorted(lin.state_dict()) == ['bias', 'weight'] and lin.num_format == 'bfp' and callable(lin.linear_op)
    conv = bfp_ops.BFPConv2d(3, 4, 3, **pkg.BFPConfig().to_kwargs())
    assert sorted(conv.state_dict()) == ['bias', 'weight'] and conv.num_format == 'fp32'
    x = torch.randn(2, 3, 8, 8)
    assert conv(x).shape == (2, 4, 6, 6)                              # 'fp32' format never touches the engine
    assert bfp_ops.F_linear_bfp() is torch.nn.functional.linear and bfp_ops.F_matmul_bfp() is torch.matmul
    t = torch.randn(4, 8)
    assert bfp_ops._sparsify(t, False, 'structured', 'cpu', 2, 4, 0.5) is t              # bfp_ops.py:102
    assert bfp_ops._quantize(t, 'fp32', 0, 0, 0, False, 1e-8, 'determ', 'cpu', 'w') is t  # bfp_ops.py:106
    with pytest.raises(ValueError):
        bfp_ops._sparsify(t, True, 'banded', 'cpu', 2, 4, 0.5)
    with pytest.raises(ValueError):
        bfp_ops._quantize(t, 'fp8', 0, 0, 0, False, 1e-8, 'determ', 'cpu', 'w')
    with pytest.raises(AssertionError):
        bfp_ops._structured_N_M_sparsity(t, 'cpu', 5, 4)
    with pytest.raises(AssertionError):
        bfp_ops._unstructured_sparsity(t, 'cpu', 0)
    with pytest.raises(AssertionError):
        bfp_ops.float_to_bfp_blocked(t, **dict(pkg.BFPConfig.hbfp(8, 32).to_kwargs(), num_format='fp32'))


def test_no_cpu_fallback():
    t = torch.randn(4, 64)
    cfg = pkg.BFPConfig.hbfp(4, 64).to_kwargs()
    with pytest.raises(pkg.NativeUnavailable):
        bfp_ops.float_to_bfp_blocked(t, **cfg, identifier='w')
    with pytest.raises(pkg.NativeUnavailable):
        bfp_ops._structured_N_M_sparsity(t, 'cpu', 2, 4)
    with pytest.raises(pkg.NativeUnavailable):
        bfp_ops._unstructured_sparsity(t, 'cpu', 0.5)


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, "quantization-sparsity-interplay_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower().replace("# oracle", ""), os.path.join(dirpath, f)


def test_selection_histogram_swizzle_invariants():
    """what k_select_hist relies on (csrc/bfpq_unstructured.hip, swz15): the LDS index map of the 15-bit histogram is a bijection that keeps every
    aligned group of 32 bins in place (so sums over coarse bins of 128 do not see it), maps 0 to 0 only (the zero test runs on the swizzled key),
    the packed form mixes both 16-bit halves of a dword exactly like the scalar one, and a coarse-keyed bin's 32 replica words
    swz15(b) ^ lane are exactly its own group"""
    import numpy as np
    b = np.arange(1 << 15, dtype=np.uint32)
    swz = lambda x: x ^ (((x >> 5) ^ (x >> 10)) & 31)
    s = swz(b)
    assert np.array_equal(np.sort(s), b)                                   # bijection
    assert np.array_equal(s >> 5, b >> 5)                                  # every aligned group of 32 stays in place
    assert int((s == 0).sum()) == 1 and s[0] == 0
    rng = np.random.default_rng(5)
    lo, hi = rng.integers(0, 1 << 15, 100000, dtype=np.uint32), rng.integers(0, 1 << 15, 100000, dtype=np.uint32)
    k2 = lo | (hi << 16)
    s2 = k2 ^ (((k2 >> 5) ^ (k2 >> 10)) & 0x001F001F)
    assert np.array_equal(s2 & 0xFFFF, swz(lo)) and np.array_equal(s2 >> 16, swz(hi))
    coarse = b[(b & 31) == 0]
    for lane in range(64):
        assert np.array_equal((swz(coarse) ^ (lane & 31)) >> 5, coarse >> 5)
    assert all(len({int(swz(c) ^ (lane & 31)) for lane in range(32)}) == 32 for c in coarse[:64])
