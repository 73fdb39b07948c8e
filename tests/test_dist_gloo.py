"""world_size-2 `gloo` tests (CPU) of the row-sharded path in quantization-sparsity-interplay_amd/dist.py.
The product has no CPU compute path, so the per-rank math is injected: the oracle for quantize / N:M, and a
small numpy stand-in of the engine's select/apply protocol (same callbacks, same histogram layout) for the
unstructured exchange.  What is under test is the distributed logic: slab ownership, ragged gathers, the
histogram all-gather (which also carries the tie counts of lower ranks), and "gathered == single-process"."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class NumpyEngine:
    """stand-in for native.select_threshold / threshold_apply on CPU bf16/fp16 tensors (one 15-bit radix pass; same
    histogram layout and the same all-gather callback as the HIP engine).  It does NOT model fp32's three radix passes
    (11 + 11 + 9 bits, a histogram all-gather per pass, the prefix carried between them): those run with the real kernels
    and 2 / 3 ranks in tests/test_dist_gpu.py (fp32 tensors, real-valued and tie-heavy)."""

    class WS:
        pass

    def workspace(self, device):
        return NumpyEngine.WS()

    @staticmethod
    def _keys(t):
        b = t.contiguous().view(torch.int16).numpy().view(np.uint16).reshape(-1).astype(np.int64) & 0x7FFF
        inf = 0x7F80 if t.dtype == torch.bfloat16 else 0x7C00
        return np.minimum(b, inf + 1)

    def select_threshold(self, t, k, ws, numel_global=None, allgather=None):
        fine = np.bincount(self._keys(t), minlength=32768).astype(np.int32) if t.numel() else np.zeros(32768, np.int32)
        local = torch.from_numpy(np.concatenate([fine, fine.reshape(256, 128).sum(1).astype(np.int32)]))   # fine + coarse bins
        allh, R, rank = allgather(local) if allgather is not None else (local.view(1, -1), 1, 0)
        allh = allh.numpy().astype(np.int64)[:, :32768]
        hist = allh.sum(0)
        c = np.cumsum(hist)
        if k == 0:
            ws.tau, ws.need, ws.ties, ws.k, ws.base = 0, 0, int(hist[0]), 0, 0
            return
        tau = int(np.searchsorted(c, k, side="left"))
        ws.tau, ws.k = tau, k
        ws.need = int(k - (c[tau - 1] if tau > 0 else 0))
        ws.ties = int(hist[tau])
        ws.base = int(allh[:rank, tau].sum())              # ties held by lower ranks: read off the gathered histograms

    def threshold_apply(self, t, ws, out=None):
        keys = self._keys(t)
        eq = keys == ws.tau
        rank = ws.base + np.cumsum(eq) - 1
        prune = (ws.k > 0) & ((keys < ws.tau) | (eq & (rank < ws.need)))
        flat = t.contiguous().view(-1).clone()
        flat[torch.from_numpy(prune)] = 0
        return flat.view(t.shape)


def _cfg(**kw):
    base = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cpu', block_size=64,
                num_format='bfp', weight_mant_bits=15, in_sparsity=False, w_sparsity=True,
                grad_sparsity=False, sparsity_frac=0.5, N=2, M=4, sparsity_num_format='bfp',
                first='s', sparsity_mode='structured')
    base.update(kw)
    return base


def _worker(rank, world, port, tmpdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quantization_sparsity_interplay_amd import dist as qd
        from oracle import oracle as O
        g = torch.Generator().manual_seed(1234)
        results = {}
        for rows in (64, 37):                                       # even split and ragged split
            full = (torch.randn(rows, 256, generator=g) * 0.02).to(torch.bfloat16)
            local = qd.shard_rows(full, world, rank)
            lo, hi = qd.row_range(rows, world, rank)
            assert local.shape[0] == hi - lo
            # structured / dense: no collective in the data path, gather only on request
            for first in ('s', 'q'):
                c = _cfg(first=first)
                out = qd.float_to_bfp_blocked_sharded(local, rows, gather=True, compute=O.float_to_bfp_blocked, identifier='w', **c)
                want = O.float_to_bfp_blocked(full, **c, identifier='w')
                assert torch.equal(out.view(torch.int16), want.view(torch.int16)), ("structured", rows, first)
                loc = qd.float_to_bfp_blocked_sharded(local, rows, gather=False, compute=O.float_to_bfp_blocked, identifier='w', **c)
                assert torch.equal(loc.view(torch.int16), want[lo:hi].view(torch.int16))
            # unstructured: one global threshold, ties lowest-global-index first == the single-process stand-in
            eng = NumpyEngine()
            for first in ('s', 'q'):
                c = _cfg(first=first, sparsity_mode='unstructured', sparsity_frac=0.5)
                out = qd.float_to_bfp_blocked_sharded(local, rows, gather=True, compute=O.float_to_bfp_blocked, engine=eng, identifier='w', **c)
                dense = dict(c, w_sparsity=False)
                ws = eng.workspace(None)
                if first == 's':
                    eng.select_threshold(full, int(full.numel() * 0.5), ws)
                    single = O.float_to_bfp_blocked(eng.threshold_apply(full, ws), **dense, identifier='w')
                else:
                    q = O.float_to_bfp_blocked(full, **dense, identifier='w')
                    eng.select_threshold(q, int(q.numel() * 0.5), ws)
                    single = eng.threshold_apply(q, ws)
                assert torch.equal(out.view(torch.int16), single.view(torch.int16)), ("unstructured", rows, first)
                if first == 's':                                    # and the same count / threshold as the reference algorithm
                    ref = O.unstructured_sparsity(full, 0.5)
                    pruned = qd.unstructured_sparsity_sharded(local, 0.5, full.numel(), engine=eng)
                    allp = qd.all_gather_rows(pruned, rows)
                    assert int((allp == 0).sum()) == int((ref == 0).sum())
                    assert torch.equal(allp.float().abs().min(dim=1)[0] >= 0, torch.ones(rows, dtype=torch.bool))
            results[rows] = True
        # the list form (on CPU: tensor after tensor, same collectives in the same order on every rank)
        fulls = [(torch.randn(r, 256, generator=g) * 0.02).to(torch.bfloat16) for r in (64, 37, 5)]
        slabs = [qd.shard_rows(f, world, rank) for f in fulls]
        cu = _cfg(sparsity_mode='unstructured', sparsity_frac=0.5)
        many = qd.float_to_bfp_blocked_many_sharded(slabs, [f.shape[0] for f in fulls], gather=True, compute=O.float_to_bfp_blocked, engine=NumpyEngine(),
                                                    identifier='w', **cu)
        for f, m in zip(fulls, many):
            one = qd.float_to_bfp_blocked_sharded(qd.shard_rows(f, world, rank), f.shape[0], gather=True, compute=O.float_to_bfp_blocked, engine=NumpyEngine(),
                                                  identifier='w', **cu)
            assert torch.equal(m.view(torch.int16), one.view(torch.int16))
        # chunked gather (overlap path; on CPU the chunks simply run in order)
        full = (torch.randn(64, 256, generator=g) * 0.02).to(torch.bfloat16)
        c = _cfg()
        got = qd.gather_overlapped(qd.shard_rows(full, world, rank), 64,
                                   lambda p: O.float_to_bfp_blocked(p, **c, identifier='w'), chunks=3)
        assert torch.equal(got.view(torch.int16), O.float_to_bfp_blocked(full, **c, identifier='w').view(torch.int16))
        # ragged all_gather of a different dtype (packed codes / exponents travel as uint8 / int8)
        codes = torch.arange(37 * 8, dtype=torch.uint8).view(37, 8)
        got = qd.all_gather_rows(qd.shard_rows(codes, world, rank), 37)
        assert torch.equal(got, codes)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_row_sharded_world2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_row_range_partition():
    from quantization_sparsity_interplay_amd import dist as qd
    for rows in (0, 1, 7, 8, 4096, 11008, 13824):
        for world in (1, 2, 4, 8):
            spans = [qd.row_range(rows, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == rows
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
