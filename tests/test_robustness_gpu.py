"""GPU tests of the caches and bound-pointer helpers around the engine (no counterpart in the reference, which caches nothing):
the activation-image memo under hipGraph capture and torch.inference_mode(), the weight cache with tensors that have no
version counter, PreparedList after a storage move, the select workspace born inside a capture."""
import pytest
import torch

from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops
from quantization_sparsity_interplay_amd.patch import PackedBFPLinear

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


def test_activation_image_memo_is_not_filled_under_capture():
    """an image "made" while a hipGraph is being captured is recorded, not computed: the memo must not keep it.  Capture a
    PackedBFP.linear call, do NOT replay, then call eagerly on the same input: the result must equal a fresh computation."""
    w = synth(256, 512, torch.bfloat16).to(DEV)
    x = synth(200, 512, torch.bfloat16, 1.0, seed=4).to(DEV)
    pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
    native.forget_shared_images()
    try:
        native.SHARE_ACT_IMAGE = False
        want = pw.linear(x, x_mant_bits=3)                           # reference result with the memo out of the picture
    finally:
        native.SHARE_ACT_IMAGE = True
    warm = pw.linear(x, x_mant_bits=3)                               # (allocator / library warm-up outside the capture)
    assert torch.equal(warm, want)
    native.forget_shared_images()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            captured = pw.linear(x, x_mant_bits=3)
        assert native._last_image.get(x.device) is None, "an entry was stored during capture"
        eager = pw.linear(x, x_mant_bits=3)                          # same tensor, same stream, never replayed
        side.synchronize()
        assert torch.equal(eager, want)
        graph.replay()
        side.synchronize()
        assert torch.equal(captured, want)
    torch.cuda.current_stream().wait_stream(side)


def test_packed_and_cached_modules_under_inference_mode():
    """tensors created under torch.inference_mode() have no version counter (`_version` raises): the memo and the weight cache
    must step aside instead of crashing -- PackedBFPLinear and the matrix-unit BFPLinear with >= 64 tokens"""
    c = cfg(w_sparsity=True)
    lin = bfp_ops.BFPLinear(512, 384, True, **dict(c)).to(DEV).to(torch.bfloat16).eval()
    with torch.no_grad():
        lin.weight.copy_(synth(384, 512, torch.bfloat16).to(DEV))
        lin.bias.copy_(synth(1, 384, torch.bfloat16, 1.0, seed=5).view(384).to(DEV))
        x0 = synth(128, 512, torch.bfloat16, 1.0, seed=11).to(DEV)
        ref = lin(x0)
    lin.enable_weight_cache(matrix_unit=True)
    packed = PackedBFPLinear.from_linear(lin, dict(c))
    with torch.inference_mode():
        x = x0.clone()                                               # an inference tensor
        assert x.is_inference()
        with pytest.raises(RuntimeError):
            x._version                                               # noqa: B018  (the premise of this test)
        got = lin(x)
        got2 = lin(x)
        assert torch.equal(got, got2)
        assert float((got.float() - ref.float()).abs().max() / ref.float().abs().max()) < 2e-2
        p1 = packed(x)
        p2 = packed(x * 1)                                           # another inference tensor
        assert torch.equal(p1, p2)
        assert float((p1.float() - ref.float()).abs().max() / ref.float().abs().max()) < 2e-2
        # weights created under inference mode: the cache is bypassed (no version counter to key on), results unchanged
        lin2 = bfp_ops.BFPLinear(512, 384, False, **dict(c)).to(DEV).to(torch.bfloat16).eval().enable_weight_cache()
        lin2.weight = torch.nn.Parameter(lin.weight.detach().clone(), requires_grad=False)
        assert lin2.weight.is_inference()
        a = lin2(x)
        lin2.weight.mul_(0.5)                                        # in-place update that no counter records
        b = lin2(x)
        assert not torch.equal(a, b), "a stale cached weight was used"
        assert lin2.linear_op.weight_cache.hits == 0


def test_prepared_list_follows_moved_storage():
    """PreparedList binds raw pointers; after `p.data = ...` (or model.to(...)) the old storage is gone: run() must follow the
    tensor, and refuse a changed dtype instead of reading freed memory"""
    c = cfg(w_sparsity=True)
    ps = [torch.nn.Parameter(synth(64, 256, torch.bfloat16, seed=i).to(DEV), requires_grad=False) for i in range(3)]
    prep = bfp_ops.PreparedMany(ps, identifier='w', **c)
    out1 = [o.clone() for o in prep.run()]
    for p, o in zip(ps, out1):
        assert torch.equal(o, bfp_ops.float_to_bfp_blocked(p, **c, identifier='w'))
    new = synth(64, 256, torch.bfloat16, seed=77).to(DEV)
    old_ptr = ps[1].data_ptr()
    ps[1].data = new                                                 # storage moves, the old one is freed
    assert ps[1].data_ptr() != old_ptr
    junk = torch.full((64, 256), 3.0, dtype=torch.bfloat16, device=DEV)   # likely lands in the freed block
    out2 = prep.run()
    assert torch.equal(out2[1], bfp_ops.float_to_bfp_blocked(new, **c, identifier='w'))
    assert torch.equal(out2[0], out1[0]) and torch.equal(out2[2], out1[2])
    del junk
    ps[2].data = ps[2].data.float()
    with pytest.raises(RuntimeError, match="changed dtype"):
        prep.run()


def test_select_workspace_and_graph_capture():
    """a stream first met inside a capture (torch.cuda.graph captures on its own stream) gets a workspace that was made and
    zeroed eagerly; an eager call on that stream BEFORE any replay works; without any eager call beforehand the capture is refused"""
    c = cfg(w_sparsity=True, sparsity_mode='unstructured')
    x = synth(64, 256, torch.bfloat16).to(DEV)
    want = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')               # eager: this stream's workspace + the spares
    assert len(bfp_ops._spare_ws[x.device.index]) == bfp_ops._SPARES
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        bfp_ops._select_ws.pop((x.device.index, side.cuda_stream), None)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            got = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
        ws = bfp_ops._select_ws[(x.device.index, side.cuda_stream)]
        assert ws.pinned
        eager = bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')          # same stream, before any replay
        side.synchronize()
        assert torch.equal(eager, want)
        graph.replay()
        side.synchronize()
        assert torch.equal(got, want)
        # no spares, unknown stream: refused with a hint instead of recording a zero-fill that never runs
        saved = bfp_ops._spare_ws.pop(x.device.index)
        other = torch.cuda.Stream()
        other.wait_stream(side)
        try:
            with torch.cuda.stream(other):
                g2 = torch.cuda.CUDAGraph()
                with pytest.raises(RuntimeError, match="outside"):
                    with torch.cuda.graph(g2, stream=other):
                        bfp_ops.float_to_bfp_blocked(x, **c, identifier='w')
        finally:
            bfp_ops._spare_ws[x.device.index] = saved
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # the bounded workspace table: many streams do not accumulate workspaces (pinned ones stay)
    streams = [torch.cuda.Stream() for _ in range(bfp_ops._SELECT_WS_MAX + 4)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            assert torch.equal(bfp_ops.float_to_bfp_blocked(x, **c, identifier='w'), want)
    torch.cuda.synchronize()
    assert sum(1 for w in bfp_ops._select_ws.values() if not w.pinned) <= bfp_ops._SELECT_WS_MAX
    assert (x.device.index, side.cuda_stream) in bfp_ops._select_ws


@pytest.mark.parametrize("dname", ["bf16", "f32"])
def test_unstructured_list_pipelined_over_two_streams(dname):
    """float_to_bfp_blocked_many / PreparedMany with unstructured pruning (BASELINE config 4's "all Linear weights"): the
    selection launch of tensor i + 1 beside the prune + quantize launch of tensor i on a side stream, workspaces in rotation.
    Every result equals the per-tensor call; serial == pipelined; repeated runs; captured in a hipGraph (fork / join)."""
    dt = torch.bfloat16 if dname == "bf16" else torch.float32
    shapes = [(512, 1024), (100, 192), (0, 64), (2048, 4096), (33, 100), (1024, 1024), (5, 64), (777, 320), (256, 256), (64, 4096), (1536, 512)]
    xs_c = [synth(r, c, dt, 0.02, seed=200 + i) for i, (r, c) in enumerate(shapes)]
    xs_c[5] = (xs_c[5].float() * 64).round().div(64).to(dt)                # a tie-heavy one
    xs = [x.to(DEV) for x in xs_c]
    c = cfg(w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=0.5)
    want = [bfp_ops.float_to_bfp_blocked(x, **c, identifier='w') for x in xs]
    got = bfp_ops.float_to_bfp_blocked_many(xs, identifier='w', **c)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape and torch.equal(g.view(torch.int32 if dt == torch.float32 else torch.int16),
                                                  w.view(torch.int32 if dt == torch.float32 else torch.int16)), (i, shapes[i])
    prep = bfp_ops.PreparedMany(xs, identifier='w', **c)
    for rep in range(3):
        outs = prep.run()
        torch.cuda.synchronize()
        for i, (g, w) in enumerate(zip(outs, want)):
            assert torch.equal(g, w), (rep, i)
    for idx, pl in prep._groups:                                            # the serial form of the same call
        for o in pl.outputs:
            o.zero_()
        pl.run(pipelined=False)
    for g, w in zip(prep.run(), want):
        assert torch.equal(g, w)
    # captured: the side stream joins the capture through the fork event and is joined back before the call returns
    for o in [o for _, pl in prep._groups for o in pl.outputs]:
        o.zero_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            outs = prep.run()
        graph.replay()
        graph.replay()
        side.synchronize()
        for i, (g, w) in enumerate(zip(outs, want)):
            assert torch.equal(g, w), ("graph", i)
    torch.cuda.current_stream().wait_stream(side)
    # other fractions through the one-shot entry point
    for frac in (0.25, 0.9):
        c2 = dict(c, sparsity_frac=frac)
        for g, x in zip(bfp_ops.float_to_bfp_blocked_many(xs[:5], identifier='w', **c2), xs[:5]):
            assert torch.equal(g, bfp_ops.float_to_bfp_blocked(x, **c2, identifier='w'))
