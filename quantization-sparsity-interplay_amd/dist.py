"""Row-sharded execution over the GPUs of one node (one process per GPU, torch.distributed; backend
'nccl' is RCCL over xGMI on ROCm).

No counterpart in the reference (it has no collectives on this path); the parity contract is
"gathered result == single-GPU result, byte for byte".

Why rows: every HBFP block and every N:M group lies inside one row (bfp_ops.py:50-59, :79-91), so a
contiguous slab of rows is an independent unit -- quantize / N:M need NO data-path collective.  The
only real exchange is unstructured pruning's global k-th magnitude: one all-gather of the per-rank radix
histograms per pass (132 KB per rank), from which every rank also reads the ties lower ranks hold.  An all-gather of
the result happens only when the caller asks for the tensor whole; on a fully connected xGMI node it
maps to concurrent point-to-point sends, so packed codes (2.9 MB per rank at the headline shape) or
the dequantised slab (11.3 MB) move at per-link rate on all 7 links at once.
"""
import torch
import torch.distributed as dist

from .bfp import bfp_ops
from . import native


def row_range(rows, world_size, rank):
    """rows [lo, hi) owned by `rank`: ceil(rows / world) per rank, last ranks may get fewer / none"""
    per = (rows + world_size - 1) // world_size
    lo = min(rows, rank * per)
    return lo, min(rows, lo + per)


def shard_rows(t, world_size, rank):
    """the slab of a [rows, ...] tensor that `rank` owns (a view)"""
    lo, hi = row_range(t.shape[0], world_size, rank)
    return t[lo:hi]


def all_gather_into(out, inp, group=None):
    """dist.all_gather_into_tensor; with the `gloo` backend (CPU rehearsal of the multi-rank paths, also with several
    ranks on ONE device) device tensors are staged through host memory, since gloo moves host bytes"""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host.view(-1), inp.contiguous().view(-1).cpu(), group=group)
        out.copy_(host)
        return
    dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1), group=group)


def all_gather_rows(local, rows_total, group=None):
    """reassemble a row-sharded tensor on every rank (plain concatenation along dim 0)"""
    world = dist.get_world_size(group)
    per = (rows_total + world - 1) // world
    if per * world == rows_total and local.shape[0] == per:
        out = torch.empty((rows_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        all_gather_into(out, local.contiguous(), group)
        return out
    # ragged: pad every slab to `per` rows, gather, cut
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((per * world,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    all_gather_into(out, pad, group)
    return out[:rows_total]


def _hist_allgather(group):
    """the one exchange of the unstructured path: every rank's radix histogram (132 KB) to every rank.  The resolve launch
    sums them for the global threshold and reads, from the per-rank counts of the threshold bin, how many ties lower
    ranks hold -- so there is no second (tie-count) exchange."""
    def fn(hist):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        allh = torch.empty(world * hist.numel(), dtype=hist.dtype, device=hist.device)
        all_gather_into(allh, hist.contiguous().view(-1), group)
        return allh.view(world, hist.numel()), world, rank
    return fn


def unstructured_sparsity_sharded(local, sparsity_frac, numel_global, group=None, engine=None):
    """_unstructured_sparsity (bfp_ops.py:61-71) of a row-sharded tensor: one threshold for the whole
    tensor, k = int(numel_global * frac), ties pruned lowest global flat index first.  A rank whose slab is
    empty (ragged split) still joins the histogram gather.
    engine: object with select_threshold / threshold_apply / workspace (default: the HIP engine)."""
    assert (sparsity_frac > 0)
    eng = engine or _NativeEngine()
    k = int(numel_global * sparsity_frac)
    if k > numel_global:
        raise RuntimeError("selected index k out of range")             # what torch.topk raises in the reference
    ws = eng.workspace(local.device)
    eng.select_threshold(local, k, ws, numel_global=numel_global, allgather=_hist_allgather(group))
    if local.numel() == 0:
        return local.clone()
    return eng.threshold_apply(local, ws).view(local.shape)


class _NativeEngine:
    def workspace(self, device):
        return bfp_ops._workspace(device)

    select_threshold = staticmethod(native.select_threshold)
    threshold_apply = staticmethod(native.threshold_apply)

    @staticmethod
    def quantize_threshold(t, ws, block_size, mant_bits, epsilon, seed):
        return native.quantize_threshold(t, ws, block_size, mant_bits, epsilon, stoch_seed=seed)[0]


def float_to_bfp_blocked_sharded(local, rows_total, group=None, gather=False, compute=None, engine=None,
                                 identifier='', **bfp_args):
    """float_to_bfp_blocked (bfp_ops.py:124-149) on this rank's row slab of a [rows_total, ...] tensor.
    gather=True reassembles the full fake-quantised tensor on every rank.
    compute: callable(tensor, **bfp_args, identifier=...) for the per-rank math (default: the HIP engine)."""
    compute = compute or bfp_ops.float_to_bfp_blocked
    sparsity = bfp_ops._select_sparsity(bfp_args.get('in_sparsity'), bfp_args.get('w_sparsity'),
                                        bfp_args.get('grad_sparsity'), identifier)
    if sparsity and bfp_args.get('sparsity_mode') == 'unstructured':
        # the one path with a real exchange step: global threshold
        cols = 1
        for d in local.shape[1:]:
            cols *= int(d)
        numel_global = rows_total * cols
        dense = dict(bfp_args, in_sparsity=False, w_sparsity=False, grad_sparsity=False)
        eng = engine or _NativeEngine()
        fused = (bfp_args.get('first') == 's' and bfp_args.get('sparsity_num_format') == 'bfp' and hasattr(eng, 'quantize_threshold')
                 and compute is bfp_ops.float_to_bfp_blocked)
        if fused:
            # threshold, then prune + quantize in ONE pass over the slab (as on a single device)
            assert (bfp_args['sparsity_frac'] > 0)
            assert (bfp_args['block_size'] > 0)                         # bfp_ops.py:130
            if int(numel_global * bfp_args['sparsity_frac']) > numel_global:
                raise RuntimeError("selected index k out of range")     # what torch.topk raises in the reference
            ws = eng.workspace(local.device)
            eng.select_threshold(local, int(numel_global * bfp_args['sparsity_frac']), ws, numel_global=numel_global,
                                 allgather=_hist_allgather(group))
            if local.numel() == 0:
                out = local.clone()
            else:
                mb = bfp_args['weight_mant_bits'] if bfp_args.get('sgd_update') else bfp_args['mant_bits']
                seed = bfp_ops._seed_for(bfp_args['rounding_mode'])
                out = bfp_ops._stoc_dtype(eng.quantize_threshold(local, ws, bfp_args['block_size'], mb, bfp_args['epsilon'], seed)
                                          .view(local.shape), bfp_args['rounding_mode'])
        elif bfp_args.get('first') == 's':
            pruned = unstructured_sparsity_sharded(local, bfp_args['sparsity_frac'], numel_global, group, engine)
            out = compute(pruned, **dense, identifier=identifier)
        else:
            q = compute(local, **dense, identifier=identifier)
            out = unstructured_sparsity_sharded(q, bfp_args['sparsity_frac'], numel_global, group, engine)
    else:
        out = compute(local, **bfp_args, identifier=identifier)
    return all_gather_rows(out, rows_total, group) if gather else out


_list_ws = {}


def _list_workspaces(device, n):
    """n select workspaces of the device for the list exchange (one per tensor of a group: its windows live there from the
    histogram launch to the apply launch); kept, they are 7.4 MB each"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _list_ws.setdefault(idx, [])
    while len(pool) < n:
        pool.append(native.SelectWorkspace(torch.device("cuda", idx)))
    return pool[:n]


def _block_allgather(group):
    """the ONE exchange per radix pass of a whole group of tensors: [n, ENTRIES] (copies folded) -> [world, n, ENTRIES]"""
    def fn(block):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        allb = torch.empty((world,) + tuple(block.shape), dtype=block.dtype, device=block.device)
        all_gather_into(allb, block.contiguous(), group)
        return allb, world, rank
    return fn


def _many_sharded_one_exchange(locals_, rows_totals, group, gather, identifier, group_size, bfp_args):
    """the fused unstructured path (prune, then quantize) of a list of row-sharded tensors with ONE histogram exchange per radix pass
    and GROUP of tensors: all selection histograms of the group first, one all-gather of their folded copies (132 KB per tensor and
    rank), all resolve launches, then the prune + quantize launches over two streams.  LLaMA-13B's 280 weights: 9 collectives of
    ~4 MB per rank instead of 280 of 1 MB (each of which costs its full latency with nothing to run beside it)."""
    n = len(locals_)
    frac = bfp_args['sparsity_frac']
    assert (frac > 0)
    assert (bfp_args['block_size'] > 0)                                 # bfp_ops.py:130
    mb = bfp_args['weight_mant_bits'] if bfp_args.get('sgd_update') else bfp_args['mant_bits']
    dev = locals_[0].device
    main = torch.cuda.current_stream(dev)
    side = native.aux_streams(dev, 1)[0]
    outs = [None] * n
    for g0 in range(0, n, group_size):
        idx = list(range(g0, min(n, g0 + group_size)))
        ngs, ks = [], []
        for i in idx:
            cols = 1
            for d in locals_[i].shape[1:]:
                cols *= int(d)
            ng = rows_totals[i] * cols
            k = int(ng * frac)
            if k > ng:
                raise RuntimeError("selected index k out of range")     # what torch.topk raises in the reference
            ngs.append(ng); ks.append(k)
        wss = _list_workspaces(dev, len(idx))
        native.select_threshold_list([locals_[i] for i in idx], ks, wss, ngs, _block_allgather(group))
        side.wait_stream(main)
        for j, i in enumerate(idx):
            t = locals_[i]
            if t.numel() == 0:
                outs[i] = t.clone()
                continue
            s = side if (j & 1) else main
            with torch.cuda.stream(s):
                seed = bfp_ops._seed_for(bfp_args['rounding_mode'])
                o = native.quantize_threshold(t, wss[j], bfp_args['block_size'], mb, bfp_args['epsilon'], stoch_seed=seed)[0]
                outs[i] = bfp_ops._stoc_dtype(o.view(t.shape), bfp_args['rounding_mode'])
                if s is side:
                    outs[i].record_stream(main)
        main.wait_stream(side)                                            # (also: the group's workspaces are free for the next group)
    return [all_gather_rows(o, r, group) for o, r in zip(outs, rows_totals)] if gather else outs


def float_to_bfp_blocked_many_sharded(locals_, rows_totals, group=None, gather=False, identifier='', lanes=4, compute=None, engine=None,
                                      exchange='list', group_size=32, **bfp_args):
    """float_to_bfp_blocked_sharded for a LIST of row-sharded tensors -- every Linear weight of a model, each rank holding its
    row slab of every weight (BASELINE config 4 as specified: LLaMA-13B, 50 % unstructured, row-sharded).
    Structured / dense configurations have no exchange: the slabs go through one list call (bfp_ops.float_to_bfp_blocked_many:
    large slabs in launches of their own over two streams), then the optional all-gathers.
    Unstructured pruning has one histogram all-gather per tensor (three for fp32) between its launches; issued tensor after
    tensor on one stream, every exchange would leave the device idle for its whole latency.  Two remedies:
      exchange='list' (default; prune-then-quantize configurations): ONE exchange per radix pass for a whole group of `group_size`
        tensors -- all their selection histograms first, the copies of each folded, one all-gather, all resolve launches, then the
        prune + quantize launches (_many_sharded_one_exchange; native.select_threshold_list);
      exchange='tensor' (and every other configuration): the tensors are dealt round-robin to `lanes` streams (the current one and
        side streams, each with its own select workspace), so that the exchange of one tensor runs beside the kernels of the
        others; the collectives are still ISSUED in list order on every rank, which is all a communicator needs.
    Results are byte-identical to the per-tensor call.
    compute / engine: as in float_to_bfp_blocked_sharded (CPU rehearsal of the protocol with stand-ins: tensor after tensor)."""
    locals_ = list(locals_)
    n = len(locals_)
    assert len(rows_totals) == n
    sparsity = bfp_ops._select_sparsity(bfp_args.get('in_sparsity'), bfp_args.get('w_sparsity'),
                                        bfp_args.get('grad_sparsity'), identifier)
    unstructured = bool(sparsity) and bfp_args.get('sparsity_mode') == 'unstructured'
    on_gpu = n > 0 and all(t.device.type == "cuda" for t in locals_)
    if not unstructured and on_gpu and compute is None:
        outs = bfp_ops.float_to_bfp_blocked_many(locals_, identifier=identifier, **bfp_args)
        return [all_gather_rows(o, r, group) for o, r in zip(outs, rows_totals)] if gather else outs
    fused = (bfp_args.get('first') == 's' and bfp_args.get('sparsity_num_format') == 'bfp' and compute is None and engine is None)
    if on_gpu and n >= 2 and exchange == 'list' and fused and len({(t.dtype, t.device) for t in locals_}) == 1:
        return _many_sharded_one_exchange(locals_, rows_totals, group, gather, identifier, max(1, int(group_size)), bfp_args)
    if not on_gpu or lanes <= 1 or n < 2:
        return [float_to_bfp_blocked_sharded(t, r, group, gather, compute, engine, identifier=identifier, **bfp_args)
                for t, r in zip(locals_, rows_totals)]
    dev = locals_[0].device
    main = torch.cuda.current_stream(dev)
    streams = [main] + native.aux_streams(dev, min(lanes, n) - 1)
    for s in streams[1:]:
        s.wait_stream(main)
    outs = [None] * n
    for i, (t, r) in enumerate(zip(locals_, rows_totals)):
        s = streams[i % len(streams)]
        with torch.cuda.stream(s):
            outs[i] = float_to_bfp_blocked_sharded(t, r, group, gather, compute, engine, identifier=identifier, **bfp_args)
            if s is not main:
                outs[i].record_stream(main)                        # (allocated on a side stream, consumed on the caller's)
    for s in streams[1:]:
        main.wait_stream(s)
    return outs


def float_to_bfp_packed_sharded(local, rows_total, mant_bits, block_size, group=None, gather=False, **kw):
    """packed codes + exponents of this rank's slab; gather=True all-gathers both (the cheap wire format)"""
    codes, exps = bfp_ops.float_to_bfp_packed(local, mant_bits, block_size, **kw)
    if gather:
        return all_gather_rows(codes, rows_total, group), all_gather_rows(exps, rows_total, group)
    return codes, exps


_side_streams = {}


def _side_stream(device):
    """the persistent side stream (one per device) that carries the overlapped gathers"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _side_streams.get(idx)
    if s is None:
        s = _side_streams[idx] = torch.cuda.Stream(torch.device("cuda", idx))
    return s


def gather_overlapped(local, rows_total, compute, chunks=4, group=None, out=None):
    """out = all-gather over ranks of compute(local), with the gather of row-chunk i overlapped with the
    compute of chunk i+1: the slab is cut into `chunks` row pieces; each piece is computed on the current
    stream and, as soon as it is done (an event orders the two), all-gathered on a persistent side stream with
    ONE all_gather_into_tensor into a contiguous [world, piece rows, ...] staging buffer, from where one strided
    copy (also on the side stream) puts every rank's piece into its rows of the full result.  Needs an even row
    split (rows_total % world == 0).
    compute(piece) -> tensor with the same number of rows (quantize / N:M: rows are independent)."""
    world = dist.get_world_size(group)
    per = rows_total // world
    assert per * world == rows_total and local.shape[0] == per, "gather_overlapped needs an even row split"
    bounds = [per * i // chunks for i in range(chunks + 1)]
    on_gpu = local.device.type == "cuda"
    side = _side_stream(local.device) if on_gpu else None
    if on_gpu:
        side.wait_stream(torch.cuda.current_stream(local.device))          # (the previous call's readers of `out` / the stage are done)
    for i in range(chunks):
        lo, hi = bounds[i], bounds[i + 1]
        if hi == lo:
            continue
        piece = compute(local[lo:hi]).contiguous()
        if out is None:
            out = torch.empty((rows_total,) + tuple(piece.shape[1:]), dtype=piece.dtype, device=piece.device)
        dst = out.view((world, per) + tuple(piece.shape[1:]))[:, lo:hi]       # every rank's rows [lo, hi) of its slab
        if on_gpu:
            done = torch.cuda.Event()
            done.record()
            with torch.cuda.stream(side):
                side.wait_event(done)
                stage = torch.empty((world,) + tuple(piece.shape), dtype=piece.dtype, device=piece.device)
                all_gather_into(stage, piece, group)
                dst.copy_(stage)
                piece.record_stream(side)
        else:
            stage = torch.empty((world,) + tuple(piece.shape), dtype=piece.dtype)
            dist.all_gather_into_tensor(stage.view(-1), piece.view(-1), group=group)
            dst.copy_(stage)
    if on_gpu:
        torch.cuda.current_stream(local.device).wait_stream(side)
    return out


def all_gather_packed(local, rows_total, mant_bits, block_size, group=None, dequantize=True, **kw):
    """Quantize this rank's slab to packed HBFP, all-gather the PACKED bytes (codes + exponents: ~0.52 B per
    element for HBFP4 instead of 2 B), and decode locally.  Returns the full fake-quantised tensor
    (dequantize=True) or the gathered PackedBFP."""
    p = bfp_ops.PackedBFP.quantize(local, mant_bits, block_size, **kw)
    codes = all_gather_rows(p.codes, rows_total, group)
    exps = all_gather_rows(p.exps, rows_total, group)
    full = bfp_ops.PackedBFP(codes, exps, (rows_total,) + tuple(local.shape[1:]), local.dtype, mant_bits, block_size, p.code_bits)
    return full.dequantize() if dequantize else full
