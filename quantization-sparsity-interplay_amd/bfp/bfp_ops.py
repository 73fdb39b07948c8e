"""Drop-in operator surface of the reference's src/transformers/bfp/bfp_ops.py, backed by the
MI355X HIP engine (libbfpq.so).  Same names, argument order, aliasing and exceptions as the
reference (file:line cited per function); the tensor math itself is NOT here -- it is in
csrc/bfpq_kernels.hip, reached through ..native.  ROCm-device tensors only: a CPU tensor raises
NativeUnavailable (there is deliberately no CPU fallback).

Differences from the reference, all documented in DESIGN.md:
  * 'stoc' rounding draws its dither from a counter-based generator seeded from torch's CPU RNG
    (torch.manual_seed controls it); the reference's torch.rand stream is not reproduced.
  * _unstructured_sparsity: which of the elements EQUAL to the threshold are pruned is "lowest flat
    index first"; the reference's choice is an artefact of a sequential introselect.
"""
import collections

import torch
import torch.nn.functional as F

from .. import native
from .. import ops as _ops          # registers torch.ops.bfpq.* (traceable entry points)

__all__ = ["rounding_modes", "round_tensor", "get_exponent", "float_to_bfp_blocked", "float_to_bfp_blocked_many", "PreparedMany", "float_to_bfp_packed",
           "sparsify", "unpack_bfp_args", "F_linear_bfp", "F_matmul_bfp", "BFPLinear", "BFPConv2d", "WeightCache", "PackedBFP"]


class rounding_modes:
    """reference: bfp_ops.py:16-18"""
    STOC, DETERM = 'stoc', 'determ'
    modes = [STOC, DETERM]


def _seed_for(mode):
    """0 -> round-half-even; otherwise a non-zero 62-bit seed taken from torch's CPU generator"""
    if mode == rounding_modes.DETERM:
        return 0
    if mode == rounding_modes.STOC:
        return int(torch.randint(1, 2 ** 62, (1,)).item())
    raise NotImplementedError("Rounding mode %s is not implemented", mode)


def _quantize_nm_ref_dtype(t, block_size, mant_bits, epsilon, rounding_mode, N=0, M=0, sparsify_first=True):
    """quantize_nm with the reference's output dtype: 'stoc' on a half tensor comes back as fp32.
    The single-pass kernel writes that fp32 image directly; other shapes convert afterwards."""
    seed = _seed_for(rounding_mode)
    if seed == 0:                                      # round-half-even
        if torch.compiler.is_compiling():              # under torch.compile / export: the registered (traceable) op
            return torch.ops.bfpq.fake_quantize(t, int(block_size), int(mant_bits), float(epsilon), int(N), int(M), bool(sparsify_first), 0)
        return _fast_quant(block_size, mant_bits, epsilon, N, M, sparsify_first)(t)
    src = t.contiguous()
    if seed and t.dtype != torch.float32 and t.numel() and block_size > 0 and native.is_fused(src, block_size, N, M):
        _, y, _ = native.quantize_nm(src, block_size, mant_bits, epsilon, N=N, M=M, sparsify_first=sparsify_first,
                                     want_deq=False, code_bits=32, stoch_seed=seed)
        return y.view(t.shape)
    y, _, _ = native.quantize_nm(src, block_size, mant_bits, epsilon, N=N, M=M, sparsify_first=sparsify_first, stoch_seed=seed)
    return _stoc_dtype(y.view(t.shape), rounding_mode)


_fast_quants = {}


def _fast_quant(block_size, mant_bits, epsilon, N, M, sparsify_first):
    """the bound form of the drop-in call for one configuration (native.FastQuant), made once"""
    key = (block_size, mant_bits, epsilon, N, M, sparsify_first)
    f = _fast_quants.get(key)
    if f is None:
        f = _fast_quants[key] = native.FastQuant(block_size, mant_bits, epsilon, N, M, sparsify_first)
    return f


def _stoc_dtype(t, mode):
    """reference quirk: 'stoc' adds fp32 noise, so a half input comes back as fp32"""
    return t.float() if (mode == rounding_modes.STOC and t.dtype != torch.float32) else t


def round_tensor(t, mode, device):
    """reference: bfp_ops.py:20-27 (elementwise; kept for API completeness, not on the fused path)"""
    if mode == rounding_modes.STOC:
        sampled = torch.rand(t.shape, device=t.device) - 0.5
        return sampled.add_(t).round()
    elif mode == rounding_modes.DETERM:
        return t.round()
    raise NotImplementedError("Rounding mode %s is not implemented", mode)


def get_exponent(t, epsilon):
    """reference: bfp_ops.py:29-33 -- t is [nblk, block]; returns the shared exponent per row, [nblk, 1],
    in t.dtype.  (int8 transport: exponents outside [-127, 127] saturate; NaN blocks come back as NaN.)"""
    native.require_device_tensor(t)
    assert t.dim() == 2
    _, _, e = native.quantize_nm(t, t.shape[1], 0, epsilon, want_deq=False, want_exp=True)
    out = e.to(t.dtype).view(-1, 1)
    return torch.where(e.view(-1, 1) == -128, torch.full_like(out, float('nan')), out)


def _convert_blocked_float_to_bfp(t, mant_bits, epsilon, rounding_mode, device):
    """reference: bfp_ops.py:35-44 -- every row of the 2-D t is one block"""
    native.require_device_tensor(t)
    assert t.dim() == 2
    y, _, _ = native.quantize_nm(t, t.shape[1], mant_bits, epsilon, stoch_seed=_seed_for(rounding_mode))
    return _stoc_dtype(y, rounding_mode)


def _no_sparsity_float_to_bfp(t, block_size, mant_bits, epsilon, rounding_mode, device):
    """reference: bfp_ops.py:46-59 -- blocks of block_size along the last dim, zero-padded per row"""
    native.require_device_tensor(t)
    return _quantize_nm_ref_dtype(t, block_size, mant_bits, epsilon, rounding_mode)


_select_ws = collections.OrderedDict()
_spare_ws = {}                     # device index -> workspaces made (and really zeroed) outside any capture, for streams first met inside one
_SELECT_WS_MAX = 16                # workspaces kept (5 MB each): least recently used (device, stream) pairs beyond that are dropped
_SPARES = 2


def _workspace(device):
    """select workspace of the current stream of `device` (the launches of one call talk through it, so two streams
    must not share one).  Kept per (device, stream), least recently used first out -- except workspaces a captured graph
    refers to, which stay for good (the graph holds their address).
    A workspace cannot be BORN inside a hipGraph capture: its zero-fill would be recorded instead of run, and its memory would
    belong to the graph's pool.  torch.cuda.graph() captures on a stream of its own, so a stream first met inside a capture
    gets one of the spare workspaces that every eager call keeps in stock; only a capture with no eager unstructured call
    before it on this device (capture needs a warm-up anyway) finds none."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    capturing = torch.cuda.is_current_stream_capturing()
    ws = _select_ws.get(key)
    if ws is None:
        if capturing:
            spares = _spare_ws.get(idx)
            if not spares:
                raise RuntimeError("the unstructured path needs a workspace that exists before graph capture starts: "
                                   "run one unstructured call on this device outside torch.cuda.graph(...) first (outside the capture)")
            ws = spares.pop()
        else:
            ws = native.SelectWorkspace(torch.device("cuda", idx))
        _select_ws[key] = ws
        if len(_select_ws) > _SELECT_WS_MAX:
            for old in [k for k, w in _select_ws.items() if not w.pinned and k != key][:len(_select_ws) - _SELECT_WS_MAX]:
                del _select_ws[old]
    else:
        _select_ws.move_to_end(key)
    if capturing:
        ws.pinned = True
    else:
        spares = _spare_ws.setdefault(idx, [])
        while len(spares) < _SPARES:
            spares.append(native.SelectWorkspace(torch.device("cuda", idx)))
    return ws


def _unstructured_sparsity(t, device, sparsity_frac=0):
    """reference: bfp_ops.py:61-71 -- zero the int(numel*frac) smallest |v| of the whole tensor"""
    assert (sparsity_frac > 0)
    native.require_device_tensor(t)
    if t.numel() == 0:
        return t.clone()
    k = int(t.numel() * sparsity_frac)
    if k > t.numel():
        raise RuntimeError("selected index k out of range")      # what torch.topk raises in the reference
    ws = _workspace(t.device)
    native.select_threshold(t, k, ws)
    return native.threshold_apply(t, ws).view(t.shape)


def _structured_N_M_sparsity(t, device, N=0, M=0):
    """reference: bfp_ops.py:73-91 -- per group of M along the last dim keep the N largest |v|"""
    assert ((N > 0) and (M > 0) and (N <= M))
    native.require_device_tensor(t)
    if N == M:
        return t.contiguous().clone().view(t.shape)
    y, _, _ = native.quantize_nm(t, 0, 0, 0.0, N=N, M=M)
    return y.view(t.shape)


def _sparsify(t, sparsity, sparsity_mode, device, N, M, sparsity_frac):
    """reference: bfp_ops.py:93-102"""
    if sparsity == True:  # noqa: E712  (the reference compares with ==; 1 counts as True)
        if sparsity_mode == 'structured':
            return _structured_N_M_sparsity(t, device, N, M)
        elif sparsity_mode == 'unstructured':
            return _unstructured_sparsity(t, device, sparsity_frac)
        else:
            raise ValueError(f'Unknown sparsity mode: {sparsity_mode} given as argument')
    else:
        return t


def _quantize(t, num_format, block_size, mant_bits, weight_mant_bits, sgd_update, epsilon, rounding_mode, device, identifier):
    """reference: bfp_ops.py:104-122"""
    if num_format == 'fp32':
        return t
    elif num_format == 'bfp':
        if sgd_update:
            mant_bits = weight_mant_bits
        return _no_sparsity_float_to_bfp(t, block_size, mant_bits, epsilon, rounding_mode, device)
    elif num_format == 'int':
        # per-channel symmetric integer grid (int_ops.Quantizer with its defaults); fp32 result as in the reference
        if sgd_update:
            mant_bits = weight_mant_bits
        quant_t = native.int_quantize(t, mant_bits, identifier == 'w')
        assert (t.shape == quant_t.shape)
        return quant_t
    else:
        raise ValueError(f'Unknown quantization format: {num_format} given as argument')


def _select_sparsity(in_sparsity, w_sparsity, grad_sparsity, identifier):
    """reference: bfp_ops.py:132-139"""
    if in_sparsity == True and identifier == 'in':  # noqa: E712
        return True
    elif w_sparsity == True and identifier == 'w':  # noqa: E712
        return True
    elif grad_sparsity == True and identifier == 'grad':  # noqa: E712
        return True
    return False


def float_to_bfp_blocked(t, mant_bits, epsilon, rounding_mode, device, block_size,
                         num_format, weight_mant_bits, in_sparsity, w_sparsity, grad_sparsity,
                         sparsity_frac, N, M, sparsity_num_format, first, sparsity_mode, identifier='', sgd_update=False,
                         mx_w_elem_format='', mx_a_elem_format='', scale_bits=0, bfloat=0):
    """reference: bfp_ops.py:124-149.  Structured sparsity + 'bfp' (or 'fp32') runs as ONE fused
    kernel launch in either order; everything else composes the same steps the reference does."""
    assert (num_format == 'bfp')
    assert (((sparsity_num_format == 'bfp') and (block_size > 0)) or (sparsity_num_format == 'fp32') or (sparsity_num_format == 'int'))

    sparsity = _select_sparsity(in_sparsity, w_sparsity, grad_sparsity, identifier)

    if sparsity and sparsity_mode == 'structured' and sparsity_num_format == 'bfp':
        assert ((N > 0) and (M > 0) and (N <= M))
        native.require_device_tensor(t)
        mb = weight_mant_bits if sgd_update else mant_bits
        return _quantize_nm_ref_dtype(t, block_size, mb, epsilon, rounding_mode, N=N if N < M else 0, M=M if N < M else 0,
                                      sparsify_first=(first == 's'))

    if sparsity and sparsity_mode == 'unstructured' and sparsity_num_format == 'bfp' and first == 's':
        # global threshold (radix select), then prune + quantize in ONE pass over the tensor
        assert (sparsity_frac > 0)
        native.require_device_tensor(t)
        if t.numel() == 0:
            return t.clone()
        k = int(t.numel() * sparsity_frac)
        if k > t.numel():
            raise RuntimeError("selected index k out of range")
        mb = weight_mant_bits if sgd_update else mant_bits
        ws = _workspace(t.device)
        if rounding_mode == rounding_modes.DETERM:
            # one entry point: selection launch + fused prune + quantize launch
            return native.prune_quantize(t, k, ws, block_size, mb, epsilon).view(t.shape)
        native.select_threshold(t, k, ws)
        y, _, _ = native.quantize_threshold(t, ws, block_size, mb, epsilon, stoch_seed=_seed_for(rounding_mode))
        return _stoc_dtype(y.view(t.shape), rounding_mode)

    if first == 's':
        sparse_t = _sparsify(t, sparsity, sparsity_mode, device, N, M, sparsity_frac)
        return _quantize(sparse_t, sparsity_num_format, block_size, mant_bits, weight_mant_bits, sgd_update, epsilon, rounding_mode, device, identifier)
    else:
        quant_t = _quantize(t, sparsity_num_format, block_size, mant_bits, weight_mant_bits, sgd_update, epsilon, rounding_mode, device, identifier)
        return _sparsify(quant_t, sparsity, sparsity_mode, device, N, M, sparsity_frac)


# ---- additive public surface (no counterpart in the reference) -------------------------------
def _unstructured_lean(a, sparsity, sgd_update=False):
    """the configuration is the s-first unstructured drop-in op (prune int(numel * frac) elements, then HBFP, round-half-even)"""
    return (sparsity and a['num_format'] == 'bfp' and a['sparsity_num_format'] == 'bfp' and a['sparsity_mode'] == 'unstructured'
            and a['first'] == 's' and a['rounding_mode'] == rounding_modes.DETERM and a['block_size'] > 0 and a['sparsity_frac'] > 0
            and not sgd_update)


def _prune_lists(tensors, a):
    """PruneQuantizeList per (device, dtype) group of `tensors` for configuration `a`; [(indices, list)]"""
    groups = {}
    for i, t in enumerate(tensors):
        native.require_device_tensor(t)
        k = int(t.numel() * a['sparsity_frac'])
        if k > t.numel():
            raise RuntimeError("selected index k out of range")      # what torch.topk raises in the reference
        groups.setdefault((t.device, t.dtype), []).append((i, k))
    return [([i for i, _ in g], native.PruneQuantizeList([tensors[i].contiguous() for i, _ in g], [k for _, k in g],
                                                         a['block_size'], a['mant_bits'], a['epsilon'])) for g in groups.values()]


MANY_LANES = 3                       # streams a list of tensors without a list kernel is dealt to (the current one + side streams)
MANY_LANES_MIN_BYTES = 24 << 20      # ... when at least two of its tensors are this large (smaller ones: the host call costs more than it hides)


def _many_over_streams(tensors, fn):
    """[fn(t) for t in tensors] for configurations that have no list form in the library (several launches per tensor: quantize-first
    unstructured pruning, stochastic rounding, the 'int' format, N:8 on ragged rows ...).  The calls are independent, so large device
    tensors are dealt round-robin to MANY_LANES streams -- one tensor's launch boundaries, ramps and tails beside the other tensors'
    streaming, as the library's own list calls do it (native.PruneQuantizeList) -- with one fork and one join around the list."""
    big = [t for t in tensors if t.device.type == "cuda" and t.numel() * t.element_size() >= MANY_LANES_MIN_BYTES]
    if MANY_LANES < 2 or len(big) < 2 or len({t.device for t in big}) != 1 or any(t.device != big[0].device for t in tensors):
        return [fn(t) for t in tensors]
    dev = big[0].device
    main = torch.cuda.current_stream(dev)
    streams = [main] + native.aux_streams(dev, MANY_LANES - 1)
    for s in streams[1:]:
        s.wait_stream(main)
    out = []
    for i, t in enumerate(tensors):
        s = streams[i % len(streams)]
        with torch.cuda.stream(s):
            y = fn(t)
            if s is not main and y is not t:
                y.record_stream(main)                       # (allocated on a side stream, consumed on the caller's)
        out.append(y)
    for s in streams[1:]:
        main.wait_stream(s)
    return out


def float_to_bfp_blocked_many(tensors, identifier='', **bfp_args):
    """float_to_bfp_blocked (bfp_ops.py:124-149) for a LIST of tensors with one configuration -- e.g. every Linear weight of
    a model -- in as few launches as possible: one per 64 tensors per dtype/device for the 'bfp' format with structured or no
    pruning and round-half-even (large tensors: a launch each, spread over two streams); for unstructured pruning before
    quantization, two launches per tensor over four independent lanes (native.PruneQuantizeList); anything else is the
    per-tensor call, large tensors dealt to a few streams (_many_over_streams).  Returns the list of results in order."""
    a = unpack_bfp_args(dict(bfp_args))
    tensors = list(tensors)
    sparsity = _select_sparsity(a['in_sparsity'], a['w_sparsity'], a['grad_sparsity'], identifier)
    if _unstructured_lean(a, sparsity, bool(bfp_args.get('sgd_update'))):
        out = [None] * len(tensors)
        for idx, pl in _prune_lists(tensors, a):
            for i, y in zip(idx, pl.run()):
                out[i] = y.view(tensors[i].shape)
        return out
    lean = (a['num_format'] == 'bfp' and a['sparsity_num_format'] == 'bfp' and a['rounding_mode'] == rounding_modes.DETERM
            and a['block_size'] > 0 and not bfp_args.get('sgd_update')
            and (not sparsity or (a['sparsity_mode'] == 'structured' and 0 < a['N'] <= a['M'])))
    if not lean:
        return _many_over_streams(tensors, lambda t: float_to_bfp_blocked(t, **a, identifier=identifier, sgd_update=bool(bfp_args.get('sgd_update'))))
    nm = sparsity and a['N'] < a['M']
    f = _fast_quant(a['block_size'], a['mant_bits'], a['epsilon'], a['N'] if nm else 0, a['M'] if nm else 0, a['first'] == 's')
    out = [None] * len(tensors)
    groups = {}
    for i, t in enumerate(tensors):
        native.require_device_tensor(t)
        groups.setdefault((t.device, t.dtype), []).append(i)
    for idx in groups.values():
        for i, y in zip(idx, f.many([tensors[i] for i in idx])):
            out[i] = y.view(tensors[i].shape)
    return out


class PreparedMany:
    """float_to_bfp_blocked_many with the per-call host work done once: run() re-quantizes the bound tensors (whose storage
    must stay put, e.g. a model's weights) into the same output tensors -- one launch per 64 tensors (structured / dense), or
    the two-stream pipeline of the unstructured path; one ctypes call per dtype/device group."""

    def __init__(self, tensors, identifier='', **bfp_args):
        a = unpack_bfp_args(dict(bfp_args))
        self.tensors = list(tensors)
        self._args, self._ident = a, identifier
        sparsity = _select_sparsity(a['in_sparsity'], a['w_sparsity'], a['grad_sparsity'], identifier)
        lean = (a['num_format'] == 'bfp' and a['sparsity_num_format'] == 'bfp' and a['rounding_mode'] == rounding_modes.DETERM
                and a['block_size'] > 0 and (not sparsity or (a['sparsity_mode'] == 'structured' and 0 < a['N'] <= a['M'])))
        self._groups = None
        if _unstructured_lean(a, sparsity):
            for t in self.tensors:
                if not t.is_contiguous():
                    raise ValueError("PreparedMany needs contiguous inputs (their storage is bound)")
            self._groups = _prune_lists(self.tensors, a)
        elif lean:
            nm = sparsity and a['N'] < a['M']
            f = _fast_quant(a['block_size'], a['mant_bits'], a['epsilon'], a['N'] if nm else 0, a['M'] if nm else 0, a['first'] == 's')
            groups = {}
            for i, t in enumerate(self.tensors):
                native.require_device_tensor(t)
                groups.setdefault((t.device, t.dtype), []).append(i)
            self._groups = [(idx, f.prepare([self.tensors[i] for i in idx])) for idx in groups.values()]

    def run(self):
        out = [None] * len(self.tensors)
        if self._groups is None:
            return [float_to_bfp_blocked(t, **self._args, identifier=self._ident) for t in self.tensors]
        for idx, prep in self._groups:
            for i, y in zip(idx, prep.run()):
                out[i] = y.view(self.tensors[i].shape)
        return out


def sparsify(t, sparsity_mode, N=0, M=0, sparsity_frac=0):
    """public alias of _sparsify with sparsity=True"""
    return _sparsify(t, True, sparsity_mode, str(t.device), N, M, sparsity_frac)


def float_to_bfp_packed(t, mant_bits, block_size, epsilon=1e-8, N=0, M=0, first='s', code_bits=None, with_dequant=False):
    """Packed HBFP: integer mantissas (two's complement, 4/8/16 bits; 4-bit = two per byte, low nibble
    first) + one int8 shared exponent per block, value = code * 2**(exp - mant_bits).  N:M pruning is
    implied by zero codes.  Returns (codes, exps) or (codes, exps, dequantised)."""
    native.require_device_tensor(t)
    if code_bits is None:
        code_bits = 4 if mant_bits <= 3 else (8 if mant_bits <= 7 else 16)
    deq, codes, exps = native.quantize_nm(t, block_size, mant_bits, epsilon, N=N, M=M, sparsify_first=(first == 's'),
                                          want_deq=with_dequant, code_bits=code_bits, want_exp=True)
    return (codes, exps, deq.view(t.shape)) if with_dequant else (codes, exps)


class PackedBFP:
    """A tensor in packed HBFP form: integer mantissas + one int8 exponent per block (+ what is needed to
    decode it).  N:M / unstructured pruning needs no extra field: pruned elements are zero codes."""

    def __init__(self, codes, exps, shape, dtype, mant_bits, block_size, code_bits):
        self.codes, self.exps = codes, exps
        self.shape, self.dtype = tuple(shape), dtype
        self.mant_bits, self.block_size, self.code_bits = int(mant_bits), int(block_size), int(code_bits)

    @classmethod
    def quantize(cls, t, mant_bits, block_size, epsilon=1e-8, N=0, M=0, first='s', code_bits=None):
        if code_bits is None:
            code_bits = 4 if mant_bits <= 3 else (8 if mant_bits <= 7 else 16)
        codes, exps = float_to_bfp_packed(t, mant_bits, block_size, epsilon, N, M, first, code_bits)
        return cls(codes, exps, t.shape, t.dtype, mant_bits, block_size, code_bits)

    @classmethod
    def quantize_unstructured(cls, t, mant_bits, block_size, sparsity_frac, epsilon=1e-8, code_bits=None):
        """packed Q(S_unstructured(t)) -- global magnitude pruning of int(numel * frac) elements, then HBFP (the reference's
        float_to_bfp_blocked with sparsity_mode 'unstructured', first 's': bfp_ops.py:61-71, 141-144) -- in the three launches of the
        unstructured path, the last one writing codes + exponents instead of the tensor"""
        native.require_device_tensor(t)
        if code_bits is None:
            code_bits = 4 if mant_bits <= 3 else (8 if mant_bits <= 7 else 16)
        assert sparsity_frac > 0
        k = int(t.numel() * sparsity_frac)
        if k > t.numel():
            raise RuntimeError("selected index k out of range")
        ws = _workspace(t.device)
        native.select_threshold(t, k, ws)
        _, codes, exps = native.quantize_threshold(t, ws, block_size, mant_bits, epsilon, want_deq=False, code_bits=code_bits, want_exp=True)
        return cls(codes, exps, t.shape, t.dtype, mant_bits, block_size, code_bits)

    def dequantize(self, out=None):
        """== the reference's fake-quantised tensor (a -0.0 there is +0.0 here)"""
        cols = self.shape[-1] if len(self.shape) else 1
        return native.dequantize(self.codes, self.exps, cols, self.dtype, self.block_size, self.mant_bits, self.code_bits,
                                 out=None if out is None else out.view(-1, cols)).view(self.shape)

    def nbytes(self):
        return self.codes.numel() * self.codes.element_size() + self.exps.numel()

    def _tiled_ok(self):
        N, K = self.shape
        return N % 16 == 0 and K % 128 == 0 and K >= 256

    def linear_decode(self, x, x_mant_bits=7, epsilon=1e-8):
        """x @ W^T for <= 16 tokens (<= 64 with the MFMA-tiled layout) straight from the packed weight (self is W [N, K],
        4-bit codes, block 64):
        the activation is quantized to HBFP(x_mant_bits + 1) block 64 and every block's dot product is an exact
        integer sum on the int8 matrix cores (native.hbfp_linear_decode)."""
        assert self.code_bits == 4 and self.block_size == 64 and len(self.shape) == 2
        N, K = self.shape
        if N % 16 == 0 and K % 128 == 0 and K >= 256:                  # MFMA-tiled copy of the weight, built once
            if getattr(self, "_tiles", None) is None:
                self._tiles = native.mfma_tiles(self.codes, self.exps)
            return native.hbfp_linear_decode_tiled(x, self._tiles[0], self._tiles[1], N, self.mant_bits, x_mant_bits, epsilon)
        return native.hbfp_linear_decode(x, self.codes, self.exps, self.mant_bits, x_mant_bits, epsilon)

    def linear(self, x, bias=None, x_mant_bits=7, epsilon=1e-8, decode_tokens=64):
        """F.linear(Q_in(x), W_packed, bias) for any number of tokens, W = self [N, K] (the forward of a BFPLinear whose
        weight is held packed; activations HBFP(x_mant_bits + 1), block 64, round-half-even).
        Up to `decode_tokens` tokens: chunks of 16 through the integer block-dot-product kernel (linear_decode), which
        never materialises the weight.  More tokens (prefill): when the activation mantissas fit 4 bits (x_mant_bits <= 4)
        and K % 256 == 0, the block-scaled matrix instruction multiplies the two HBFP operands as they are
        (native.hbfp_linear_mx8); otherwise the weight is decoded to its dtype once per call (bfpq_dequantize, a
        streaming pass) and multiplied by the library GEMM."""
        K = self.shape[-1]
        lead = x.shape[:-1]
        x2 = x.reshape(-1, K)
        T = x2.shape[0]
        if T == 0:
            return x.new_zeros(lead + (self.shape[0],))
        if T <= decode_tokens and self.code_bits == 4 and self.block_size == 64 and self.shape[0] % 16 == 0 and K % 256 == 0:
            out = self.linear_decode(x2, x_mant_bits, epsilon) if (T <= 16 or self._tiled_ok()) else \
                torch.cat([self.linear_decode(x2[i:i + 16], x_mant_bits, epsilon) for i in range(0, T, 16)], 0)
        elif self.code_bits in (4, 8) and native.hbfp_linear_mx8_ok(T, self.shape[0], K, self.mant_bits, x_mant_bits, self.block_size):
            # prefill: both operands as e4m3 mantissas + E8M0 block scales on the block-scaled matrix instruction
            # (exact block dot products, fp32 across blocks); the weight's image is made once per device
            out = native.hbfp_linear_mx8(x2, *self._mx8_image(), x_mant_bits, epsilon, bias=bias)
            return out.view(lead + (self.shape[0],))
        else:
            xq = _quantize_nm_ref_dtype(x2, 64, x_mant_bits, epsilon, rounding_modes.DETERM)
            out = F.linear(xq, self.dequantize().to(x.dtype))
        if bias is not None:
            out = out + bias
        return out.view(lead + (self.shape[0],))

    def _mx8_image(self):
        if getattr(self, "_mx8", None) is None or self._mx8[0].device != self.codes.device:
            self._mx8 = native.mx8_from_hbfp(self.codes, self.exps, self.shape[-1], self.mant_bits, self.code_bits)
        return self._mx8

    def save(self, path, compact24=False):
        """safetensors file: tensors `codes`, `exps`; the rest as string metadata.  compact24=True stores a 2:4-sparse
        4-bit tensor as 2 nibbles + 2 x 2-bit positions per group of 4 (`vals`, `idx`: 0.375 B/element instead of 0.5)."""
        from safetensors.torch import save_file
        meta = dict(format="hbfp-packed-v1", shape=",".join(str(d) for d in self.shape), dtype=str(self.dtype).replace("torch.", ""),
                    mant_bits=str(self.mant_bits), block_size=str(self.block_size), code_bits=str(self.code_bits))
        if compact24:
            assert self.code_bits == 4 and self.codes.numel() % 4 == 0
            vals, idx = native.compact24(self.codes)
            meta["layout"] = "compact24"
            meta["codes_shape"] = ",".join(str(d) for d in self.codes.shape)
            save_file({"vals": vals.cpu(), "idx": idx.cpu(), "exps": self.exps.cpu().contiguous()}, path, metadata=meta)
            return
        save_file({"codes": self.codes.cpu().contiguous(), "exps": self.exps.cpu().contiguous()}, path, metadata=meta)

    @classmethod
    def load(cls, path, device="cuda"):
        from safetensors import safe_open
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata()
            assert meta.get("format") == "hbfp-packed-v1", meta
            exps = f.get_tensor("exps")
            if meta.get("layout") == "compact24":
                cshape = tuple(int(d) for d in meta["codes_shape"].split(","))
                codes = native.expand24(f.get_tensor("vals").to(device), f.get_tensor("idx").to(device)).view(cshape)
            else:
                codes = f.get_tensor("codes")
        shape = tuple(int(d) for d in meta["shape"].split(",")) if meta["shape"] else ()
        return cls(codes.to(device), exps.to(device), shape, getattr(torch, meta["dtype"]), int(meta["mant_bits"]),
                   int(meta["block_size"]), int(meta["code_bits"]))


# ---- module / functional wrappers (reference: bfp_ops.py:151-287) -----------------------------
def MxM_pre_processing(x, w, transpose, **bfp_args):
    """reference: bfp_ops.py:151-155 -- quantize both GEMM operands; with transpose the second operand is
    quantized along its contraction dim (blocks must run along K) and handed back in its original layout"""
    xq = float_to_bfp_blocked(x, **bfp_args, identifier='in')
    if transpose == True:  # noqa: E712
        wq = float_to_bfp_blocked(w.transpose(-1, -2), **bfp_args, identifier='w').transpose(-1, -2)
    else:
        wq = float_to_bfp_blocked(w, **bfp_args, identifier='w')
    return xq, wq


def _get_op_name(name, epsilon, mant_bits, rounding_mode, **kwargs):
    """reference: bfp_ops.py:157-158"""
    return '%s_BFP_%s_%d' % (name, rounding_mode, mant_bits)


FUSE_OPERAND_PAIR = True           # BFPLinear / BFPConv2d forward: quantize activation and weight in one launch when possible


_CACHE_KEYS = ('mant_bits', 'epsilon', 'rounding_mode', 'block_size', 'num_format', 'weight_mant_bits', 'w_sparsity',
               'sparsity_frac', 'N', 'M', 'sparsity_num_format', 'first', 'sparsity_mode')


class WeightCache:
    """Opt-in, INFERENCE-ONLY cache of the quantized (and sparsified) weight operand.  The reference re-runs the whole
    path on the unchanged weight in every forward (bfp_ops.py:151-166, 7 Linear layers per LLaMA block); with the cache an
    inference loop quantizes each weight once.  Off by default (reference semantics).
    Valid while (storage pointer, in-place version counter, dtype, device, shape, the configuration values that shape
    the weight operand) are unchanged.  It is BYPASSED whenever a stale entry could go unnoticed: while gradients are being
    recorded for the weight, or the owning module is in training mode (optimizers and EMA code update `p.data` in place,
    which does not bump the version counter), and for stochastic rounding.  After such an update in eval mode call
    invalidate()."""

    def __init__(self, module=None, matrix_unit=False):
        import weakref
        self.key = None
        self.value = None
        self.hits = 0
        self.misses = 0
        self._module = weakref.ref(module) if module is not None else None
        # matrix_unit: F.linear on the two fake-quantised operands is replaced by the block-scaled matrix instruction on their e4m3 + E8M0
        # images where the configuration allows it (same products and block sums, fp32 across blocks; see PackedBFP.linear); the weight's
        # image is cached like the fake-quantised weight, under the same key
        self.matrix_unit = bool(matrix_unit)
        self.image_key = None
        self.image = None
        self.mx_calls = 0

    @staticmethod
    def _key(w, bfp_args):
        return (w.data_ptr(), native.tensor_version(w), w.dtype, w.device, tuple(w.shape), tuple(bfp_args.get(k) for k in _CACHE_KEYS))

    def usable(self, w, bfp_args):
        if bfp_args.get('rounding_mode') != rounding_modes.DETERM:
            return False
        if w.is_inference() or native.tensor_version(w) is None:
            return False                                     # no version counter: an in-place update could not be seen
        if torch.is_grad_enabled() and w.requires_grad:
            return False
        m = self._module() if self._module is not None else None
        return not (m is not None and m.training)

    def lookup(self, w, bfp_args):
        if not self.usable(w, bfp_args):
            return None
        if self.key == self._key(w, bfp_args):
            self.hits += 1
            return self.value.view_as(self.value)            # a fresh alias: the stored tensor never enters an autograd graph
        return None

    def store(self, w, wq, bfp_args=None):
        bfp_args = bfp_args or {}
        if bfp_args and not self.usable(w, bfp_args):
            return
        self.key = self._key(w, bfp_args)
        self.value = wq.detach()
        self.misses += 1

    def invalidate(self):
        self.key = None
        self.value = None
        self.image_key = None
        self.image = None
        native.forget_shared_images()

    def weight_image(self, w, bfp_args):
        """e4m3 + E8M0 image of Q_w(w) for the matrix unit, made from the packed codes of the weight (not from the fake-quantised
        tensor: HBFP quantization is not idempotent on a block whose maximum rounded down to a power of two)"""
        key = self._key(w, bfp_args)
        if self.image_key != key:
            a = bfp_args
            sp = a['w_sparsity'] == True  # noqa: E712
            cb = 4 if a['mant_bits'] <= 3 else 8
            if sp and a['sparsity_mode'] == 'unstructured':
                pk = PackedBFP.quantize_unstructured(w.detach(), a['mant_bits'], 64, a['sparsity_frac'], a['epsilon'], cb)
                codes, exps = pk.codes, pk.exps
            else:
                codes, exps = float_to_bfp_packed(w.detach(), a['mant_bits'], 64, a['epsilon'], a['N'] if sp else 0, a['M'] if sp else 0, a['first'], cb)
            self.image = native.mx8_from_hbfp(codes, exps, w.shape[-1], a['mant_bits'], cb)
            self.image_key = key
        return self.image


def _pair_in_one_launch(x, w, bfp_args):
    """activation + weight of one Linear through ONE launch (native.FastQuant.many) when both take the drop-in kernel with
    the same plan: 'bfp' format, round-half-even, dense or structured pruning per operand; else None"""
    a = bfp_args
    if (a['sparsity_num_format'] != 'bfp' or a['rounding_mode'] != rounding_modes.DETERM or a['num_format'] != 'bfp'
            or not a['block_size'] > 0                        # (the asserting path says what the reference says, bfp_ops.py:130)
            or x.dtype != w.dtype or x.device != w.device or x.device.type != 'cuda' or torch.compiler.is_compiling()):
        return None
    sp_in, sp_w = a['in_sparsity'] == True, a['w_sparsity'] == True  # noqa: E712
    if (sp_in or sp_w):
        if a['sparsity_mode'] != 'structured' or not (0 < a['N'] <= a['M']):
            return None
    nm = (sp_in or sp_w) and a['N'] < a['M']
    f = _fast_quant(a['block_size'], a['mant_bits'], a['epsilon'], a['N'] if nm else 0, a['M'] if nm else 0, a['first'] == 's')
    xq, wq = f.many([x, w], [sp_in, sp_w])
    return xq.view(x.shape), wq.view(w.shape)


def _quantize_operands(x, w, transpose, bfp_args, cache):
    """(Q_in(x), Q_w(w)) -- reference MxM_pre_processing (bfp_ops.py:151-155), with the optional weight cache"""
    if cache is None or not cache.usable(w, bfp_args):
        if transpose != True and FUSE_OPERAND_PAIR:  # noqa: E712
            pair = _pair_in_one_launch(x, w, bfp_args)
            if pair is not None:
                return pair
        return MxM_pre_processing(x, w, transpose, **bfp_args)
    xq = float_to_bfp_blocked(x, **bfp_args, identifier='in')
    wq = cache.lookup(w, bfp_args)
    if wq is None:
        wq = _quantize_weight_operand(w, transpose, bfp_args)
        cache.store(w, wq, bfp_args)
    return xq, wq


class _OperandQuantizer(torch.autograd.Function):
    """forward: (x, w) -> (Q_in(x), Q_w(w)); backward: straight-through (reference NewOpIn, bfp_ops.py:163-170)"""

    @staticmethod
    def forward(ctx, x, w, transpose, bfp_args, cache):
        return _quantize_operands(x, w, transpose, bfp_args, cache)

    @staticmethod
    def backward(ctx, grad_x, grad_w):
        return grad_x, grad_w, None, None, None


def _quantize_weight_operand(w, transpose, bfp_args):
    """the second operand of MxM_pre_processing alone"""
    if transpose == True:  # noqa: E712
        return float_to_bfp_blocked(w.transpose(-1, -2), **bfp_args, identifier='w').transpose(-1, -2)
    return float_to_bfp_blocked(w, **bfp_args, identifier='w')


class _GradQuantizer(torch.autograd.Function):
    """forward: identity; backward: Q_grad(dL/dout) (reference NewOpOut, bfp_ops.py:175-182)"""

    @staticmethod
    def forward(ctx, out, bfp_args):
        ctx.bfp_args = bfp_args
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad_out):
        return float_to_bfp_blocked(grad_out, **ctx.bfp_args, identifier='grad'), None


def _matrix_unit_ok(x, w, a, cache):
    """the cached Linear forward can run on the block-scaled matrix instruction: HBFP with block 64 and <= 4 mantissa bits (exact in e4m3),
    round-half-even, dense activations, dense or N:M weights, a 2-D weight with K % 256 == 0, enough tokens to fill a tile row"""
    if (a['num_format'] != 'bfp' or a['sparsity_num_format'] != 'bfp' or a['block_size'] != 64 or not (1 <= a['mant_bits'] <= 4)
            or a['rounding_mode'] != rounding_modes.DETERM or a['in_sparsity'] == True or w.dim() != 2 or x.device.type != 'cuda'  # noqa: E712
            or torch.compiler.is_compiling() or not cache.usable(w, a)):
        return False
    if a['w_sparsity'] == True:  # noqa: E712
        if a['sparsity_mode'] == 'unstructured':
            if a['first'] != 's' or not (0 < a['sparsity_frac'] <= 1) or w.shape[-1] % 64:       # the fused prune + quantize launch writes the codes
                return False
        elif a['sparsity_mode'] != 'structured' or not (0 < a['N'] <= a['M']) or a['M'] not in (2, 4, 8):
            return False
    K = x.shape[-1]
    T = x.numel() // K if K else 0
    return T >= 32 and native.hbfp_linear_mx8_ok(T, w.shape[0], K, a['mant_bits'], a['mant_bits'], 64)


def _gen_bfp_op(op, name, bfp_args, transpose=False, cache=None):
    """reference: bfp_ops.py:160-192 -- wraps `op(x, w, ...)` so that both operands are BFP-quantized
    (and sparsified) on the way in and the output gradient on the way back.  cache: optional WeightCache."""
    def bfp_op(x, w, *args, **kwargs):
        if not torch.is_grad_enabled():                 # inference: same values, no autograd nodes to build
            if cache is not None and cache.matrix_unit and op is F.linear and _matrix_unit_ok(x, w, bfp_args, cache):
                cache.mx_calls += 1
                bias = args[0] if args else kwargs.get('bias')
                w8, wsc = cache.weight_image(w, bfp_args)
                return native.hbfp_linear_mx8(x, w8, wsc, bfp_args['mant_bits'], bfp_args['epsilon'], bias=bias)
            xq, wq = _quantize_operands(x, w, transpose, bfp_args, cache)
            return op(xq, wq, *args, **kwargs)
        xq, wq = _OperandQuantizer.apply(x, w, transpose, bfp_args, cache)
        return _GradQuantizer.apply(op(xq, wq, *args, **kwargs), bfp_args)

    bfp_op.__name__ = _get_op_name(name, **bfp_args)
    bfp_op.weight_cache = cache
    return bfp_op


def _get_bfp_op(op, name, bfp_args, transpose=False, cache=None):
    """reference: bfp_ops.py:194-200 (its cache dict is a local, so there too every layer gets its own op)"""
    return _gen_bfp_op(op, name, bfp_args, transpose, cache)


_BFP_ARG_DEFAULTS = dict(num_format='fp32', sparsity_num_format='fp32', rounding_mode='stoc', epsilon=1e-8,
                         mant_bits=0, block_size=0, weight_mant_bits=0, in_sparsity=False, w_sparsity=False,
                         grad_sparsity=False, N=0, M=0, first='s', sparsity_mode='unstructured', sparsity_frac=0,
                         mx_w_elem_format='', mx_a_elem_format='', bfloat=16, scale_bits=8, device='cpu')


def unpack_bfp_args(kwargs):
    """reference: bfp_ops.py:202-231 -- moves the 20 known keys out of `kwargs` (which is mutated) into a
    new dict, filling defaults; unknown keys stay behind in `kwargs`."""
    return {key: kwargs.pop(key, default) for key, default in _BFP_ARG_DEFAULTS.items()}


def F_linear_bfp(**kwargs):
    """reference: bfp_ops.py:233-238"""
    bfp_args = unpack_bfp_args(kwargs)
    return _get_bfp_op(F.linear, 'linear', bfp_args) if bfp_args['num_format'] == 'bfp' else F.linear


def F_matmul_bfp(**kwargs):
    """reference: bfp_ops.py:240-245"""
    bfp_args = unpack_bfp_args(kwargs)
    return _get_bfp_op(torch.matmul, 'matmul', bfp_args, True) if bfp_args['num_format'] == 'bfp' else torch.matmul


class _BFPModule:
    """shared by BFPLinear / BFPConv2d: config capture before nn.Module.__init__, format dispatch in forward"""

    def _bfp_setup(self, kwargs, functional, op_name):
        self.bfp_args = unpack_bfp_args(kwargs)
        self._functional = functional
        self._op_name = op_name

    def _bfp_finish(self):
        self.num_format = self.bfp_args['num_format']
        return _get_bfp_op(self._functional, self._op_name, self.bfp_args)

    def enable_weight_cache(self, enabled=True, matrix_unit=False):
        """opt in to (or out of) caching the quantized weight across forwards; see WeightCache.  matrix_unit=True additionally
        runs the Linear itself on the block-scaled matrix instruction where the configuration allows (HBFP4-class, block 64,
        >= 32 tokens): the same block products, summed exactly inside a block and in fp32 across blocks instead of by the bf16
        library GEMM -- results agree with F.linear on the fake-quantised operands up to that summation order."""
        cache = WeightCache(self, matrix_unit=matrix_unit) if enabled else None
        self._weight_cache = cache                               # (patch.prime_weight_caches fills it for a whole model in one list call)
        op = _get_bfp_op(self._functional, self._op_name, self.bfp_args, cache=cache)
        if hasattr(self, 'linear_op'):
            self.linear_op = op
        else:
            self.conv_op = op
        return self

    def _bfp_forward(self, bfp_op, *operands):
        if self.num_format == 'fp32':
            return self._functional(*operands)
        if self.num_format == 'bfp':
            return bfp_op(*operands)
        raise NotImplementedError('NumFormat not implemented')


class BFPConv2d(_BFPModule, torch.nn.Conv2d):
    """reference: bfp_ops.py:247-268 (no extra parameters or buffers: state_dict == nn.Conv2d's)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, bias=True, **kwargs):
        self._bfp_setup(kwargs, F.conv2d, 'Conv2d')
        torch.nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.conv_op = self._bfp_finish()

    def forward(self, input):
        return self._bfp_forward(self.conv_op, input, self.weight, self.bias, self.stride, self.padding,
                                 self.dilation, self.groups)


class BFPLinear(_BFPModule, torch.nn.Linear):
    """reference: bfp_ops.py:270-287 (no extra parameters or buffers: state_dict == nn.Linear's)"""

    def __init__(self, in_features, out_features, bias=True, **kwargs):
        self._bfp_setup(kwargs, F.linear, 'linear')
        torch.nn.Linear.__init__(self, in_features, out_features, bias)
        self.linear_op = self._bfp_finish()

    def forward(self, input):
        return self._bfp_forward(self.linear_op, input, self.weight, self.bias)
