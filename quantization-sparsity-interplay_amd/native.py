"""ctypes binding of libbfpq.so (include/bfpq.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns device memory and the HIP stream; every byte of arithmetic on the
hot path happens in the HIP kernels of csrc/bfpq_kernels.hip.  There is NO CPU fallback: if the
library is missing, or a tensor is not on a ROCm device, the calls raise.
"""
import ctypes
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libbfpq.so"
_lock = threading.Lock()
_lib = None

F32, F16, BF16 = 0, 1, 2
DTYPE_CODE = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
BFPQ_E_UNSUPPORTED = -2

EXP_WIN_ENTRIES = 320
NM4_LUT_ENTRIES = 729
NM8_LUT_ENTRIES = 1 << 24
USE_NM8_TABLE = True               # False: N:8 groups with straddling ties replay nth_element in the kernel (tests)
SELECT_STATE_BYTES = 80
SELECT_HIST_ENTRIES = 32768 + 256
SELECT_HIST_COPIES = 8


class NativeUnavailable(RuntimeError):
    """libbfpq.so is not built / not loadable, or the tensor is not on a ROCm device."""


def lib_path():
    return os.environ.get("BFPQ_LIB") or os.path.join(_HERE, _LIB_NAME)     # (BFPQ_LIB: an instrumented build, tools_dev/)


def build_library(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return lib_path()


def load_library():
    """dlopen libbfpq.so and declare the prototypes of include/bfpq.h.  Raises NativeUnavailable."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise NativeUnavailable(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                    f"or `make -C {os.path.join(_HERE, 'csrc')}`")
        try:
            L = ctypes.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise NativeUnavailable(f"cannot load {path}: {e}") from e
        vp, i64, i32, u64, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_double
        L.bfpq_version.restype = i32
        L.bfpq_tune.argtypes = [i32, i32]
        L.bfpq_tune.restype = i32
        L.bfpq_error_string.restype = ctypes.c_char_p
        L.bfpq_error_string.argtypes = [i32]
        L.bfpq_exp_window_host.argtypes = [i32, vp]
        L.bfpq_nm4_lut_host.argtypes = [i32, vp]
        L.bfpq_nm8_lut_host.argtypes = [i32, vp]
        L.bfpq_compact24.argtypes = [vp, vp, vp, i64, vp, vp]
        L.bfpq_expand24.argtypes = [vp, vp, vp, i64, vp]
        L.bfpq_nm_prune_mask_host.argtypes = [vp, i32, i32]
        L.bfpq_nm_prune_mask_host.restype = u64
        L.bfpq_quantize_nm.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, dbl, i32, i32, i32, i32, u64, vp, vp, vp, vp]
        L.bfpq_fake_quantize.argtypes = [vp, vp, vp, i64, i64, vp]
        L.bfpq_fake_quantize_batched.argtypes = [vp, vp, i32, vp]
        L.bfpq_fake_quantize_list.argtypes = [vp, vp, i32, vp, vp, i32]
        L.bfpq_is_fused.argtypes = [i64, i64, i32, i32, i32, i32]
        L.bfpq_nm_sparsify.argtypes = [vp, vp, i64, i64, i32, i32, i32, vp, vp]
        L.bfpq_select_passes.argtypes = [i32]
        L.bfpq_select_ws_bytes.argtypes = []
        L.bfpq_select_ws_bytes.restype = i64
        L.bfpq_select.argtypes = [vp, i64, i32, i64, vp, vp]
        L.bfpq_prune_quantize.argtypes = [vp, vp, i64, i64, i32, i32, i32, dbl, i64, vp, vp, vp]
        L.bfpq_prune_quantize_batched.argtypes = [vp, i32, i32, i32, i32, dbl, vp, vp, i32, vp, vp]
        L.bfpq_prune_quantize_list.argtypes = [vp, i32, i32, i32, i32, dbl, vp, vp, i32, vp, vp, i32]
        L.bfpq_select_hist.argtypes = [vp, i64, i32, i32, i64, i64, vp, vp, vp]
        L.bfpq_select_resolve.argtypes = [vp, i64, i32, i32, i64, vp, i32, i32, vp, vp, vp]
        L.bfpq_select_resolve_ex.argtypes = [vp, i64, i32, i32, i64, vp, i32, i32, i32, i64, vp, vp, vp]
        L.bfpq_threshold_apply.argtypes = [vp, vp, i64, i32, vp, vp]
        L.bfpq_select_reset.argtypes = [vp, vp]
        L.bfpq_quantize_threshold.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, dbl, i32, u64, vp, vp, vp, vp]
        L.bfpq_int_workspace_elems.argtypes = [i64]
        L.bfpq_int_workspace_elems.restype = i64
        L.bfpq_int_quantize.argtypes = [vp, vp, i64, i64, i64, i32, i32, vp, vp]
        L.bfpq_dequantize.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, i32, vp]
        L.bfpq_hbfp_linear_slices.argtypes = [i64, i64]
        L.bfpq_hbfp_linear_decode.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, vp]
        L.bfpq_hbfp_linear_tiled_ok.argtypes = [i64, i64]
        L.bfpq_hbfp_linear_decode_tiled.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, vp]
        L.bfpq_mx8_from_hbfp.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, vp]
        L.bfpq_quantize_mx8.argtypes = [vp, vp, vp, i64, i64, i32, i32, dbl, vp, vp]
        L.bfpq_hbfp_linear_mx8_ok.argtypes = [i64, i64, i64]
        L.bfpq_hbfp_linear_mx8.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]
        L.bfpq_hbfp_linear_mx8_parts.argtypes = [i64, i64, i64]
        L.bfpq_hbfp_linear_mx8_splitk.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i64, i64, i64, i32, vp]
        for name in ("bfpq_exp_window_host", "bfpq_nm4_lut_host", "bfpq_nm8_lut_host", "bfpq_compact24", "bfpq_expand24", "bfpq_quantize_nm", "bfpq_fake_quantize", "bfpq_fake_quantize_batched", "bfpq_fake_quantize_list", "bfpq_is_fused", "bfpq_nm_sparsify",
                     "bfpq_select_passes", "bfpq_select", "bfpq_select_hist", "bfpq_select_resolve", "bfpq_select_resolve_ex", "bfpq_select_reset",
                     "bfpq_prune_quantize", "bfpq_prune_quantize_batched", "bfpq_prune_quantize_list",
                     "bfpq_threshold_apply", "bfpq_quantize_threshold", "bfpq_int_quantize", "bfpq_dequantize", "bfpq_hbfp_linear_slices",
                     "bfpq_hbfp_linear_decode", "bfpq_hbfp_linear_tiled_ok", "bfpq_hbfp_linear_decode_tiled",
                     "bfpq_mx8_from_hbfp", "bfpq_quantize_mx8", "bfpq_hbfp_linear_mx8_ok", "bfpq_hbfp_linear_mx8",
                     "bfpq_hbfp_linear_mx8_parts", "bfpq_hbfp_linear_mx8_splitk"):
            getattr(L, name).restype = i32
        _lib = L
        return _lib


EXPORTED_SYMBOLS = ("bfpq_hbfp_linear_mx8_parts", "bfpq_hbfp_linear_mx8_splitk", "bfpq_quantize_mx8", "bfpq_mx8_from_hbfp", "bfpq_hbfp_linear_mx8_ok", "bfpq_hbfp_linear_mx8", "bfpq_hbfp_linear_tiled_ok", "bfpq_hbfp_linear_decode_tiled", "bfpq_hbfp_linear_slices", "bfpq_hbfp_linear_decode", "bfpq_dequantize", "bfpq_tune", "bfpq_int_workspace_elems", "bfpq_int_quantize", "bfpq_select_ws_bytes", "bfpq_version", "bfpq_error_string", "bfpq_exp_window_host", "bfpq_nm4_lut_host", "bfpq_nm8_lut_host", "bfpq_compact24", "bfpq_expand24",
                    "bfpq_nm_prune_mask_host", "bfpq_quantize_nm", "bfpq_fake_quantize", "bfpq_fake_quantize_batched", "bfpq_fake_quantize_list", "bfpq_is_fused", "bfpq_nm_sparsify",
                    "bfpq_select_passes", "bfpq_select", "bfpq_select_hist", "bfpq_select_resolve", "bfpq_select_resolve_ex", "bfpq_select_reset",
                    "bfpq_prune_quantize", "bfpq_prune_quantize_batched", "bfpq_prune_quantize_list",
                    "bfpq_threshold_apply", "bfpq_quantize_threshold")


def check(rc, what):
    if rc != 0:
        msg = load_library().bfpq_error_string(rc).decode()
        raise RuntimeError(f"{what}: {msg} (code {rc})")


# ---- host-side tables -----------------------------------------------------------------------
def exp_window_host(dtype):
    buf = (ctypes.c_uint8 * EXP_WIN_ENTRIES)()
    check(load_library().bfpq_exp_window_host(DTYPE_CODE[dtype], ctypes.addressof(buf)), "bfpq_exp_window_host")
    return bytes(buf)


def nm4_lut_host(N):
    buf = (ctypes.c_uint8 * NM4_LUT_ENTRIES)()
    check(load_library().bfpq_nm4_lut_host(int(N), ctypes.addressof(buf)), "bfpq_nm4_lut_host")
    return bytes(buf)


def nm_prune_mask_host(keys, N, M):
    arr = (ctypes.c_uint32 * M)(*[int(k) for k in keys])
    return int(load_library().bfpq_nm_prune_mask_host(ctypes.addressof(arr), int(N), int(M)))


_table_cache = {}


def _device_table(kind, key, device, maker):
    ck = (kind, key, device.index if device.index is not None else torch.cuda.current_device())
    t = _table_cache.get(ck)
    if t is None:
        t = torch.frombuffer(bytearray(maker()), dtype=torch.uint8).to(device)
        _table_cache[ck] = t
    return t


def exp_window_dev(dtype, device):
    return _device_table("win", dtype, device, lambda: exp_window_host(dtype))


def nm4_lut_dev(N, device):
    return _device_table("lut", int(N), device, lambda: nm4_lut_host(N))


def nm8_lut_host(N):
    buf = (ctypes.c_uint8 * NM8_LUT_ENTRIES)()
    check(load_library().bfpq_nm8_lut_host(int(N), ctypes.addressof(buf)), "bfpq_nm8_lut_host")
    return buf                                            # (buffer protocol: _device_table copies it once)


def nm8_lut_dev(N, device):
    """16 MiB rank table for N:8 (built on first use per (N, device), ~0.1 s)"""
    return _device_table("lut8", int(N), device, lambda: nm8_lut_host(N))


# ---- tensor plumbing ------------------------------------------------------------------------
def require_device_tensor(t, what="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise NativeUnavailable(f"{what} is on {t.device}; the BFP engine runs on a ROCm device only (no CPU fallback)")
    if t.dtype not in DTYPE_CODE:
        raise TypeError(f"{what} has dtype {t.dtype}; supported: float32, float16, bfloat16")


def rows_cols(t):
    cols = t.shape[-1] if t.dim() > 0 else 1
    rows = (t.numel() // cols) if cols else 0
    return rows, cols


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def quantize_nm(t, block_size, mant_bits, epsilon, N=0, M=0, sparsify_first=True, want_deq=True, code_bits=0,
                want_exp=False, stoch_seed=0, out=None, codes_out=None, exps_out=None):
    """Launch bfpq_quantize_nm on t's device/stream.  Returns (deq | None, codes | None, exps | None)."""
    require_device_tensor(t)
    L = load_library()
    src = t.contiguous()
    rows, cols = rows_cols(src)
    dev = src.device
    if src.numel() == 0:                       # nothing to launch; shapes as below
        nb = (cols + block_size - 1) // block_size if block_size > 0 else 0
        return (torch.empty_like(src) if want_deq else None,
                torch.empty((rows, (cols + 1) // 2 if code_bits == 4 else cols), dtype=torch.uint8 if code_bits == 4 else
                            (torch.int8 if code_bits == 8 else torch.int16), device=dev) if code_bits else None,
                torch.empty((rows, nb), dtype=torch.int8, device=dev) if (want_exp and block_size > 0) else None)
    with torch.cuda.device(dev):
        deq = None
        if want_deq:
            deq = out if out is not None else torch.empty_like(src)
        codes = exps = None
        if code_bits and codes_out is not None:
            codes = codes_out                                  # caller-owned, contiguous, right shape
        elif code_bits:
            if code_bits == 4:
                codes = torch.empty((rows, (cols + 1) // 2), dtype=torch.uint8, device=dev)
            elif code_bits == 8:
                codes = torch.empty((rows, cols), dtype=torch.int8, device=dev)
            elif code_bits == 32:
                codes = torch.empty((rows, cols), dtype=torch.float32, device=dev)
            else:
                codes = torch.empty((rows, cols), dtype=torch.int16, device=dev)
        if want_exp and block_size > 0 and exps_out is not None:
            exps = exps_out
        elif want_exp and block_size > 0:
            exps = torch.empty((rows, (cols + block_size - 1) // block_size), dtype=torch.int8, device=dev)
        win = exp_window_dev(src.dtype, dev) if block_size > 0 else None
        lut = nm4_lut_dev(N, dev) if M == 4 else (nm8_lut_dev(N, dev) if (M == 8 and USE_NM8_TABLE) else None)
        fused = L.bfpq_is_fused(rows, cols, DTYPE_CODE[src.dtype], int(block_size), int(N), int(M))
        scratch = None
        if not fused and deq is None and M > 0 and block_size > 0:
            scratch = torch.empty_like(src)
        rc = L.bfpq_quantize_nm(_ptr(src), _ptr(deq), _ptr(codes), _ptr(exps), rows, cols, DTYPE_CODE[src.dtype],
                                int(block_size), int(mant_bits), float(epsilon), int(N), int(M),
                                1 if sparsify_first else 0, int(code_bits), int(stoch_seed),
                                _ptr(win), _ptr(lut), _ptr(scratch), _stream(src))
        check(rc, "bfpq_quantize_nm")
    return deq, codes, exps


class _Plan(ctypes.Structure):
    """include/bfpq.h: bfpq_plan"""
    _fields_ = [("dtype", ctypes.c_int), ("block_size", ctypes.c_int), ("mant_bits", ctypes.c_int), ("N", ctypes.c_int),
                ("M", ctypes.c_int), ("sparsify_first", ctypes.c_int), ("epsilon", ctypes.c_double),
                ("exp_win_dev", ctypes.c_void_p), ("nm_lut_dev", ctypes.c_void_p)]


class _TensorDesc(ctypes.Structure):
    """include/bfpq.h: bfpq_tensor_desc"""
    _fields_ = [("in_dev", ctypes.c_void_p), ("out_dev", ctypes.c_void_p), ("rows", ctypes.c_int64), ("cols", ctypes.c_int64),
                ("apply_nm", ctypes.c_int), ("reserved", ctypes.c_int)]


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class FastQuant:
    """The drop-in call (round-half-even, dequantised tensor out) with everything constant bound once: per call it costs
    one output allocation and one six-argument ctypes call (bfpq_fake_quantize).  This is what the module wrappers issue
    per tensor; native.quantize_nm stays the general entry point."""

    __slots__ = ("block_size", "mant_bits", "epsilon", "N", "M", "sparsify_first", "_plans", "_fn")

    def __init__(self, block_size, mant_bits, epsilon, N=0, M=0, sparsify_first=True):
        self.block_size, self.mant_bits, self.epsilon = int(block_size), int(mant_bits), float(epsilon)
        self.N, self.M = (int(N), int(M)) if 0 < N < M else (0, 0)
        self.sparsify_first = bool(sparsify_first)
        self._plans = {}
        self._fn = load_library().bfpq_fake_quantize

    def _plan(self, dtype, dev):
        key = (dtype, dev.index)
        p = self._plans.get(key)
        if p is None:
            win = exp_window_dev(dtype, dev) if self.block_size > 0 else None
            lut = nm4_lut_dev(self.N, dev) if self.M == 4 else (nm8_lut_dev(self.N, dev) if (self.M == 8 and USE_NM8_TABLE) else None)
            plan = _Plan(DTYPE_CODE[dtype], self.block_size, self.mant_bits, self.N, self.M, 1 if self.sparsify_first else 0,
                         self.epsilon, win.data_ptr() if win is not None else None, lut.data_ptr() if lut is not None else None)
            p = self._plans[key] = (plan, ctypes.addressof(plan), win, lut)        # (the tensors are kept alive with the plan)
        return p

    def __call__(self, t, out=None):
        if t.device.type != "cuda" or t.dtype not in DTYPE_CODE:
            require_device_tensor(t)
        if self.block_size == 0 and self.M == 0:
            return t
        src = t if t.is_contiguous() else t.contiguous()
        dst = torch.empty_like(src) if out is None else out
        n = src.numel()
        if n == 0:
            return dst
        dev = src.device
        cols = src.shape[-1] if src.dim() else 1
        plan = self._plan(src.dtype, dev)
        if torch.cuda.current_device() != dev.index:
            with torch.cuda.device(dev):
                rc = self._fn(plan[1], src.data_ptr(), dst.data_ptr(), n // cols, cols, torch.cuda.current_stream(dev).cuda_stream)
        else:
            stream = _raw_stream(dev.index) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream
            rc = self._fn(plan[1], src.data_ptr(), dst.data_ptr(), n // cols, cols, stream)
        if rc:
            check(rc, "bfpq_fake_quantize")
        return dst

    def prepare(self, tensors, apply_nm=None, outs=None):
        """PreparedList for these tensors: outputs allocated and descriptors built once, .run() re-issues the launches.
        For tensors whose storage stays put (the weights of a model quantized every forward)."""
        return PreparedList(self, tensors, apply_nm, outs)

    def many(self, tensors, apply_nm=None, outs=None):
        """the same for a LIST of tensors of one dtype on one device, in as few launches as possible
        (bfpq_fake_quantize_batched: up to 64 tensors per launch).  apply_nm: per tensor, whether this plan's N:M pruning
        applies to it (default: to all).  Returns the list of results (a tensor the plan leaves untouched comes back itself)."""
        n = len(tensors)
        if n == 0:
            return []
        flags = [True] * n if apply_nm is None else [bool(f) for f in apply_nm]
        t0 = tensors[0]
        require_device_tensor(t0)
        dev, dt = t0.device, t0.dtype
        res = [None] * n
        descs = (_TensorDesc * n)()
        keep = []                                              # contiguous copies live until the call returns (stream-ordered after that)
        k = 0
        for i, t in enumerate(tensors):
            if t.device != dev or t.dtype != dt:
                raise ValueError("FastQuant.many: the tensors of one call share device and dtype")
            if self.block_size == 0 and not (flags[i] and self.M > 0):
                res[i] = t                                     # identity for this tensor
                continue
            src = t if t.is_contiguous() else t.contiguous()
            dst = torch.empty_like(src) if outs is None or outs[i] is None else outs[i]
            res[i] = dst
            if src.numel() == 0:
                continue
            cols = src.shape[-1] if src.dim() else 1
            d = descs[k]
            d.in_dev, d.out_dev, d.rows, d.cols, d.apply_nm = src.data_ptr(), dst.data_ptr(), src.numel() // cols, cols, 1 if flags[i] else 0
            res[i] = dst
            k += 1
            keep.append(src)
        if k:
            plan = self._plan(dt, dev)
            with torch.cuda.device(dev):
                aux = aux_stream_array(dev, 1) if k > 1 else None
                rc = load_library().bfpq_fake_quantize_list(plan[1], ctypes.addressof(descs), k, torch.cuda.current_stream(dev).cuda_stream,
                                                            ctypes.addressof(aux) if aux is not None else None, 1 if aux is not None else 0)
            if rc:
                check(rc, "bfpq_fake_quantize_list")
        return res


class PreparedList:
    """A list of tensors bound to one FastQuant plan: input pointers, output tensors and the descriptor array are set up
    once; run() is ONE ctypes call (bfpq_fake_quantize_batched) whatever the number of tensors.  The inputs are expected to
    keep their storage between runs (a model's weights); run() checks every bound pointer against the tensor's current one
    and re-binds a tensor whose storage moved (`model.to(...)`, `p.data = ...`, `load_state_dict(assign=True)`) -- a changed
    dtype, device or element count raises instead of reading freed memory.  Results are written in place into .outputs."""

    def __init__(self, fq, tensors, apply_nm=None, outs=None):
        tensors = list(tensors)
        n = len(tensors)
        flags = [True] * n if apply_nm is None else [bool(f) for f in apply_nm]
        self.fq = fq
        self.outputs = [None] * n
        self._keep = []
        self._descs = (_TensorDesc * max(n, 1))()
        self._k = 0
        self.device = self.dtype = None
        for i, t in enumerate(tensors):
            require_device_tensor(t)
            if self.device is None:
                self.device, self.dtype = t.device, t.dtype
            if t.device != self.device or t.dtype != self.dtype:
                raise ValueError("PreparedList: the tensors of one list share device and dtype")
            if not t.is_contiguous():
                raise ValueError("PreparedList needs contiguous inputs (their storage is bound)")
            if fq.block_size == 0 and not (flags[i] and fq.M > 0):
                self.outputs[i] = t
                continue
            dst = torch.empty_like(t) if outs is None or outs[i] is None else outs[i]
            self.outputs[i] = dst
            if t.numel() == 0:
                continue
            cols = t.shape[-1] if t.dim() else 1
            d = self._descs[self._k]
            d.in_dev, d.out_dev, d.rows, d.cols, d.apply_nm = t.data_ptr(), dst.data_ptr(), t.numel() // cols, cols, 1 if flags[i] else 0
            self._k += 1
            self._keep.append(t)
        self._bound = [(t, t.data_ptr(), t.numel()) for t in self._keep]
        self._plan = fq._plan(self.dtype, self.device) if self._k else None
        self._fn = load_library().bfpq_fake_quantize_list
        self._addr = ctypes.addressof(self._descs)
        # (large tensors of the list get launches of their own, spread over the current stream and this one)
        self._aux = aux_stream_array(self.device, 1) if self._k > 1 else None

    def _rebind(self):
        """descriptor j reads self._keep[j]: follow tensors whose storage moved since they were bound"""
        for j, (t, ptr, n) in enumerate(self._bound):
            cur = t.data_ptr()
            if cur == ptr:
                continue
            if t.dtype != self.dtype or t.device != self.device or t.numel() != n or not t.is_contiguous():
                raise RuntimeError("PreparedList: a bound tensor changed dtype / device / size / layout since it was bound "
                                   f"(was {n} x {self.dtype} on {self.device}, is {t.numel()} x {t.dtype} on {t.device}); build a new list")
            self._descs[j].in_dev = cur
            self._bound[j] = (t, cur, n)

    def run(self):
        if self._k:
            self._rebind()
            dev = self.device
            aux, n_aux = (ctypes.addressof(self._aux), len(self._aux)) if self._aux is not None else (None, 0)
            if torch.cuda.current_device() != dev.index:
                with torch.cuda.device(dev):
                    rc = self._fn(self._plan[1], self._addr, self._k, torch.cuda.current_stream(dev).cuda_stream, aux, n_aux)
            else:
                rc = self._fn(self._plan[1], self._addr, self._k,
                              _raw_stream(dev.index) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream, aux, n_aux)
            if rc:
                check(rc, "bfpq_fake_quantize_list")
        return self.outputs


class SelectWorkspace:
    """Device scratch of the unstructured path (include/bfpq.h: ws_dev): bfpq_select_state, the ticket and coarse histogram of
    the one-launch selection (16-bit dtypes; left zero by its last workgroup), the histogram buffers of the fp32 launch pairs
    (cleared by the apply launch of the same call), and the tie bookkeeping.  One per (device, stream): the launches of one
    call communicate through it."""

    def __init__(self, device):
        self.device = device
        nbytes = int(load_library().bfpq_select_ws_bytes())
        self.ws = torch.zeros(nbytes // 8, dtype=torch.int64, device=device)
        self.dirty = False                                 # a select whose histograms no apply launch has cleared yet
        self.pinned = False                                # a captured graph refers to this workspace: never dropped
        self.hist_ext = None                               # multi-GPU: the local histogram that gets all-gathered

    def ext_hist(self):
        if self.hist_ext is None:
            self.hist_ext = torch.zeros(SELECT_HIST_COPIES * SELECT_HIST_ENTRIES, dtype=torch.int32, device=self.device)
        return self.hist_ext

    def read_state(self):
        """host copy of bfpq_select_state (synchronises; for tests / diagnostics only)"""
        import struct
        raw = self.ws[:SELECT_STATE_BYTES // 8].cpu().numpy().tobytes()
        prefix, mask, k_rem, tau, done, need, ties, k, tie_base, flags, cut_lo, cut_hi, cut_within, cut_total = struct.unpack_from("<IIqIIqqqqIIIII", raw, 0)
        return dict(prefix=prefix, prefix_mask=mask, k_rem=k_rem, tau=tau, done=done, need=need, ties=ties, k=k,
                    tie_base=tie_base, flags=flags, cut_lo=cut_lo, cut_hi=cut_hi, cut_within=cut_within, cut_total=cut_total)


def select_threshold(t, k, ws, numel_global=None, allgather=None):
    """radix-select the k-th smallest magnitude of t (device tensor) and leave threshold + tie bookkeeping in ws.
    Single device: bfpq_select -- ONE launch for a 16-bit dtype (the histogram launch's last workgroup resolves), two for
    fp32 (the high 15 bits of the key, then the low 16).
    Multi-GPU: t is this rank's slab, k / numel_global are global, and allgather(hist) -> (hist_all [R, entries], R, rank)
    gathers the per-rank histograms (the one exchange of the path; an empty slab still joins it)."""
    require_device_tensor(t)
    L = load_library()
    src = t.contiguous()
    code = DTYPE_CODE[src.dtype]
    n = src.numel()
    ng = n if numel_global is None else int(numel_global)
    null = ctypes.c_void_p(0)
    with torch.cuda.device(src.device):
        st = _stream(src)
        if ws.dirty:                                       # the previous fp32 select was never applied (diagnostic use)
            check(L.bfpq_select_reset(_ptr(ws.ws), st), "bfpq_select_reset")
            ws.dirty = False
        if allgather is None:
            if ng != n:
                raise ValueError("select_threshold: numel_global without an allgather")
            check(L.bfpq_select(_ptr(src), n, code, int(k), _ptr(ws.ws), st), "bfpq_select")
            return                                         # (leaves the workspace ready for the next call by itself)
        for p in range(L.bfpq_select_passes(code)):
            hist = ws.ext_hist()                           # zero on entry: resolve clears it again after the gather
            check(L.bfpq_select_hist(_ptr(src) if n else null, n, code, p, int(k), ng, _ptr(ws.ws), _ptr(hist), st), "bfpq_select_hist")
            hist_all, R, rank = allgather(hist)
            check(L.bfpq_select_resolve(_ptr(src) if n else null, n, code, p, int(k), _ptr(hist_all), int(R), int(rank), _ptr(ws.ws),
                                        _ptr(hist), st), "bfpq_select_resolve")


def select_threshold_list(ts, ks, wss, numel_globals, allgather):
    """select_threshold for a LIST of row-sharded tensors of one dtype on one device with ONE exchange per radix pass for the whole
    list (instead of one per tensor): every tensor's local histogram is launched first (its own workspace wss[i]), the 8 copies of each
    are summed, allgather(block [n, ENTRIES]) -> (gathered [R, n, ENTRIES], R, rank) moves them all in one collective, and tensor i
    resolves from its column of the gathered block (bfpq_select_resolve_ex: copies = 1, rank stride = n * ENTRIES).  A tensor whose
    slab is empty keeps a zero row.  Same thresholds, tie bases and cuts as the per-tensor exchange (the sums are the same integers)."""
    n = len(ts)
    if n == 0:
        return
    L = load_library()
    srcs = [t.contiguous() for t in ts]
    for t in srcs:
        require_device_tensor(t)
        if t.dtype != srcs[0].dtype or t.device != srcs[0].device:
            raise ValueError("select_threshold_list: the tensors of one list share device and dtype")
    dev, code = srcs[0].device, DTYPE_CODE[srcs[0].dtype]
    null = ctypes.c_void_p(0)
    E = SELECT_HIST_ENTRIES
    with torch.cuda.device(dev):
        st = _stream(srcs[0])
        for ws in wss[:n]:
            if ws.dirty:
                check(L.bfpq_select_reset(_ptr(ws.ws), st), "bfpq_select_reset")
                ws.dirty = False
        for p in range(L.bfpq_select_passes(code)):
            hists = [ws.ext_hist() for ws in wss[:n]]          # zero on entry: the resolve launch clears them again
            for src, k, ng, ws, h in zip(srcs, ks, numel_globals, wss, hists):
                m = src.numel()
                check(L.bfpq_select_hist(_ptr(src) if m else null, m, code, p, int(k), int(ng), _ptr(ws.ws), _ptr(h), st), "bfpq_select_hist")
            block = torch.stack(hists).view(n, SELECT_HIST_COPIES, E).sum(dim=1, dtype=torch.int32)       # [n, E]: the copies folded
            gathered, R, rank = allgather(block)
            gathered = gathered.contiguous()
            base = gathered.data_ptr()
            for i, (src, k, ws, h) in enumerate(zip(srcs, ks, wss, hists)):
                m = src.numel()
                check(L.bfpq_select_resolve_ex(_ptr(src) if m else null, m, code, p, int(k), ctypes.c_void_p(base + 4 * i * E), int(R), int(rank),
                                               1, n * E, _ptr(ws.ws), _ptr(h), st), "bfpq_select_resolve_ex")
            del gathered


def quantize_threshold(t, ws, block_size, mant_bits, epsilon, want_deq=True, code_bits=0, want_exp=False,
                       stoch_seed=0, out=None):
    """S-first unstructured: prune with the threshold held in ws and HBFP-quantize, one pass
    (bfpq_quantize_threshold).  Returns (deq | None, codes | None, exps | None) like quantize_nm."""
    require_device_tensor(t)
    L = load_library()
    src = t.contiguous()
    rows, cols = rows_cols(src)
    code = DTYPE_CODE[src.dtype]
    dev = src.device
    with torch.cuda.device(dev):
        st = _stream(src)
        fused = L.bfpq_is_fused(rows, cols, code, int(block_size), 0, 0)
        deq = (out if out is not None else torch.empty_like(src)) if want_deq else None
        codes = exps = None
        if code_bits:
            shape = (rows, (cols + 1) // 2) if code_bits == 4 else (rows, cols)
            codes = torch.empty(shape, dtype={4: torch.uint8, 8: torch.int8, 16: torch.int16}[code_bits], device=dev)
        if want_exp:
            exps = torch.empty((rows, (cols + block_size - 1) // block_size), dtype=torch.int8, device=dev)
        scratch = torch.empty_like(src) if (not fused and deq is None) else None
        check(L.bfpq_quantize_threshold(_ptr(src), _ptr(deq), _ptr(codes), _ptr(exps), rows, cols, code, int(block_size),
                                        int(mant_bits), float(epsilon), int(code_bits), int(stoch_seed),
                                        _ptr(exp_window_dev(src.dtype, dev)), _ptr(ws.ws), _ptr(scratch), st),
              "bfpq_quantize_threshold")
        ws.dirty = False
    return deq, codes, exps


def prune_quantize(t, k, ws, block_size, mant_bits, epsilon, out=None):
    """the whole s-first unstructured drop-in op on one device, out = Q(S_k(t)) (bfpq_prune_quantize): the selection launch +
    the fused prune + quantize launch behind one call.  Round-half-even."""
    require_device_tensor(t)
    L = load_library()
    src = t.contiguous()
    rows, cols = rows_cols(src)
    dev = src.device
    with torch.cuda.device(dev):
        dst = out if out is not None else torch.empty_like(src)
        if ws.dirty:
            check(L.bfpq_select_reset(_ptr(ws.ws), _stream(src)), "bfpq_select_reset")
        check(L.bfpq_prune_quantize(_ptr(src), _ptr(dst), rows, cols, DTYPE_CODE[src.dtype], int(block_size), int(mant_bits), float(epsilon),
                                    int(k), _ptr(exp_window_dev(src.dtype, dev)), _ptr(ws.ws), _stream(src)), "bfpq_prune_quantize")
        ws.dirty = False
    return dst


class _PruneDesc(ctypes.Structure):
    """include/bfpq.h: bfpq_prune_desc"""
    _fields_ = [("in_dev", ctypes.c_void_p), ("out_dev", ctypes.c_void_p), ("rows", ctypes.c_int64), ("cols", ctypes.c_int64), ("k", ctypes.c_int64)]


_aux_streams = {}


def aux_streams(device, n=1):
    """n persistent side streams of the device: the aux lanes of the list calls (bfpq_fake_quantize_list, bfpq_prune_quantize_list)"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    lst = _aux_streams.setdefault(idx, [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(torch.device("cuda", idx)))
    return lst[:n]


def aux_stream(device):
    return aux_streams(device, 1)[0]


def aux_stream_array(device, n):
    """the raw handles of aux_streams(device, n) as a C array of void* (the streams themselves live as long as the process)"""
    return (ctypes.c_void_p * n)(*[s.cuda_stream for s in aux_streams(device, n)])


class PruneQuantizeList:
    """A list of tensors of one dtype on one device bound to the s-first unstructured drop-in op (bfpq_prune_quantize_list):
    outputs, descriptors and the workspaces are set up once; run() is ONE ctypes call that spreads the tensors over N_WS
    independent lanes (the current stream and side streams, a workspace each; a tensor's two launches stay on its lane).
    ks: elements to prune per tensor, int(numel * frac) as the reference computes it (bfp_ops.py:66)."""

    N_WS = 4                 # lanes: the current stream + 3 side streams, a workspace each (13B list: 1 lane 16.8 ms, 2: 14.2, 3: 12.7, 4: 12.5)
    GRAPH_FROM = 8           # lists of at least this many tensors replay a hipGraph of the pipeline (captured on the first run)

    def __init__(self, tensors, ks, block_size, mant_bits, epsilon, outs=None):
        tensors = list(tensors)
        self.block_size, self.mant_bits, self.epsilon = int(block_size), int(mant_bits), float(epsilon)
        self.outputs = [None] * len(tensors)
        self._bound = []
        self._descs = (_PruneDesc * max(len(tensors), 1))()
        self._n = 0
        self.device = self.dtype = None
        for i, (t, k) in enumerate(zip(tensors, ks)):
            require_device_tensor(t)
            if self.device is None:
                self.device, self.dtype = t.device, t.dtype
            if t.device != self.device or t.dtype != self.dtype:
                raise ValueError("PruneQuantizeList: the tensors of one list share device and dtype")
            if not t.is_contiguous():
                raise ValueError("PruneQuantizeList needs contiguous inputs (their storage is bound)")
            dst = torch.empty_like(t) if outs is None or outs[i] is None else outs[i]
            self.outputs[i] = dst
            if t.numel() == 0:
                continue
            rows, cols = rows_cols(t)
            d = self._descs[self._n]
            d.in_dev, d.out_dev, d.rows, d.cols, d.k = t.data_ptr(), dst.data_ptr(), rows, cols, int(k)
            self._bound.append((t, t.data_ptr(), t.numel()))
            self._n += 1
        if self._n:
            with torch.cuda.device(self.device):
                self._ws = [SelectWorkspace(self.device) for _ in range(self.N_WS)]
                self._ws_ptrs = (ctypes.c_void_p * self.N_WS)(*[w.ws.data_ptr() for w in self._ws])
                self._win = exp_window_dev(self.dtype, self.device)
                self._aux = aux_stream_array(self.device, self.N_WS - 1)
        self._fn = load_library().bfpq_prune_quantize_list
        self._graph = None

    def _issue(self, pipelined):
        dev = self.device
        with torch.cuda.device(dev):
            rc = self._fn(ctypes.addressof(self._descs), self._n, DTYPE_CODE[self.dtype], self.block_size, self.mant_bits, self.epsilon,
                          self._win.data_ptr(), ctypes.addressof(self._ws_ptrs), self.N_WS, torch.cuda.current_stream(dev).cuda_stream,
                          ctypes.addressof(self._aux) if pipelined else None, len(self._aux) if pipelined else 0)
        if rc:
            check(rc, "bfpq_prune_quantize_list")

    def run(self, pipelined=True, graph=None):
        """graph: replay a hipGraph of the whole pipeline (captured on the first such run; every pointer in it is bound --
        a tensor whose storage moved drops the graph).  Default: for lists of GRAPH_FROM tensors or more, outside any
        capture already in progress.  The ~6 host calls per tensor (2 launches, 4 event operations) otherwise cost more
        host time than a small tensor's launches take on the device."""
        if self._n:
            for j, (t, ptr, n) in enumerate(self._bound):                 # follow tensors whose storage moved (see PreparedList)
                cur = t.data_ptr()
                if cur != ptr:
                    if t.dtype != self.dtype or t.device != self.device or t.numel() != n or not t.is_contiguous():
                        raise RuntimeError("PruneQuantizeList: a bound tensor changed dtype / device / size / layout since it was bound; build a new list")
                    self._descs[j].in_dev = cur
                    self._bound[j] = (t, cur, n)
                    self._graph = None
            if graph is None:
                graph = self._n >= self.GRAPH_FROM
            if graph and pipelined and not torch.cuda.is_current_stream_capturing():
                if self._graph is None:
                    self._issue(True)                                     # (warm-up outside the capture)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.device(self.device), torch.cuda.graph(g):
                        self._issue(True)
                    self._graph = g
                with torch.cuda.device(self.device):
                    self._graph.replay()
            else:
                self._issue(pipelined)
        return self.outputs


def threshold_apply(t, ws, out=None):
    """zero everything below the threshold in ws plus the first `need` ties (flat order, lower ranks first)"""
    require_device_tensor(t)
    L = load_library()
    src = t.contiguous()
    code = DTYPE_CODE[src.dtype]
    with torch.cuda.device(src.device):
        dst = out if out is not None else torch.empty_like(src)
        check(L.bfpq_threshold_apply(_ptr(src), _ptr(dst), src.numel(), code, _ptr(ws.ws), _stream(src)), "bfpq_threshold_apply")
        ws.dirty = False
    return dst


def int_channel_view(shape, weight):
    """(outer, C, inner): the per-channel view int_ops.Quantizer.find_params takes (reference int_ops.py:38-50)"""
    shape = tuple(shape)
    n = 1
    for d in shape:
        n *= d
    if weight:
        if len(shape) < 1:
            raise ValueError("'int' format needs at least a 1-D weight")
        return 1, shape[0], n // max(shape[0], 1)
    if len(shape) == 4:
        return shape[0], shape[1], shape[2] * shape[3]
    if len(shape) in (2, 3):
        return n // max(shape[-1], 1), shape[-1], 1
    raise ValueError("'int' format handles 2-D, 3-D and 4-D activations (reference int_ops.py:43-50)")


def int_quantize(t, bits, weight):
    """per-channel symmetric integer fake-quantization (bfpq_int_quantize); fp32 result like the reference"""
    require_device_tensor(t)
    L = load_library()
    src = t.contiguous()
    outer, C, inner = int_channel_view(src.shape, weight)
    out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    if src.numel() == 0:
        return out
    with torch.cuda.device(src.device):
        ws = torch.empty(2 * C, dtype=torch.int32, device=src.device) if outer != 1 else None
        check(L.bfpq_int_quantize(_ptr(src), _ptr(out), outer, C, inner, DTYPE_CODE[src.dtype], int(bits), _ptr(ws), _stream(src)),
              "bfpq_int_quantize")
    return out


def dequantize(codes, exps, cols, dtype, block_size, mant_bits, code_bits, out=None):
    """packed codes + exponents -> [rows, cols] tensor of `dtype` (bfpq_dequantize)"""
    if exps.device.type != "cuda" or codes.device != exps.device:
        raise NativeUnavailable("packed codes / exponents must live on one ROCm device")
    if dtype not in DTYPE_CODE:
        raise TypeError(f"unsupported dtype {dtype}")
    L = load_library()
    rows = exps.shape[0]
    dst = out if out is not None else torch.empty((rows, cols), dtype=dtype, device=exps.device)
    if rows * cols == 0:
        return dst
    with torch.cuda.device(exps.device):
        c, e = codes.contiguous(), exps.contiguous()
        check(L.bfpq_dequantize(_ptr(c), _ptr(e), _ptr(dst), rows, int(cols), DTYPE_CODE[dtype], int(block_size), int(mant_bits),
                                int(code_bits), ctypes.c_void_p(torch.cuda.current_stream(exps.device).cuda_stream)), "bfpq_dequantize")
    return dst


def is_fused(t, block_size, N=0, M=0):
    """would quantize_nm take the single-pass kernel for this (contiguous) tensor?"""
    rows, cols = rows_cols(t)
    return bool(load_library().bfpq_is_fused(rows, cols, DTYPE_CODE[t.dtype], int(block_size), int(N), int(M))) and t.data_ptr() % 16 == 0


def compact24(codes):
    """2:4 compaction of a 4-bit code tensor (uint8, numel % 4 == 0): (vals uint8 [numel/2], idx uint8 [numel/4]).
    Raises ValueError if some group of 4 codes has more than 2 non-zeros (synchronises to read the flag)."""
    require_device_tensor_any(codes)
    c = codes.contiguous().view(-1)
    n = c.numel()
    if n % 4:
        raise ValueError("compact24 needs a multiple of 8 elements (4 code bytes)")
    with torch.cuda.device(c.device):
        vals = torch.empty(n // 2, dtype=torch.uint8, device=c.device)
        idx = torch.empty(n // 4, dtype=torch.uint8, device=c.device)
        status = torch.zeros(1, dtype=torch.int32, device=c.device)
        check(load_library().bfpq_compact24(_ptr(c), _ptr(vals), _ptr(idx), n, _ptr(status), _stream(c)), "bfpq_compact24")
    if int(status.item()):
        raise ValueError("compact24: a group of 4 codes has more than 2 non-zeros (the tensor is not 2:4 sparse)")
    return vals, idx


def expand24(vals, idx):
    """inverse of compact24: the 4-bit code bytes (uint8 [4 * idx.numel()])"""
    require_device_tensor_any(vals)
    n = idx.numel() * 4
    with torch.cuda.device(vals.device):
        codes = torch.empty(n, dtype=torch.uint8, device=vals.device)
        check(load_library().bfpq_expand24(_ptr(vals.contiguous()), _ptr(idx.contiguous()), _ptr(codes), n, _stream(vals)), "bfpq_expand24")
    return codes


def require_device_tensor_any(t):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise NativeUnavailable("the BFP engine runs on a ROCm device only (no CPU fallback)")


def mfma_tiles(wcodes, wexps):
    """one-time repack of a packed HBFP4 weight (codes [N, K/2] uint8, exps [N, K/64] int8, block 64) into the layout
    bfpq_hbfp_linear_decode_tiled reads: a pure permutation, done with torch views"""
    N, K = wexps.shape[0], wexps.shape[1] * 64
    assert N % 16 == 0 and K % 128 == 0
    t = wcodes.view(N // 16, 16, K // 128, 2, 4, 8).permute(0, 2, 4, 1, 3, 5).contiguous().view(N // 16, K // 128, 64, 16)
    e = wexps.view(N // 16, 16, K // 128, 2).permute(0, 2, 1, 3).contiguous()
    return t, e


def hbfp_linear_decode_tiled(x, wtiles, wexpt, N, w_mant_bits, x_mant_bits=7, epsilon=1e-8, out_dtype=None):
    """hbfp_linear_decode on the MFMA-tiled weight (see mfma_tiles)"""
    require_device_tensor(x)
    L = load_library()
    K = x.shape[-1]
    T = x.numel() // K
    if T < 1 or T > 64 or not L.bfpq_hbfp_linear_tiled_ok(N, K):
        raise ValueError(f"hbfp_linear_decode_tiled needs 1..64 tokens, N % 16 == 0, K % 128 == 0 and K >= 256 (got T={T}, N={N}, K={K})")
    dev = x.device
    out_dtype = out_dtype or x.dtype
    with torch.cuda.device(dev):
        xc16 = torch.empty((T, K), dtype=torch.int8, device=dev)          # (token columns past T re-read row T-1: no padding rows)
        xe16 = torch.empty((T, K // 64), dtype=torch.int8, device=dev)
        quantize_nm(x.reshape(T, K), 64, x_mant_bits, epsilon, want_deq=False, code_bits=8, want_exp=True,
                    codes_out=xc16, exps_out=xe16)
        out = torch.empty((T, N), dtype=out_dtype, device=dev)
        check(L.bfpq_hbfp_linear_decode_tiled(_ptr(wtiles), _ptr(wexpt), _ptr(xc16), _ptr(xe16), _ptr(out), T, N, K,
                                              DTYPE_CODE[out_dtype], int(w_mant_bits), int(x_mant_bits), _stream(x)),
              "bfpq_hbfp_linear_decode_tiled")
    return out.view(tuple(x.shape[:-1]) + (N,))


def hbfp_linear_decode(x, wcodes, wexps, w_mant_bits, x_mant_bits=7, epsilon=1e-8, out_dtype=None):
    """out = x @ W^T for <= 16 tokens, W given as packed HBFP (4-bit codes [N, K/2], int8 exponents [N, K/64], block 64),
    x quantized on the fly to HBFP(x_mant_bits + 1) block 64; integer block dot products on the int8 matrix cores."""
    require_device_tensor(x)
    L = load_library()
    K = x.shape[-1]
    T = x.numel() // K
    N = wexps.shape[0]
    slices = L.bfpq_hbfp_linear_slices(N, K)
    if T < 1 or T > 16 or slices < 0:
        raise ValueError(f"hbfp_linear_decode needs 1..16 tokens, N % 16 == 0 and K % 256 == 0 (got T={T}, N={N}, K={K})")
    dev = x.device
    out_dtype = out_dtype or x.dtype
    with torch.cuda.device(dev):
        # the activation quantizer writes straight into the 16-row operand buffers (rows >= T stay uninitialised:
        # each token's results live in its own lanes and rows >= T are never read back)
        xc16 = torch.empty((16, K), dtype=torch.int8, device=dev)
        xe16 = torch.empty((16, K // 64), dtype=torch.int8, device=dev)
        quantize_nm(x.reshape(T, K), 64, x_mant_bits, epsilon, want_deq=False, code_bits=8, want_exp=True,
                    codes_out=xc16[:T], exps_out=xe16[:T])
        slabs = torch.empty((slices, 16, N), dtype=torch.float32, device=dev)
        out = torch.empty((T, N), dtype=out_dtype, device=dev)
        check(L.bfpq_hbfp_linear_decode(_ptr(wcodes.contiguous()), _ptr(wexps.contiguous()), _ptr(xc16), _ptr(xe16), _ptr(out),
                                        _ptr(slabs), T, N, K, DTYPE_CODE[out_dtype], int(w_mant_bits), int(x_mant_bits), _stream(x)),
              "bfpq_hbfp_linear_decode")
    return out.view(tuple(x.shape[:-1]) + (N,))


def mx8_from_hbfp(codes, exps, cols, mant_bits, code_bits):
    """packed HBFP (codes + int8 exponents, block 64, mantissas <= 4 bits) -> the operand image of the block-scaled matrix
    instruction: e4m3 bytes [rows, cols] + E8M0 scale bytes [rows, cols/64] (bfpq_mx8_from_hbfp)"""
    require_device_tensor_any(codes)
    L = load_library()
    rows = exps.numel() // (cols // 64)
    dev = codes.device
    with torch.cuda.device(dev):
        o8 = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
        osc = torch.empty((rows, cols // 64), dtype=torch.uint8, device=dev)
        check(L.bfpq_mx8_from_hbfp(_ptr(codes.contiguous()), _ptr(exps.contiguous()), _ptr(o8), _ptr(osc), rows, cols, int(code_bits), int(mant_bits),
                                   torch.cuda.current_stream(dev).cuda_stream), "bfpq_mx8_from_hbfp")
    return o8, osc


def quantize_mx8(x, mant_bits, epsilon=1e-8):
    """a 2-D tensor [T, K] -> its HBFP(mant_bits + 1) block-64 image for the block-scaled matrix unit: e4m3 bytes [T, K] + E8M0
    scales [T, K/64], in one pass (bfpq_quantize_mx8); the two-step route (int8 codes + mx8_from_hbfp) only for shapes it refuses."""
    require_device_tensor(x)
    L = load_library()
    T, K = x.shape
    dev = x.device
    with torch.cuda.device(dev):
        if K % 64 == 0:
            x = x.contiguous()
            x8 = torch.empty((T, K), dtype=torch.uint8, device=dev)
            xs = torch.empty((T, K // 64), dtype=torch.uint8, device=dev)
            rc = L.bfpq_quantize_mx8(_ptr(x), _ptr(x8), _ptr(xs), T, K, DTYPE_CODE[x.dtype], int(mant_bits), float(epsilon),
                                     _ptr(exp_window_dev(x.dtype, dev)), _stream(x))
            if rc == 0:
                return x8, xs
            if rc != BFPQ_E_UNSUPPORTED:
                check(rc, "bfpq_quantize_mx8")
        xc = torch.empty((T, K), dtype=torch.int8, device=dev)
        xe = torch.empty((T, K // 64), dtype=torch.int8, device=dev)
        quantize_nm(x, 64, mant_bits, epsilon, want_deq=False, code_bits=8, want_exp=True, codes_out=xc, exps_out=xe)
        return mx8_from_hbfp(xc, xe, K, mant_bits, 8)


_last_image = {}          # device -> (key, the tensor itself, image): q/k/v (and gate/up) of a block are called on the same tensor
SPLIT_K = True            # False: never split K (A/B measurements)
SHARE_ACT_IMAGE = True    # False: every call quantizes its activation (single-layer benchmarks that re-use one input tensor)


def tensor_version(t):
    """in-place version counter of a tensor, or None where it has none (tensors created under torch.inference_mode())"""
    try:
        return t._version
    except RuntimeError:
        return None


def _shared_image(x2, mant_bits, epsilon):
    """quantize_mx8 with a one-entry memo per device: the projections that share an input (q/k/v, gate/up) quantize it once.
    The entry holds the tensor it was made from, so that tensor's memory cannot be handed to another tensor while the entry
    lives; the key includes the in-place version counter.  The memo is bypassed -- the image is made afresh and nothing is
    stored -- whenever the key cannot vouch for the bytes: while a hipGraph is being captured (the launch is recorded, not
    run: an entry made then would hold an image that was never computed), for inference-mode tensors (no version counter),
    and with SHARE_ACT_IMAGE off.  Writes the version counter does not see (`.data` assignments, raw-pointer writes such as
    this library's own `out=` arguments) are outside what the key can detect: callers that mutate an activation that way
    between two projections must call forget_shared_images() (PackedBFP.linear and the cached BFPLinear never do)."""
    if not SHARE_ACT_IMAGE or torch.cuda.is_current_stream_capturing():
        return quantize_mx8(x2, mant_bits, epsilon)
    ver = None if x2.is_inference() else tensor_version(x2)
    if ver is None:
        return quantize_mx8(x2, mant_bits, epsilon)
    key = (x2.data_ptr(), ver, tuple(x2.shape), tuple(x2.stride()), x2.dtype, int(mant_bits), float(epsilon),
           torch.cuda.current_stream(x2.device).cuda_stream)
    hit = _last_image.get(x2.device)
    if hit is not None and hit[0] == key:
        return hit[2]
    img = quantize_mx8(x2, mant_bits, epsilon)
    _last_image[x2.device] = (key, x2, img)
    return img


def forget_shared_images():
    """drop the per-device activation-image memo (and the activation each entry keeps alive)"""
    _last_image.clear()


def hbfp_linear_mx8_ok(T, N, K, w_mant_bits, x_mant_bits, block_size=64):
    return block_size == 64 and 1 <= w_mant_bits <= 4 and 1 <= x_mant_bits <= 4 and bool(load_library().bfpq_hbfp_linear_mx8_ok(T, N, K))


def hbfp_linear_mx8(x, w8, wscale, x_mant_bits, epsilon=1e-8, bias=None, out_dtype=None):
    """out = Q(x) @ W^T (+ bias) for any number of tokens on the block-scaled matrix instruction: x is quantized to
    HBFP(x_mant_bits + 1) block 64 (int8 codes + exponents), turned into its e4m3 / E8M0 image and multiplied with the
    weight's image (mx8_from_hbfp of the packed weight, made once)."""
    require_device_tensor(x)
    L = load_library()
    K = x.shape[-1]
    T = x.numel() // K
    N = w8.shape[0]
    if not hbfp_linear_mx8_ok(T, N, K, 1, x_mant_bits):
        raise ValueError(f"hbfp_linear_mx8 needs K % 256 == 0 and activation mantissas of <= 4 bits (got T={T}, N={N}, K={K}, x_mant_bits={x_mant_bits})")
    dev = x.device
    out_dtype = out_dtype or x.dtype
    with torch.cuda.device(dev):
        x8, xs = _shared_image(x.reshape(T, K), x_mant_bits, epsilon)
        out = torch.empty((T, N), dtype=out_dtype, device=dev)
        if bias is not None:
            bias = bias.to(out_dtype).contiguous()
        parts = L.bfpq_hbfp_linear_mx8_parts(T, N, K) if SPLIT_K else 1
        if parts > 1:                                                  # few tokens: K split over several workgroups per tile, fp32 slabs
            slabs = torch.empty((parts, T, N), dtype=torch.float32, device=dev)
            check(L.bfpq_hbfp_linear_mx8_splitk(_ptr(x8), _ptr(xs), _ptr(w8), _ptr(wscale), _ptr(bias) if bias is not None else None, _ptr(out),
                                                _ptr(slabs), parts, T, N, K, DTYPE_CODE[out_dtype], _stream(x)), "bfpq_hbfp_linear_mx8_splitk")
        else:
            check(L.bfpq_hbfp_linear_mx8(_ptr(x8), _ptr(xs), _ptr(w8), _ptr(wscale), _ptr(bias) if bias is not None else None, _ptr(out),
                                         T, N, K, DTYPE_CODE[out_dtype], _stream(x)), "bfpq_hbfp_linear_mx8")
    return out.view(tuple(x.shape[:-1]) + (N,))
