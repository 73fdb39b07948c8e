"""The 'int' per-column activation path through the C ABI (bfpq_int_quantize: fill + min/max + quantize launches), hipGraph of L calls over R rotating
inputs.  Development builds read BFPQ_INT_QG / BFPQ_INT_MG (workgroup targets of the quantize / min-max launch).
usage: python tools_dev/ab_int.py [rows cols dtype] [weight] [name=path.so ...]   (weight: the per-row form, outer = 1; builds interleaved in one process)"""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_sparsity_interplay_amd import native
pos = [a for a in sys.argv[1:] if '=' not in a and a != 'weight']
weight = 'weight' in sys.argv
rows = int(pos[0]) if len(pos) > 0 else 4096
cols = int(pos[1]) if len(pos) > 1 else 4096
dname = pos[2] if len(pos) > 2 else 'bf16'
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dname]
L, R, ROUNDS = 40, 8, 9
dev = torch.device('cuda:0')
lib = native.load_library()
ins = [(torch.randn(rows, cols, generator=torch.Generator().manual_seed(r))).to(dt).to(dev) for r in range(R)]
outs = [torch.empty(rows, cols, dtype=torch.float32, device=dev) for _ in range(R)]
n_ws = int(lib.bfpq_int_workspace_elems(cols))
ws_any = torch.empty(n_ws, dtype=torch.int32, device=dev)
ws_kept = torch.full((n_ws,), -1, dtype=torch.int32, device=dev)
variants = {}
for a in sys.argv[1:]:
    if '=' in a:
        n, p = a.split('=')
        l2 = ctypes.CDLL(os.path.abspath(p))
        l2.bfpq_int_quantize.argtypes = lib.bfpq_int_quantize.argtypes
        l2.bfpq_int_quantize.restype = ctypes.c_int
        variants[n] = (l2.bfpq_int_quantize, ws_any)
if not variants: variants = {'int': (lib.bfpq_int_quantize, ws_any)}
outer, C, inner = (1, rows, cols) if weight else (rows, cols, 1)
graphs, ref = {}, None
for n, (fn, ws) in variants.items():
    def run():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(L):
            rc = fn(ins[i % R].data_ptr(), outs[i % R].data_ptr(), outer, C, inner, native.DTYPE_CODE[dt], 8, ws.data_ptr(), st)
            assert rc == 0, rc
    for o in outs: o.zero_()
    run(); torch.cuda.synchronize()
    got = [o.clone() for o in outs[:2]]
    if ref is None: ref = got
    else: assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(ref, got)), n
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    graphs[n] = g
times = {n: [] for n in graphs}
for r in range(ROUNDS):
    for n, g in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) * 1e3 / L)
bytes_ = rows * cols * (ins[0].element_size() + 4)
for n, t in times.items():
    med = statistics.median(t)
    print(f"[{rows},{cols}] {dname}{' weight' if weight else ''} {n:6s} median {med:6.2f} us  min {min(t):6.2f}  -> {bytes_/med/1e3:7.1f} GB/s ({bytes_/med/1e3/8000*100:4.1f}% of 8 TB/s)")
