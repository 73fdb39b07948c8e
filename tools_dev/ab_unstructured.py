"""A/B of the one-device unstructured call (bfpq_prune_quantize: selection launch + fused prune + quantize launch): several builds of libbfpq.so in
ONE process, interleaved rounds, hipGraph of L calls over R rotating inputs; outputs of the builds compared bit for bit.
usage: python tools_dev/ab_unstructured.py [rows cols dtype] name=path.so ..."""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_sparsity_interplay_amd import native
args = [a for a in sys.argv[1:] if '=' not in a]
rows = int(args[0]) if len(args) > 0 else 5120
cols = int(args[1]) if len(args) > 1 else 5120
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args[2] if len(args) > 2 else 'bf16']
L, R, ROUNDS = 40, 8, 9
dev = torch.device('cuda:0')
vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
libs = {}
for a in sys.argv[1:]:
    if '=' not in a: continue
    n, p = a.split('=')
    lib = ctypes.CDLL(os.path.abspath(p))
    lib.bfpq_prune_quantize.argtypes = [vp, vp, i64, i64, i32, i32, i32, dbl, i64, vp, vp, vp]
    lib.bfpq_prune_quantize.restype = i32
    lib.bfpq_select_ws_bytes.restype = i64
    libs[n] = lib
ins = [(torch.randn(rows, cols, generator=torch.Generator().manual_seed(r)) * 0.02).to(dt).to(dev) for r in range(R)]
outs = [torch.empty_like(x) for x in ins]
win = native.exp_window_dev(dt, dev)
k = rows * cols // 2
graphs, ref = {}, None
for n, lib in libs.items():
    ws = torch.zeros(int(lib.bfpq_select_ws_bytes()) // 8, dtype=torch.int64, device=dev)
    def run():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(L):
            rc = lib.bfpq_prune_quantize(ins[i % R].data_ptr(), outs[i % R].data_ptr(), rows, cols, native.DTYPE_CODE[dt], 64, 3, 1e-8, k, win.data_ptr(), ws.data_ptr(), st)
            assert rc == 0, rc
    for o in outs: o.zero_()
    run(); torch.cuda.synchronize()
    got = [o.clone() for o in outs]
    if ref is None: ref = got
    else: assert all(torch.equal(a.view(torch.int16 if dt != torch.float32 else torch.int32), b.view(torch.int16 if dt != torch.float32 else torch.int32)) for a, b in zip(ref, got)), n
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    graphs[n] = (g, ws)
times = {n: [] for n in graphs}
for r in range(ROUNDS):
    for n, (g, _) in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) * 1e3 / L)
b2 = rows * cols * ins[0].element_size() * 3
for n, t in times.items():
    med = statistics.median(t)
    print(f"[{rows},{cols}] {n:8s} median {med:6.2f} us  min {min(t):6.2f}  -> two/three-read figure {b2/med/1e3:7.1f} GB/s ({b2/med/1e3/8000*100:4.1f}% of 8 TB/s)")
