"""A/B harness: several builds of libbfpq.so in ONE process, interleaved rounds, hipGraph of L launches each.
usage: [NM=4:8] [COLS=11000] python tools_dev/ab.py name=path.so ...   (headline workload; NM: another N:M pattern, 0:0 = dense; COLS: another row length)"""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, '.')
from quantization_sparsity_interplay_amd import native
rows, cols, L, R, ROUNDS = 4096, int(os.environ.get("COLS", "11008")), 100, 8, 12
NN, MM = (int(v) for v in os.environ.get("NM", "2:4").split(":"))
dev = torch.device('cuda:0')
libs = {}
for a in sys.argv[1:]:
    n, p = a.split('=')
    lib = ctypes.CDLL(p)
    vp, i64, i32, u64, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_double
    lib.bfpq_quantize_nm.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, dbl, i32, i32, i32, i32, u64, vp, vp, vp, vp]
    lib.bfpq_quantize_nm.restype = i32
    libs[n] = lib
ins = [(torch.randn(rows, cols, generator=torch.Generator().manual_seed(r)) * 0.02).to(torch.bfloat16).to(dev) for r in range(R)]
outs = [torch.empty_like(x) for x in ins]
win = native.exp_window_dev(torch.bfloat16, dev); lut = None if MM == 0 else (native.nm4_lut_dev(NN, dev) if MM == 4 else native.nm8_lut_dev(NN, dev))
graphs = {}
for n, lib in libs.items():
    def run():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(L):
            rc = lib.bfpq_quantize_nm(ins[i % R].data_ptr(), outs[i % R].data_ptr(), None, None, rows, cols, 2, 64, 3, 1e-8, NN, MM, 1, 0, 0,
                                      win.data_ptr(), lut.data_ptr() if lut is not None else None, None, st)
            assert rc == 0, rc
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    graphs[n] = g
times = {n: [] for n in libs}
for r in range(ROUNDS):
    for n, g in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) * 1e3 / L)
bytes_ = rows * cols * 4
for n, t in times.items():
    med, mn = statistics.median(t), min(t)
    print(f"{n:10s} median {med:6.2f} us  min {mn:6.2f} us  -> {bytes_/med/1e3:7.1f} GB/s median ({bytes_/med/1e3/8000*100:4.1f}% of 8 TB/s)")
