"""sweep the workgroup cap on the real kernel (one lib, bfpq_tune), interleaved rounds"""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, '.')
from quantization_sparsity_interplay_amd import native
lib = native.load_library()
rows, cols, L, R, ROUNDS = 4096, 11008, 100, int(os.environ.get("ROTATE", "8")), 9
NN, MM = 2, 4
if len(sys.argv) >= 5:                      # usage: ab_grid.py rows cols N M
    rows, cols, NN, MM = (int(a) for a in sys.argv[1:5])
dev = torch.device('cuda:0')
ins = [(torch.randn(rows, cols, generator=torch.Generator().manual_seed(r)) * 0.02).to(torch.bfloat16).to(dev) for r in range(R)]
outs = [torch.empty_like(x) for x in ins]
PACKED = os.environ.get("PACKED") == "1"            # 4-bit codes + exponents instead of the dequantised tensor
codes = [torch.empty((rows, cols // 2), dtype=torch.uint8, device=dev) for _ in ins]
exps = [torch.empty((rows, cols // 64), dtype=torch.int8, device=dev) for _ in ins]
win = native.exp_window_dev(torch.bfloat16, dev); lut = native.nm4_lut_dev(NN, dev) if MM == 4 else None
nwg = rows * cols // 8 // 256
grids = sorted(set([(nwg + s - 1) // s for s in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 18, 20, 24, 32)] + [1280, 2048, 1024]))
if os.environ.get("GRIDS"):
    grids = [int(g) for g in os.environ["GRIDS"].split(",")]
graphs = {}
for gmax in grids:
    assert lib.bfpq_tune(0, gmax) == 0
    def run():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(L):
            if PACKED:
                rc = lib.bfpq_quantize_nm(ins[i % R].data_ptr(), None, codes[i % R].data_ptr(), exps[i % R].data_ptr(), rows, cols, 2, 64, 3, 1e-8, NN, MM, 1, 4, 0,
                                          win.data_ptr(), lut.data_ptr() if lut is not None else None, None, st)
            else:
                rc = lib.bfpq_quantize_nm(ins[i % R].data_ptr(), outs[i % R].data_ptr(), None, None, rows, cols, 2, 64, 3, 1e-8, NN, MM, 1, 0, 0,
                                          win.data_ptr(), lut.data_ptr() if lut is not None else None, None, st)
            assert rc == 0
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    graphs[gmax] = g
times = {k: [] for k in graphs}
for r in range(ROUNDS):
    for k, g in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) * 1e3 / L)
for k, t in times.items():
    med = statistics.median(t)
    bpe = 2.516 if PACKED else 4
    print(f"maxgrid {k:5d}  sweeps {nwg / min(k, nwg):6.2f}  median {med:6.2f} us  {rows*cols*bpe/med/1e3:7.0f} GB/s ({rows*cols*bpe/med/1e3/80:4.1f}%)")
