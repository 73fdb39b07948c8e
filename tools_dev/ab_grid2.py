"""workgroup-cap sweep for smaller shapes (dense HBFP4 bf16), interleaved rounds"""
import ctypes, sys, statistics, torch
sys.path.insert(0, '.')
from quantization_sparsity_interplay_amd import native
lib = native.load_library()
dev = torch.device('cuda:0')
win = native.exp_window_dev(torch.bfloat16, dev); lut = native.nm4_lut_dev(2, dev)
for rows, cols, N, M in ((4096, 4096, 0, 0), (4096, 4096, 2, 4), (5120, 5120, 0, 0), (1024, 4096, 0, 0)):
    L, R, ROUNDS = 100, 16, 7
    ins = [(torch.randn(rows, cols, generator=torch.Generator().manual_seed(r)) * 0.02).to(torch.bfloat16).to(dev) for r in range(R)]
    outs = [torch.empty_like(x) for x in ins]
    nwg = rows * cols // 8 // 256
    graphs = {}
    for gmax in (768, 1024, 1280, 1536, 2048, 3072, 4096, 8192):
        lib.bfpq_tune(0, gmax)
        def run():
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for i in range(L):
                rc = lib.bfpq_quantize_nm(ins[i % R].data_ptr(), outs[i % R].data_ptr(), None, None, rows, cols, 2, 64, 3, 1e-8, N, M, 1, 0, 0,
                                          win.data_ptr(), lut.data_ptr(), None, st)
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
    print(f"[{rows},{cols}] {N}:{M}  work = {nwg} workgroups")
    for k, t in times.items():
        med = statistics.median(t)
        print(f"   cap {k:5d}  median {med:6.2f} us  {rows*cols*4/med/1e3:7.0f} GB/s ({rows*cols*4/med/1e3/80:4.1f}%)")
