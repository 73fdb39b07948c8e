"""A/B harness for the LIST kernel (k_fused_batched): several builds of libbfpq.so in ONE process, interleaved rounds, one list call per
workload (eager, HIP events).  Workloads: 64 x [4096,11008] bf16 2:4 -> HBFP4 with the own-launch threshold switched off (the list
kernel on large tensors), ViT-L's 144 weights (f32, HBFP8 b16, 1:4), OPT-125m's 72 weights (f32, HBFP8 b32 dense).
usage: python tools_dev/ab_list.py name=path.so ..."""
import ctypes, shutil, statistics, sys, tempfile, torch
sys.path.insert(0, '.')
from quantization_sparsity_interplay_amd import native
dev = torch.device('cuda:0')
ROUNDS = 9
vp, i32 = ctypes.c_void_p, ctypes.c_int


def tensors(shapes, dt, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    xs = [(torch.randn(r, c, generator=g, device=dev) * 0.02).to(dt) for r, c in shapes]
    return xs, [torch.empty_like(x) for x in xs]


vit = [(1024, 1024)] * 4 * 24 + [(4096, 1024)] * 24 + [(1024, 4096)] * 24
opt = [(768, 768)] * 4 * 12 + [(3072, 768)] * 12 + [(768, 3072)] * 12
work = {
    "64 x [4096,11008] bf16 2:4 (list kernel forced)": (tensors([(4096, 11008)] * 64, torch.bfloat16, 1), native.FastQuant(64, 3, 1e-8, 2, 4, True), 1 << 20),
    "ViT-L 144 weights f32 HBFP8 b16 1:4": (tensors(vit, torch.float32, 2), native.FastQuant(16, 7, 1e-8, 1, 4, True), 24),
    "OPT-125m 72 weights f32 HBFP8 b32 dense": (tensors(opt, torch.float32, 3), native.FastQuant(32, 7, 1e-8, 0, 0, True), 24),
}
libs = {}
for a in sys.argv[1:]:
    n, p = a.split('=')
    q = tempfile.mktemp(suffix=f"_{n}.so"); shutil.copy(p, q)              # (a private copy: its own tuning globals)
    lib = ctypes.CDLL(q)
    lib.bfpq_fake_quantize_list.argtypes = [vp, vp, i32, vp, vp, i32]
    lib.bfpq_fake_quantize_list.restype = i32
    lib.bfpq_tune.argtypes = [i32, i32]
    libs[n] = lib
times = {(w, n): [] for w in work for n in libs}
for wname, ((xs, ys), fq, own_mb) in work.items():
    plan = fq._plan(xs[0].dtype, dev)
    descs = (native._TensorDesc * len(xs))()
    for d, x, y in zip(descs, xs, ys):
        d.in_dev, d.out_dev, d.rows, d.cols, d.apply_nm = x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], 1

    def call(lib):
        assert lib.bfpq_tune(3, own_mb) == 0
        rc = lib.bfpq_fake_quantize_list(plan[1], ctypes.addressof(descs), len(xs), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None, 0)
        assert rc == 0, rc
    ref = None
    for n, lib in libs.items():
        call(lib); torch.cuda.synchronize()
        it = torch.int32 if ys[0].dtype == torch.float32 else torch.int16
        got = [y.view(it).clone() for y in ys[:3]]
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), "builds disagree"
    for r in range(ROUNDS):
        for n, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call(lib); e1.record(); torch.cuda.synchronize()
            times[(wname, n)].append(e0.elapsed_time(e1) * 1e3)
    numel = sum(x.numel() for x in xs)
    for n in libs:
        med = statistics.median(times[(wname, n)])
        b = numel * 2 * xs[0].element_size()
        print(f"{wname:52s} {n:10s} {med:9.1f} us  {b / med / 1e3:7.0f} GB/s ({b / med / 80e3:5.1f} %)", flush=True)
