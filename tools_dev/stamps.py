#!/usr/bin/env python3
"""Phase timing of the unstructured path's histogram / resolve launches from in-kernel stamps (100 MHz clock).
Build the instrumented library first:
    make -C quantization-sparsity-interplay_amd/csrc -j8 EXTRA=-DBFPQ_STAMPS OUT=$PWD/tools_dev/_build/libbfpq_stamps.so OBJDIR=$PWD/tools_dev/_build/obj
    BFPQ_LIB=$PWD/tools_dev/_build/libbfpq_stamps.so python tools_dev/stamps.py
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5120, 5120)
xs = [(torch.randn(rows, cols, generator=torch.Generator().manual_seed(1 + i)) * 0.02).to(torch.bfloat16).cuda() for i in range(int(os.environ.get("ROTATE", "8")))]
x = xs[0]
L = native.load_library()
ws = bfp_ops._workspace(x.device)
for i in range(3 * len(xs) + 1):                  # (rotating inputs and outputs: neither L2 nor the Infinity Cache serves the stream)
    x = xs[i % len(xs)]
    native.select_threshold(x, x.numel() // 2, ws)
torch.cuda.synchronize()
buf = np.zeros((3, 512, 8), dtype=np.uint64)
assert L.bfpq_debug_stamps(ctypes.c_void_p(buf.ctypes.data)) == 0
s = buf[0][:256, :8].astype(np.int64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
rel = (s - t0) * 0.01          # us
names = ["start", "loop end", "barrier", "coarse", "flush/seg_win", "window stored", "ticket won (last wg)", "resolved (last wg)"]
print("hist launch: workgroups", len(s))
for i in range(6):
    print(f"  stamp {i} {names[i]:24s}: min {rel[:, i].min():7.2f}  median {np.median(rel[:, i]):7.2f}  max {rel[:, i].max():7.2f} us")
last = s[s[:, 7] > s[:, 5]]      # the workgroup whose stamps 6/7 are from this call
if len(last):
    r = (last[-1] - t0) * 0.01
    print(f"  last workgroup: window stored {r[5]:.2f}, ticket won {r[6]:.2f}, state written {r[7]:.2f} us")
    r2 = buf[2][:256, :6].astype(np.int64)
    r2 = r2[r2[:, 0] >= last[-1][6]]
    if len(r2):
        rr = (r2[-1] - t0) * 0.01
        print("  resolve step: entry %.2f | coarse bin known %.2f | slice offsets %.2f | slices in %.2f | tau %.2f | cut %.2f us" % tuple(rr))
