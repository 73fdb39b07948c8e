#!/usr/bin/env python3
"""Phase timing of the unstructured path's histogram / resolve launches from in-kernel stamps (100 MHz clock).
Build the instrumented library first:
    make -C quantization-sparsity-interplay_amd/csrc -j4 EXTRA=-DBFPQ_STAMPS OUT=$PWD/tools_dev/_build/libbfpq_stamps.so OBJDIR=$PWD/tools_dev/_build/obj
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
x = (torch.randn(rows, cols, generator=torch.Generator().manual_seed(1)) * 0.02).to(torch.bfloat16).cuda()
L = native.load_library()
ws = bfp_ops._workspace(x.device)
for _ in range(5):
    native.select_threshold(x, x.numel() // 2, ws)
torch.cuda.synchronize()
buf = np.zeros((3, 512, 8), dtype=np.uint64)
assert L.bfpq_debug_stamps(ctypes.c_void_p(buf.ctypes.data)) == 0
for kern, name, n in ((0, "hist", 6), (1, "resolve", 5)):
    s = buf[kern][:256, :n].astype(np.int64)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    rel = (s - t0) * 0.01          # us
    print(name, "workgroups", len(s))
    for i in range(n):
        print(f"  stamp {i}: min {rel[:, i].min():7.2f}  median {np.median(rel[:, i]):7.2f}  max {rel[:, i].max():7.2f} us")
