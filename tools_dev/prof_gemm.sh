#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_g
rm -rf $OUT && mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
from quantization_sparsity_interplay_amd.bfp import bfp_ops
w = (torch.randn(4096, 11008, device="cuda") * 0.02).to(torch.bfloat16)
x = torch.randn(16, 11008, device="cuda").to(torch.bfloat16)
pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
for _ in range(30):
    pw.linear_decode(x)
    torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/log.txt 2>&1 || true
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-160 | head -14
