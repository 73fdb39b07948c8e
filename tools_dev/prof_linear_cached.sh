#!/bin/bash
# which kernels run in one cached BFPLinear forward (decode, 16 tokens)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_lc
rm -rf $OUT && mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
import quantization_sparsity_interplay_amd as bfpq
from quantization_sparsity_interplay_amd.bfp import bfp_ops
cfg = bfpq.BFPConfig.hbfp(4, 64, w_sparsity=True, N=2, M=4, sparsity_mode='structured', first='s').to_kwargs()
lin = bfp_ops.BFPLinear(11008, 4096, False, **dict(cfg)).to("cuda").to(torch.bfloat16)
lin.enable_weight_cache()
x = torch.randn(16, 11008, device="cuda").to(torch.bfloat16)
with torch.no_grad():
    for _ in range(30):
        lin(x)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/log.txt 2>&1 || true
python3 - <<PY
import csv, glob, os
f = max(glob.glob("$OUT/trace/*/*_kernel_stats.csv"), key=os.path.getmtime)
for r in list(csv.reader(open(f)))[1:10]: print(r[0][:90], r[1], r[3])
PY
