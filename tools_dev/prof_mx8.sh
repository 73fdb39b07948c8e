#!/bin/bash
# per-kernel time and SQ / L2 counters of the prefill consumer (bfpq_hbfp_linear_mx8) on the gate_proj shape, 2048 tokens
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_mx
rm -rf $OUT && mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
from quantization_sparsity_interplay_amd import native
from quantization_sparsity_interplay_amd.bfp import bfp_ops
native.SHARE_ACT_IMAGE = False
w = (torch.randn(11008, 4096, device="cuda") * 0.02).to(torch.bfloat16)
x = torch.randn(2048, 4096, device="cuda").to(torch.bfloat16)
pw = bfp_ops.PackedBFP.quantize(w, 3, 64, N=2, M=4)
for _ in range(20):
    pw.linear(x, x_mant_bits=3)
    torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/log.txt 2>&1 || true
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-200 | head -12
rocprofv3 --pmc SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- python3 $OUT/run.py > $OUT/p1.log 2>&1 || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- python3 $OUT/run.py > $OUT/p2.log 2>&1 || true
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python3 $OUT/run.py > $OUT/p3.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p4 -- python3 $OUT/run.py > $OUT/p4.log 2>&1 || true
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2","p3","p4"):
    fs=glob.glob("$OUT/%s/*/*_counter_collection.csv"%p)
    if not fs: print(p,"no file"); continue
    v=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_mx8_gemm" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,x in v.items(): print(p,k,len(x),sum(x)/len(x))
PY
