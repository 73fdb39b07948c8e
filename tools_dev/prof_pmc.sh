#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_pmc
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 --eager > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -- python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 --eager > $OUT/p2.json 2> $OUT/p2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/p3 -- python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 --eager > $OUT/p3.json 2> $OUT/p3.err
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2","p3"):
    fs=glob.glob("$OUT/%s/*/*_counter_collection.csv"%p)
    if not fs: print(p,"no file"); continue
    v=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_fused_flat" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,x in v.items(): print(p,k,len(x),sum(x)/len(x))
PY
tail -3 $OUT/p1.err
