#!/bin/bash
# SQ / TCC counters of the kernels behind one or more bench_suite.py rows (rocprofv3 --pmc, one counter group per pass,
# no trace flags beside it; the program itself after "--").  usage: prof_pmc_cases.sh <tag> "<row substring>" [...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G2="SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
G3="FETCH_SIZE GRBM_GUI_ACTIVE"
G4="WRITE_SIZE GRBM_COUNT"
G5="SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"
i=0
for CASE in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c${i}_trace -- python3 $R/bench_suite.py --only "$CASE" --launches 20 --rounds 3 --out $OUT/c${i}_trace.json > $OUT/c${i}_trace.log 2>&1
  cat $OUT/c${i}_trace/*/*_kernel_stats.csv | cut -c1-200 > $OUT/c${i}_kernel_stats.txt
  for g in 1 2 3 4 5; do
    eval CNT=\$G$g
    rocprofv3 --pmc $CNT --output-format csv -d $OUT/c${i}_p$g -- python3 $R/bench_suite.py --only "$CASE" --launches 6 --rounds 2 --eager --out $OUT/c${i}_p$g.json > $OUT/c${i}_p$g.log 2>&1
    echo "case $i pass $g rc=$?" >> $OUT/progress.txt
  done
done
python3 - <<PY
import csv,glob,collections,json
out={}
for d in sorted(glob.glob("$OUT/c*_p*/")):
    fs=glob.glob(d+"*/*_counter_collection.csv")
    if not fs: print(d,"no file"); continue
    v=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        kn=r["Kernel_Name"]
        if kn.startswith("void at::") or "elementwise" in kn or "distribution" in kn: continue
        v[(kn[:150],r["Counter_Name"])].append(float(r["Counter_Value"]))
    tag=d.rstrip("/").split("/")[-1]
    for (kn,c),x in sorted(v.items()):
        x=x[len(x)//3:]          # drop the warm-up dispatches
        out.setdefault(tag.split("_")[0],{}).setdefault(kn,{})[c]=[len(x),sum(x)/len(x)]
json.dump(out,open("$OUT/summary.json","w"),indent=1)
for c,ks in out.items():
    for kn,cs in ks.items():
        print(c,kn[:110])
        for n,(cnt,mean) in cs.items(): print("     %-28s %4d  %.1f"%(n,cnt,mean))
PY
