#!/bin/bash
# per-kernel times of the fp32 unstructured path (rocprofv3 kernel trace of bench_suite.py's cfg4 f32 row)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_f32u
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench_suite.py --only "q_proj f32 HBFP4 + 50% unstructured" --launches 20 --rounds 3 --out $OUT/suite.json > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-200 > $OUT/kernel_stats.txt
cat $OUT/kernel_stats.txt
