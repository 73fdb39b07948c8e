#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_i
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench_suite.py --only "int8" --launches 10 --rounds 2 > $OUT/log.txt 2>&1 || true
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-170 | head -8
grep "^int8" $OUT/log.txt
