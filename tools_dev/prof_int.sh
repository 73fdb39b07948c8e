#!/bin/bash
# per-kernel times of the 'int' format rows of bench_suite.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_i
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench_suite.py --only "int8 per-column" --launches 20 --rounds 3 --out $OUT/suite_int.json > $OUT/log.txt 2>&1 || true
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-200
