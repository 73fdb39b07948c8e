#!/bin/bash
# per-kernel times of the unstructured path (rocprofv3 kernel trace of bench_suite.py's cfg4 rows)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_u
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench_suite.py --only "cfg4 13B" --launches 20 --rounds 3 --out $OUT/suite_cfg4.json > $OUT/log.txt 2>&1 || true
cat $OUT/log.txt | tail -5
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-220
