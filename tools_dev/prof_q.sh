#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_q
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench_suite.py --only "unstructured q" --launches 20 --rounds 3 --eager --out $OUT/suite.json > $OUT/log.txt 2>&1 || true
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-180 | head -8
