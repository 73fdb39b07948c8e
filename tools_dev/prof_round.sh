#!/bin/bash
# round summary on ONE box: the suite with the whole-model rows, the bench line, and the kernel trace of the bench's timed region
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
OUT=$R/gpurun_out/round_$TAG
rm -rf $OUT && mkdir -p $OUT
python3 $R/bench_suite.py --models --out $OUT/suite.json > $OUT/suite.log 2>&1
echo "suite rc=$?" >> $OUT/progress.txt
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --headline-only > $OUT/bench_traced.json 2> $OUT/trace.err
echo "trace rc=$?" >> $OUT/progress.txt
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-220 > $OUT/kernel_stats_headline.csv
rm -rf $OUT/trace
cat $OUT/progress.txt; tail -32 $OUT/suite.log | cut -c1-160; cat $OUT/kernel_stats_headline.csv | head -5
