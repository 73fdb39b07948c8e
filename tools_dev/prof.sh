#!/bin/bash
# rocprofv3 kernel trace + PMC passes for the bench command (run on the GPU box from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 > $OUT/bench_trace.json 2> $OUT/trace.err || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 --eager > $OUT/bench_fetch.json 2> $OUT/fetch.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 --eager > $OUT/bench_write.json 2> $OUT/write.err || true
find $OUT -name "*.csv" | head -20
