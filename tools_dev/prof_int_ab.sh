#!/bin/bash
# per-kernel times of the 'int' per-column path variants (tools_dev/ab_int.py under the kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_iab
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools_dev/ab_int.py "$@" > $OUT/log.txt 2>&1 || true
cat $OUT/log.txt | tail -8
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-220
