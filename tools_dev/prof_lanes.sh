#!/bin/bash
# kernel trace of the whole-model passes over lanes: start / end of every launch, so that the overlap of the lanes can be read off
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_lanes
rm -rf $OUT && mkdir -p $OUT
i=0
for CASE in "cfg3 LLaMA-7B" "cfg4 LLaMA-13B"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv -d $OUT/t$i -- python3 $R/bench_suite.py --models-only --no-list-graph --only "$CASE" --rounds 3 --out $OUT/m$i.json > $OUT/m$i.log 2>&1
  echo "case $i rc=$?" >> $OUT/progress.txt
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/t$i/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "bfpq" in r["Kernel_Name"] or "k_select" in r["Kernel_Name"] or "k_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass: the final len/4 rows (1 warm-up + 3 timed passes in the process... the list's own graph replays included)
n = len(rows)
print("launches in the trace:", n)
per = 224 if "$CASE".startswith("cfg3") else 560          # launches of ONE pass (cfg4: two per tensor)
print("passes in the trace:", n / per)
last = rows[-per:]
t0 = min(int(r["Start_Timestamp"]) for r in last); t1 = max(int(r["End_Timestamp"]) for r in last)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
names = collections.Counter(r["Kernel_Name"][:60] for r in last)
durs = collections.defaultdict(list)
for r in last: durs[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
queues = collections.Counter(r.get("Queue_Id", "?") for r in last)
print("CASE $CASE: last pass = %d launches, span %.1f us, sum of kernel durations %.1f us, mean concurrency %.2f" % (len(last), (t1 - t0) / 1e3, busy / 1e3, busy / (t1 - t0)))
for k, v in durs.items(): print("   %-62s x%4d  mean %.1f us  min %.1f  max %.1f" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
print("   queues:", dict(queues))
PY
done > $OUT/summary.txt 2>&1
cat $OUT/progress.txt $OUT/summary.txt; tail -n 2 $OUT/m1.log; tail -n 2 $OUT/m2.log
rm -rf $OUT/t1 $OUT/t2
