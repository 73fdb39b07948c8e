#!/bin/bash
# A/B of library builds on ONE box (boxes differ by 1-3 %): headline bench, variants interleaved, three rounds
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in "" $@; do
    if [ -z "$v" ]; then unset BFPQ_LIB; name=product; else export BFPQ_LIB=$R/tools_dev/_build/libbfpq_$v.so; name=$v; fi
    python $R/bench.py --steps 400 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step']*1000,3), round(d['roofline']['frac'],4))"
  done
done
