#!/bin/bash
# same-box A/B of tuning tables (MF_TUNE_CACHE) on the MuseTalk (batch 8, 64) and Wav2Lip steps   usage: tools/ab_tables.sh tableA tableB ...
R=$GRAFT_REPO_ROOT; cd $R; OUT=gpurun_out/ab_tables.txt; : > $OUT
for rep in 1 2 3; do
for t in "$@"; do
  for b in 8 64; do
    MF_TUNE_CACHE=$t timeout 300 python bench.py --batch $b --extras 0 --cpu-seconds 0 --pmc-traffic 0 --profile-iters 0 --sessions 0 --steps $([ $b = 8 ] && echo 60 || echo 12) --warmup $([ $b = 8 ] && echo 8 || echo 3) 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t rep $rep musetalk b$b', d['value'], d['ms_per_step'])" >> $OUT
  done
  MF_TUNE_CACHE=$t timeout 300 python bench.py --workload wav2lip --extras 0 --cpu-seconds 0 --pmc-traffic 0 --profile-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t rep $rep wav2lip', d['value'], d['ms_per_step'])" >> $OUT
done
done
cat $OUT
