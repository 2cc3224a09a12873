#!/bin/bash
# re-measure the batch-8 (+ Wav2Lip batch-16) launch configurations with the library at HEAD and A/B the result against the shipped table on the same box
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; rm -f gpurun_out/retune.txt
MF_TUNE_CACHE=gpurun_out/retune.txt timeout 1200 python tools/tune_one_batch.py 8 --wav2lip 16 > gpurun_out/retune_log.txt 2>&1
wc -l gpurun_out/retune.txt
cp mere-fusion_amd/tune/gfx950.txt gpurun_out/gfx950_old.txt
# merged table: shipped rows, re-measured ones override
python - <<'P'
rows, order = {}, []
for f in ("mere-fusion_amd/tune/gfx950.txt", "gpurun_out/retune.txt"):
    for l in open(f):
        l = l.rstrip("\n")
        if not l.strip() or l.startswith("#"): continue
        k = l.split()[0]
        if k not in rows: order.append(k)
        rows[k] = l
open("gpurun_out/gfx950_retuned.txt", "w").write("\n".join(rows[k] for k in order) + "\n")
old = {l.split()[0]: l.split()[1:] for l in open("mere-fusion_amd/tune/gfx950.txt") if l.strip() and not l.startswith("#")}
ch = sum(1 for l in open("gpurun_out/retune.txt") if l.strip() and old.get(l.split()[0]) != l.split()[1:])
print("rows re-measured that differ from the shipped table:", ch)
P
OUT=gpurun_out/ab_tables.txt; : > $OUT
for rep in 1 2 3; do
for t in gpurun_out/gfx950_old.txt gpurun_out/gfx950_retuned.txt; do
  MF_TUNE_CACHE=$t timeout 300 python bench.py --batch 8 --extras 0 --cpu-seconds 0 --pmc-traffic 0 --profile-iters 0 --sessions 0 --steps 60 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t rep $rep musetalk b8', d['value'], d['ms_per_step'])" >> $OUT
  MF_TUNE_CACHE=$t timeout 300 python bench.py --workload wav2lip --extras 0 --cpu-seconds 0 --pmc-traffic 0 --profile-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t rep $rep wav2lip', d['value'], d['ms_per_step'])" >> $OUT
done
done
cat $OUT
