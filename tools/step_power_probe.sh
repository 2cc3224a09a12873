# Average power / shader clock while the MuseTalk step runs back to back (GPU box): bash tools/step_power_probe.sh [batch] [steps]
cd $GRAFT_REPO_ROOT
b=${1:-8}; st=${2:-600}
python bench.py --workload musetalk --batch $b --extras 0 --cpu-seconds 0 --profile-iters 0 --pmc-traffic 0 --sessions 0 --steps $st --warmup 8 > /tmp/sp.log 2>/dev/null &
PID=$!
sleep ${WARM:-25}
for k in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.4
done
wait $PID
python -c "import json; d=json.loads(open('/tmp/sp.log').readline()); print('step', d['ms_per_step'], 'ms', d['value'], 'frames/s')"
