# Is a conv kernel power-limited?  Runs ONE layer's kernel back to back (mf_conv2d_time, ITERS launches) and samples rocm-smi (power, sclk) meanwhile.
# usage (GPU box): bash tools/power_probe.sh "<env>" cin cout hw [batch] [precision]
cd $GRAFT_REPO_ROOT
v=$1; c1=${2:-512}; c2=${3:-512}; hw=${4:-64}; b=${5:-8}; pr=${6:-f16q}
env $v python tools/conv_probe.py --cin $c1 --cout $c2 --hw $hw --batch $b --residual 0 --precision $pr --iters 3 --alone-iters ${ITERS:-30000} > /tmp/pp.log 2>&1 &
PID=$!
sleep ${WARM:-6}
for k in 1 2 3; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done
wait $PID
grep -E "alone" /tmp/pp.log | cut -c1-160
