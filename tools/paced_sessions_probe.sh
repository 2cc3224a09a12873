# paced end-to-end search with S sessions per step at most (GPU box): S="8 10" bash tools/paced_sessions_probe.sh (MF_BENCH_ASR_INLINE=1: the Whisper call on the step stream)
cd $GRAFT_REPO_ROOT
for S in ${S:-8 10}; do
python bench.py --sessions $S --cpu-seconds 0 --profile-iters 1 --pmc-traffic 0 2>gpurun_out/paced_err_$S.txt > gpurun_out/paced_line_$S.json
python - <<PY
import json
d=json.load(open("gpurun_out/paced_line_$S.json"))
ms=d["multi_session"]; ps=ms["paced_sessions"]
print("sessions per step <= $S: free-running", ms["value"], ms["ms_per_step"], "e2e max", ps["max_sessions_sustained"], {k:ps["end_to_end"]["at_max"][k] for k in ("sessions","p50_ms","p99_ms","gpu_busy_frac","frames_per_s","sessions_per_step_mean")})
print([(t["sessions"],t["seconds"],t["sustained"],t["p99_ms"],t["frames_per_s"]) for t in ps["end_to_end"]["trials"]])
print(ps["step_ms_by_sessions_in_step"])
PY
done
