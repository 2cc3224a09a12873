#!/bin/bash
# Round-end evidence run (GPU box): full test tier, smoke(), the default bench line, rocprofv3 kernel stats of the SAME timed steps (--extras 0
# --profile-iters 0: nothing but warm-up + timed steps runs under the profiler, so per-launch averages are the bench's own), for the three workloads at
# their BASELINE.json batch and for MuseTalk at the operating point (8 sessions x 8 frames = batch 64), plus the MFMA-utilisation and HBM counters of the
# MuseTalk steps at both batch sizes (each counter set in its own pass, --kernel-trace the only trace domain beside --pmc).
# Launch configurations: the tuning table shipped beside the library -- what a deployment runs.  Outputs under gpurun_out/ (copied into profiles/ afterwards).
#   usage: tools/round_profile.sh [tag] [parts: tests,bench,stats,b64,pmc]
TAG=${1:-r06}; PARTS=${2:-tests,bench,stats,b64,pmc}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
has() { case ",$PARTS," in *",$1,"*) return 0;; esac; return 1; }
if has tests; then
  timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_musetalk_full.py 2>&1 | tail -4 > gpurun_out/${TAG}_gpu_tests.txt
  timeout 900 python -m pytest tests/test_musetalk_full.py tests/test_ernerf_reference_kernels.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/${TAG}_full_size_parity.txt
  timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke > gpurun_out/${TAG}_smoke.txt
fi
if has bench; then
  python bench.py --dump-layers gpurun_out/${TAG}_layers.json > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_err.txt
  cp bench_detail.json gpurun_out/${TAG}_bench_detail.json
  python tools/unet_floor_table.py gpurun_out/${TAG}_layers.json > gpurun_out/${TAG}_unet_b8_floor.md 2>> gpurun_out/${TAG}_bench_err.txt
fi
cd /tmp && export TMPDIR=/tmp
if has stats; then
  for WL in musetalk wav2lip ernerf; do
    rm -rf /tmp/prof_$WL
    STEPS=20; [ $WL = wav2lip ] && STEPS=100; [ $WL = ernerf ] && STEPS=50
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$WL -o $WL -- python $R/bench.py --workload $WL --steps $STEPS --warmup 5 --extras 0 --cpu-seconds 0 --profile-iters 0 > /tmp/prof_$WL.log 2>&1
    python $R/tools/rocprof_summary.py $(find /tmp/prof_$WL -name "*.db" | head -1) > $R/gpurun_out/${TAG}_kernel_stats_$WL.md 2>&1
  done
fi
if has b64; then PMC=$(has pmc && echo 1 || echo 0) bash $R/tools/b64_profile.sh $TAG 64 > /dev/null 2>&1; fi
if has pmc; then PMC=1 bash $R/tools/b64_profile.sh ${TAG} 8 > /dev/null 2>&1; fi
cd $R
cat gpurun_out/${TAG}_gpu_tests.txt gpurun_out/${TAG}_smoke.txt 2>/dev/null; cut -c1-1200 gpurun_out/${TAG}_bench_line.json 2>/dev/null; head -14 gpurun_out/${TAG}_kernel_stats_musetalk.md 2>/dev/null | cut -c1-180
