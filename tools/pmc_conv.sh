#!/bin/bash
# PMC passes on one conv shape (GPU box): tools/pmc_conv.sh <tag> <conv_probe args...>
# Writes gpurun_out/pmc_<tag>_{sq1,sq2,tcc}.csv (kernel, counter values) via tools/rocprof_pmc_summary.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for pass in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
            "sq2:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
            "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "mem:FETCH_SIZE" "memw:WRITE_SIZE" "grbm:GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$name -o out -- python $R/tools/conv_probe.py "$@" --iters 5 > /tmp/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 /tmp/pmc_$name.log; }
  python $R/tools/rocprof_pmc_summary.py /tmp/pmc_$name > $R/gpurun_out/pmc_${tag}_$name.txt 2>&1
  tail -12 $R/gpurun_out/pmc_${tag}_$name.txt
done
