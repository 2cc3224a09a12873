#!/bin/bash
# A/B of the implicit-GEMM operand path (GPU box): MF_IGEMM_LD 0 (LDS-DMA) / 1 (registers, two LDS stages) / 2 (registers, one stage) x
# MF_IGEMM_BK 32 / 64 on the per-op list of the MuseTalk step at batch $1 (default 8).  Outputs gpurun_out/ldab_<ld>_<bk>.txt
B=${1:-8}
cd $GRAFT_REPO_ROOT
for LD in 0 1 2; do for BK in 32 64; do
  MF_IGEMM_LD=$LD MF_IGEMM_BK=$BK python tools/mt_oplist.py $B all > gpurun_out/ldab_${LD}_${BK}.txt 2>&1
  echo "LD=$LD BK=$BK: $(tail -1 gpurun_out/ldab_${LD}_${BK}.txt)"
done; done
