#!/bin/bash
# adds a new tile width to the shipped tuning table without re-measuring everything: MF_TUNE_EXTEND=<bn> re-measures every eligible layer's entry against the
# tiles of that width only and appends the winners; then the old and the extended table are A/B'd on the same box.   usage: tools/extend_table.sh [bn]
BN=${1:-80}; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cp mere-fusion_amd/tune/gfx950.txt gpurun_out/gfx950_ext.txt
MF_TUNE_EXTEND=$BN MF_TUNE_CACHE=gpurun_out/gfx950_ext.txt MF_DEBUG=tune timeout 2400 python tools/make_tune_cache.py > gpurun_out/ext_log.txt 2>&1
echo "appended: $(( $(wc -l < gpurun_out/gfx950_ext.txt) - $(wc -l < mere-fusion_amd/tune/gfx950.txt) )) rows"; tail -3 gpurun_out/ext_log.txt
cp mere-fusion_amd/tune/gfx950.txt gpurun_out/gfx950_old.txt
tools/ab_tables.sh gpurun_out/gfx950_old.txt gpurun_out/gfx950_ext.txt
