#!/bin/bash
# where a beam-5 step of large-v3 q5_1 spends its 2.07 ms: GPU first-to-last command vs host wall, per decode() call
cd $GRAFT_REPO_ROOT
CASE=beam5 ONLY=2 WMI_DECODE_TRACE=1 python scratch/time_v3.py 2>&1 | grep -v "^W\|^E" | tail -45 > gpurun_out/r05_beam_trace.log
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_v3b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
CASE=beam5 ONLY=2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o v3 -- python scratch/time_v3.py > $OUT/run.log 2>&1
rm -f $OUT/*kernel_trace.csv
tail -42 gpurun_out/r05_beam_trace.log
