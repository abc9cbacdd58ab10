#!/bin/bash
# in-kernel stamps of one 5-row beam step of large-v3 q5_1 (the 8th several-row call) beside a one-row greedy step's
cd $GRAFT_REPO_ROOT
CASE=beam5 ONLY=2 WMI_DECODE_STAMPS=8 python scratch/time_v3.py 2>&1 | grep "stamps n=" | head -30 > gpurun_out/r05_beam_stamps.log
python scratch/stamps_q.py 2>&1 | grep -v "^W\|^E" | head -24 >> gpurun_out/r05_beam_stamps.log
cat gpurun_out/r05_beam_stamps.log
