#!/bin/bash
for v in "WMI_NO_PREFETCH=1" "WMI_QROWS_PF_SLEEP=0" "WMI_QROWS_PF_SLEEP=3" "WMI_QROWS_PF_SLEEP=6"; do
  echo "=== $v"
  env $v python scratch/stamps_q.py 2>&1 | head -12
done > gpurun_out/r05_q_stamps_probe2.txt 2>&1
cat gpurun_out/r05_q_stamps_probe2.txt
