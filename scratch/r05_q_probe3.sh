#!/bin/bash
python scratch/stamps_q.py 2>&1 | head -14 > gpurun_out/r05_q_stamps_probe3.txt
ONLY=1 python scratch/time_v3.py >> gpurun_out/r05_q_stamps_probe3.txt 2>&1
cat gpurun_out/r05_q_stamps_probe3.txt
