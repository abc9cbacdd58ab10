#!/bin/bash
python scratch/stamps_q.py > gpurun_out/r05_q_stamps_probe.txt 2>&1
ONLY=2 python scratch/time_v3.py > gpurun_out/r05_q_time_a.txt 2>&1
tail -40 gpurun_out/r05_q_stamps_probe.txt; cat gpurun_out/r05_q_time_a.txt
