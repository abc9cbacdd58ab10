#!/bin/bash
for v in base A B; do cp scratch/ab/lib_$v.so godot-whisper_amd/libwhisper_mi355.so; echo "== $v"; ONLY=1 python scratch/time_v3.py 2>&1 | tail -2; done
cp scratch/ab/lib_base.so godot-whisper_amd/libwhisper_mi355.so
