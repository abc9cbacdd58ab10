#!/bin/bash
# A/B of two library builds on the same box: scratch/lib_old.so vs scratch/lib_new.so, lock-step batch figures
for rep in 1 2; do
for v in old new; do
  cp scratch/lib_$v.so godot-whisper_amd/libwhisper_mi355.so
  echo -n "$v: "; python bench.py --no-cpu-baseline --steps 24 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['batch8']; print(d['ms_per_step'], b['ms_per_call'], b['decode_ms'], b['encode_ms'], d.get('batch16',{}).get('ms_per_call'))"
done; done
