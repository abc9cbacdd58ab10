#!/bin/bash
# same-box A/B of two builds of the library: scratch/ab/lib_base.so vs scratch/ab/lib_new.so, alternating
for rep in 1 2; do for v in base new; do cp scratch/ab/lib_$v.so godot-whisper_amd/libwhisper_mi355.so; echo "== $v"; python ${1:-scratch/attn_probe.py} 2>&1 | tail -${2:-5}; done; done
