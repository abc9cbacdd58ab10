#!/bin/bash
# builds scratch/lab/gemm8_lab (run from anywhere)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../godot-whisper_amd/csrc gemm8_lab.hip -o gemm8_lab "$@"
