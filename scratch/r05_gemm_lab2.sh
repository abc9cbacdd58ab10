#!/bin/bash
# round 5, GEMM lab call 2: k_gemm4 (one wavefront per SIMD, 384 x 256 / 256 x 256 tiles, MFMA through inline asm with AGPR accumulators)
cd scratch/lab
L=./gemm8_lab
O=../../gpurun_out/r05_gemm8_b.txt
{
echo "== mlp.0 x8"
LAB_PROBE=1 LAB_SET=384:400 timeout 120 $L 0
LAB_SET=192:64,384:400,384:401,256:400,256:401 timeout 120 $L 0
echo "== mlp.0 x16"
LAB_SET=192:64,384:400,384:401 timeout 120 $L 6
echo "== qkv-shaped"
LAB_SET=288:32,384:400,384:401,256:401 timeout 120 $L 4
echo "== mlp.0 x1"
LAB_SET=384:400,256:401 timeout 120 $L 5
} > $O 2>&1
tail -60 $O
