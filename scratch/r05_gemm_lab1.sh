#!/bin/bash
# round 5, GEMM lab call 1: deferred epilogue stores (ks 2xx), 64-CU epilogue probe, one-round 288-row tiles, N = 512 shapes on 96-row tiles
cd scratch/lab
L=./gemm8_lab
O=../../gpurun_out/r05_gemm8_a.txt
{
echo "== mlp.0 x8: plain vs deferred stores, with probes"
LAB_PROBE=1 LAB_SET=192:64,192:264 timeout 120 $L 0
echo "== mlp.0 x8: other tiles"
LAB_SET=192:232,160:64,160:264,128:64,128:264,192:32 timeout 120 $L 0
echo "== cross x8"
LAB_SET=192:64,192:264,192:232,160:264,128:264 timeout 120 $L 3
echo "== mlp.0 x16"
LAB_SET=192:64,192:264 timeout 120 $L 6
echo "== qkv-shaped x8 (GELU epilogue): one-round 288-row tiles"
LAB_SET=288:32,192:64,192:264,160:64,160:264,96:64 timeout 120 $L 4
echo "== N = 512 shapes"
LAB_SET=96:64,96:32,128:64,192:64 timeout 120 $L 2
LAB_SET=96:64,96:32,128:64 timeout 120 $L 1
echo "== repeat for noise: mlp.0 x8 and cross x8"
LAB_SET=192:64,192:264 timeout 120 $L 0
LAB_SET=192:64,192:264 timeout 120 $L 3
} > $O 2>&1
tail -80 $O
