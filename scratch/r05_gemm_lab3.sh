#!/bin/bash
# round 5, GEMM lab call 3: is the faster K loop of the deferred-store build its unrolled K steps? (LAB_FLAGS=8192: unrolled, stores at once)
cd scratch/lab
L=./gemm8_lab
O=../../gpurun_out/r05_gemm8_c.txt
{
for rep in 1 2 3; do
echo "== mlp.0 x8, rep $rep"
LAB_SET=192:64,192:264 timeout 120 $L 0
LAB_FLAGS=8192 LAB_SET=192:264 timeout 120 $L 0
done
echo "== cross x8"
LAB_SET=192:64 timeout 120 $L 3
LAB_FLAGS=8192 LAB_SET=192:264 timeout 120 $L 3
echo "== mlp.0 x16"
LAB_SET=192:64 timeout 120 $L 6
LAB_FLAGS=8192 LAB_SET=192:264 timeout 120 $L 6
} > $O 2>&1
grep -v "^$" $O | tail -60
