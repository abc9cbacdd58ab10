#!/bin/bash
# round 5, GEMM lab call 5: q and k thirds of the big q|k|v grid in the transposed orientation (full-line stores) on k_gemm
cd scratch/lab
L=./gemm8_lab
O=../../gpurun_out/r05_gemm8_e.txt
{
for rep in 1 2; do
echo "-- q,k thirds transposed + wide stores (new default)"
LAB_SET=288:32 timeout 120 $L 8
echo "-- WMI_GEMM_QKV_ROWS=1 (round-3 choice)"
WMI_GEMM_QKV_ROWS=1 LAB_SET=288:32 timeout 120 $L 8
done
} > $O 2>&1
grep -v "^$" $O | tail -20
