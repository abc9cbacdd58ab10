#!/bin/bash
# k_front (one launch for LN + q|k|v, self-attention, out projection) against the two launches: step chain and headline, one process each
cd "$(dirname "$0")/.."
for sh in base.en tiny.en; do
  echo "== $sh one launch";  SHAPE=$sh python scratch/step_chain.py 2>&1 | grep -i "whole step\|qkv\|self-attn"
  echo "== $sh two launches"; WMI_NO_FRONT=1 SHAPE=$sh python scratch/step_chain.py 2>&1 | grep -i "whole step\|qkv\|self-attn"
done
