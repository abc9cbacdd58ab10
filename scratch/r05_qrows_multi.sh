#!/bin/bash
# several-rows prologue of k_qrows (beam step / lock-step rows of block-quantised models): parity first, then timings
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_quant.py tests/test_gpu_large_v3.py tests/test_gpu_variants.py -q -m gpu -x 2>&1 | tail -8
CASE=beam5 ONLY=2 WMI_DECODE_TRACE=1 python scratch/time_v3.py 2>&1 | grep -v "^W\|^E" | tail -8
ONLY=2 BATCH=8 python scratch/time_v3.py 2>&1 | grep -v "^W\|^E" | tail -5
