#!/bin/bash
# lock-step step after: no scratch in k_rows_mfma, two tiles in flight in the vocabulary projection
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "lockstep or full_batch or batch or rows" 2>&1 | tail -4
python scratch/time_batch.py 2>&1 | grep -v "^W\|^E" | tail -6
WMI_ROWS_BLOCKS=768 python scratch/time_batch.py 2>&1 | grep -v "^W\|^E" | tail -3
