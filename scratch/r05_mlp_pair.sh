#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_states.py -x -q -m gpu 2>&1 | tail -5
TAGNAME=two-launches WMI_NO_MLP_PAIR=1 timeout 300 python scratch/headline_ab.py 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -3
TAGNAME=pair timeout 300 python scratch/headline_ab.py 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -3
SHAPE=tiny.en TAGNAME=tiny-two WMI_NO_MLP_PAIR=1 timeout 300 python scratch/headline_ab.py 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -2
SHAPE=tiny.en TAGNAME=tiny-pair timeout 300 python scratch/headline_ab.py 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -2
