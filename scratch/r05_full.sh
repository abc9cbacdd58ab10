#!/bin/bash
TAG=${1:-r05d}
WMI_MARGINS_OUT=gpurun_out/${TAG}_parity_margins.json python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -45 gpurun_out/${TAG}_gpu_tests.log
