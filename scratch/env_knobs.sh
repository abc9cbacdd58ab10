#!/bin/bash
# A/B of HIP runtime environment knobs on the decode-step latency (bench.py headline, one chunk): ms_per_step
run() { echo -n "$1: "; env $1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['decode_ms_per_token'], d['encode_ms'])"; }
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=1
run AMD_DIRECT_DISPATCH=0
run X=1
