#!/bin/bash
python -m pytest tests/test_gpu_variants.py tests/test_gpu_quant.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 400 --no-config4 --no-cpu-baseline --stream-seconds 0 > gpurun_out/r05b_bench_quick.json 2> gpurun_out/r05b_bench_quick.err
python - <<PY
import json
d=json.load(open("gpurun_out/r05b_bench_quick.json"))
print({k:d[k] for k in ("value","ms_per_step","encode_ms","decode_ms_per_token","mel_ms")})
for t in d["decode_step_kernels"]: print(f'{t["kernel"][:60]:60s} avg={t["avg_us"]} body={t["body_us"]} gap={t["boundary_us"]}')
print(d["batch8"]["ms_per_call"], d["batch8"]["decode_ms"], d["batch16"]["ms_per_call"])
PY
