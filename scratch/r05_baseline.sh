#!/bin/bash
# round 5, first call: full GPU suite (with parity margins) + default bench line
TAG=${1:-r05a}
WMI_MARGINS_OUT=gpurun_out/${TAG}_parity_margins.json python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -40 gpurun_out/${TAG}_gpu_tests.log
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.err; python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","encode_ms","decode_ms_per_token")})
for k in ("encoder_gemm_mfma_utilisation","encoder_gemm_mfma_utilisation_batch8"):
    u=d.get(k)
    if u:
        print(k, u["achieved"], u["frac"], u["gemm_us"])
        for p in u["per_shape"]: print("   ", p)
print(d.get("encoder_gemm_mfma_utilisation_error"), d.get("roofline_error"))
print(d.get("batch8"))
PY
