#!/bin/bash
# end-of-round evidence: default bench, 8-chunk bench, rocprof summaries (one chunk + 8 chunks), 2-rank rehearsal
TAG=${1:-r05e}
# (the profiles first: bench.py takes `traffic` from the committed PMC summaries)
bash profiles/collect.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1
bash profiles/collect.sh ${TAG}8 --chunks 8 > gpurun_out/${TAG}8_collect.log 2>&1
cp gpurun_out/prof_${TAG}/${TAG}_* gpurun_out/prof_${TAG}8/${TAG}8_* profiles/ 2>/dev/null
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.err
python bench.py --chunks 8 --no-config4 --no-cpu-baseline > gpurun_out/${TAG}_bench_chunks8.json 2> gpurun_out/${TAG}_bench_chunks8.err
WMI_BENCH_REHEARSAL=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 --no-config4 --no-cpu-baseline > gpurun_out/${TAG}_rehearsal_2ranks_1gpu_gloo.log 2>&1
tail -c 600 gpurun_out/${TAG}_rehearsal_2ranks_1gpu_gloo.log
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","encode_ms","decode_ms_per_token","mel_ms")})
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_us"])
for k in ("encoder_gemm_mfma_utilisation","encoder_gemm_mfma_utilisation_batch8"):
    u=d.get(k); print(k, u and (u["achieved"], u["frac"], u["gemm_us"]))
print(d["batch8"]); print(d["batch16"])
c=d["config4_large_v3_q5_1_beam5"]; print({k:c[k] for k in ("beam5","greedy","lockstep8_greedy","beam5_8chunks_replicas","roofline_decode_step")})
PY
ls profiles/${TAG}* 2>/dev/null
