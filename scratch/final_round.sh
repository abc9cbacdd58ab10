#!/bin/bash
# end-of-round evidence: full GPU suite, default bench, 8-chunk bench, rocprof summaries (one chunk + 8 chunks), 2-rank rehearsal
TAG=${1:-r04c}
python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -3 gpurun_out/${TAG}_gpu_tests.log
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench.json
python bench.py --chunks 8 --no-config4 --no-cpu-baseline > gpurun_out/${TAG}_bench_chunks8.json 2> gpurun_out/${TAG}_bench_chunks8.err
bash profiles/collect.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1
bash profiles/collect.sh ${TAG}8 --chunks 8 > gpurun_out/${TAG}8_collect.log 2>&1
WMI_BENCH_REHEARSAL=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/${TAG}_rehearsal_2ranks_1gpu_gloo.log 2>&1
tail -c 400 gpurun_out/${TAG}_rehearsal_2ranks_1gpu_gloo.log
