#!/bin/bash
# Profile collection recipe (run on the GPU box through gpurun from the repo root):
#   bash profiles/collect.sh r01e [extra bench.py arguments]
# 1. kernel trace + stats of the default bench  2-4. PMC passes (one counter group per pass, never combined
# with the sys/hip/hsa trace domains), each on a short bench run.  Outputs land in gpurun_out/prof_<tag>/;
# the summaries worth keeping are copied to profiles/ by profiles/summarise.py.
set -u
TAG=${1:-r01}
shift 2>/dev/null
EXTRA="$*"          # extra bench.py arguments, e.g. `bash profiles/collect.sh r04a8 --chunks 8` for the lock-step kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --steps 10 --warmup 2 --profile $EXTRA > $OUT/bench_trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o pmc_fetch -- python bench.py --steps 2 --warmup 1 --profile $EXTRA > $OUT/bench_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o pmc_write -- python bench.py --steps 2 --warmup 1 --profile $EXTRA > $OUT/bench_pmc_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT -o pmc_mfma -- python bench.py --steps 2 --warmup 1 --profile $EXTRA > $OUT/bench_pmc_mfma.log 2>&1
# calibration of SQ_VALU_MFMA_BUSY_CYCLES: the MFMA-only loop of scratch/lab/mfma_peak.hip (k_peak: back-to-back MFMAs, ~2.0 PFLOP/s =
# the pipe's own ceiling) under the same counter group gives busy / gui-active of a ~100 %-busy kernel; summarise.py divides by it
hipcc -O3 -std=c++17 --offload-arch=gfx950 scratch/lab/mfma_peak.hip -o /tmp/mfma_peak > $OUT/mfma_peak_build.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT -o pmc_mfma_calib -- /tmp/mfma_peak > $OUT/mfma_peak.log 2>&1
# condense on the box: the raw per-dispatch CSVs are tens of MB per pass and gpurun merges back at most 64 MiB
python profiles/summarise.py $TAG > $OUT/summary.txt 2>&1
cp profiles/${TAG}_* $OUT/ 2>/dev/null
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_counter_collection.csv' -delete; find $OUT -name '*_agent_info.csv' -delete
ls -la $OUT; du -sh $OUT
