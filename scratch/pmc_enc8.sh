#!/bin/bash
# PMC split of the lock-step (8 x 30 s, M = 12 000) encoder kernels: where the wave cycles go (MI355X_MICROARCH.md §PMC:
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES), MFMA busy, LDS conflicts, HBM fetch.  Separate passes, kernel trace only.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_enc8
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT -o p1 -- python scratch/enc8.py > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p2 -- python scratch/enc8.py > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o p3 -- python scratch/enc8.py > $OUT/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p0 -- python scratch/enc8.py > $OUT/p0.log 2>&1
python - <<'PY' | tee $OUT/summary.txt
import csv, glob, os, collections, re
def short(n):
    m = re.search(r"(k_\w+)(<[^>]*>)?", n); return (m.group(1) + (m.group(2) or "")) if m else n[:40]
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_enc8"
dur = {}
for r in csv.DictReader(open(glob.glob(out + "/**/p0_kernel_stats.csv", recursive=True)[0])):
    n = r["Name"]; dur[short(n)] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_gemm" not in n and "k_attn_enc" not in n and "k_layernorm" not in n: continue
        if int(r["Grid_Size"]) < 150000: continue                      # the M = 12 000 launches only
        key = short(n) + " grid=" + r["Grid_Size"]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key].add(r["Dispatch_Id"])
    for k, v in sorted(agg.items()):
        n = max(len(cnt[k]), 1)
        print(k, "launches", n, {c: round(x / n) for c, x in v.items()})
print({k: v for k, v in dur.items() if v[1] > 20})
PY
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*.db' -delete
