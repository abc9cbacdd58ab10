#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_v3
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT -o p1 -- python scratch/enc_only.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p2 -- python scratch/enc_only.py > $OUT/p2.log 2>&1
ls $OUT
python - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_v3"
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_qgemm" not in n and "k_attn_enc" not in n: continue
        key = n[n.find("k_"):n.find("(")][:28] + " grid=" + r["Grid_Size"]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT"): cnt[key] += 1
    for k, v in agg.items():
        n = max(cnt[k], 1)
        print(k, "calls", n, {c: round(x / n) for c, x in v.items()})
PY
