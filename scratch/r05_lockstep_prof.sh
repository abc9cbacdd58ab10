#!/bin/bash
# rocprofv3 kernel stats of the 8-chunk lock-step call (time_batch.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ls${TAG:-}; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
NB_ONLY=8 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ls -- python scratch/time_batch.py > $OUT/run.log 2>&1
rm -f $OUT/*kernel_trace.csv
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_ls" + os.environ.get("TAG", "") + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>7}  avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Name"][:140]}')
PY
