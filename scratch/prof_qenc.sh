#!/bin/bash
# rocprofv3 kernel stats of the large-v3 q5_1 encoder (scratch/enc_time.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_qenc
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o qenc -- python scratch/enc_time.py > $OUT/run.log 2>&1
tail -3 $OUT/run.log
rm -f $OUT/*kernel_trace.csv $OUT/*.db $OUT/*/*kernel_trace.csv $OUT/*/*.db
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_qenc/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>7}  avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Name"][:140]}')
PY
