#!/bin/bash
# rocprofv3 kernel stats of the large-v3 q5_1 path (greedy + beam, max_tokens 16)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_v3
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ONLY=${ONLY:-2} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o v3 -- python scratch/time_v3.py > $OUT/run.log 2>&1
tail -5 $OUT/run.log
rm -f $OUT/*kernel_trace.csv $OUT/*.db; ls $OUT
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_v3/**/*kernel_stats.csv", recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:30]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>7}  avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Name"][:150]}')
PY
