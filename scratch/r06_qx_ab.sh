#!/bin/bash
# the two forms of the quantised models' cross-attention, one process each; compares the RESULT lines
cd "$(dirname "$0")/.."
python scratch/r06_qx_ab.py > gpurun_out/qx_one.log 2>&1
WMI_Q_XATTN_TWO_LAUNCHES=1 python scratch/r06_qx_ab.py > gpurun_out/qx_two.log 2>&1
grep -v RESULT gpurun_out/qx_one.log | tail -20; echo ---- two launches; grep -v RESULT gpurun_out/qx_two.log | tail -20
python - <<'P'
import json
a = [l for l in open('gpurun_out/qx_one.log') if l.startswith('RESULT')]; b = [l for l in open('gpurun_out/qx_two.log') if l.startswith('RESULT')]
if not a or not b: print('missing RESULT'); raise SystemExit(1)
a = json.loads(a[-1][6:]); b = json.loads(b[-1][6:])
for k in a:
    for kk in a[k]:
        print(k, kk, 'IDENTICAL' if a[k][kk] == b[k][kk] else 'DIFFERENT', len(a[k][kk][0]))
P
