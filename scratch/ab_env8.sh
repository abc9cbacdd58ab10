#!/bin/bash
# A/B of an environment setting over the 8-chunk lock-step call: bash scratch/ab_env8.sh "VAR=VAL [VAR2=VAL2]" [reps]
SET="$1"; REPS=${2:-2}
for rep in $(seq $REPS); do for v in default knob; do
  if [ $v = knob ]; then env $SET python bench.py --chunks 8 --steps 240 --warmup 40 --no-config4 --no-cpu-baseline > /tmp/ab8_$v.json 2>/tmp/ab8_$v.err
  else python bench.py --chunks 8 --steps 240 --warmup 40 --no-config4 --no-cpu-baseline > /tmp/ab8_$v.json 2>/tmp/ab8_$v.err; fi
  python - "$v" "$SET" <<'PY'
import json,sys
v=sys.argv[1]
for l in open(f"/tmp/ab8_{v}.json"):
    if l.startswith("{"):
        d=json.loads(l); st=d.get("stage_ms") or {}
        print(v if v=="default" else sys.argv[2], "ms/call", d.get("ms_per_step"), "median", (d.get("headline_spread") or {}).get("median_ms"), {k:d.get(k) for k in ("mel_ms","encode_ms","decode_ms_per_token","segments_timestamps_ms")})
PY
done; done
