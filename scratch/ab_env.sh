#!/bin/bash
# generic A/B of one environment knob over the default bench: bash scratch/ab_env.sh VAR VALUE [reps]
VAR=$1; VAL=$2; REPS=${3:-2}
for rep in $(seq $REPS); do for v in default knob; do
  if [ $v = knob ]; then export $VAR=$VAL; else unset $VAR; fi
  python bench.py --steps 200 --no-config4 --no-cpu-baseline --stream-seconds 0 > /tmp/ab_$v.json 2>/tmp/ab_$v.err
  python - "$v" "$VAR=$VAL" <<'PY'
import json,sys
v=sys.argv[1]
for l in open(f"/tmp/ab_{v}.json"):
    if l.startswith("{"):
        d=json.loads(l)
        print(v if v=="default" else sys.argv[2], "attn1", d.get("attn_layer_us"), "attn8", d.get("attn_layer_batch8_us"), "enc1", d.get("encode_ms"), "b8enc", d.get("batch8",{}).get("encode_ms"), "b8ms", d.get("batch8",{}).get("ms_per_call"), "ms", d.get("ms_per_step"))
PY
done; done
