#!/bin/bash
# attention output stores: wide (LDS-transposed 16-byte) vs narrow (8-byte) — bench fields attn_layer_us / attn_layer_batch8_us / encode_ms
for v in wide narrow wide narrow; do
  if [ $v = narrow ]; then export WMI_ATTN_NARROW_STORES=1; else unset WMI_ATTN_NARROW_STORES; fi
  python bench.py --steps 200 --no-config4 --no-cpu-baseline --stream-seconds 0 > /tmp/ab_$v.json 2>/tmp/ab_$v.err
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
for l in open(f"/tmp/ab_{v}.json"):
    if l.startswith("{"):
        d=json.loads(l)
        print(v, "attn1", d.get("attn_layer_us"), "attn8", d.get("attn_layer_batch8_us"), "enc1", d.get("encode_ms"), "b8enc", d.get("batch8",{}).get("encode_ms"), "ms", d.get("ms_per_step"))
PY
done
