#!/bin/bash
# A/B of an environment switch on the same box: $1 = VAR=value for the B side
for rep in 1 2 3; do
for v in X=0 "$1"; do
  echo -n "$v: "; env $v python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['batch8']; print(d['ms_per_step'], d['encode_ms'], d['decode_ms_per_token'], b['ms_per_call'], b['segments_timestamps_ms'], b['mel_envelope_ms'], d['batch16']['ms_per_call'])"
done; done
