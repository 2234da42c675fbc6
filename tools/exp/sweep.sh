#!/bin/bash
# usage: sweep.sh "ENV=val ENV2=val" ...   -> one bench line summary per configuration
for cfg in "$@"; do
  echo -n "$cfg :: "
  env $cfg timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('rays/s', d['value'], 'cv', c['cost_volume_ms_per_frame'], 'dec', c['decoder_ms_per_frame'], 'enc', c['encoder_ms'], 'frac', d['roofline']['frac'])"
done
