#!/bin/bash
for cfg in "$@"; do
  echo -n "$cfg :: "
  env $cfg timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | cut -c1-200
done
