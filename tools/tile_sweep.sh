#!/bin/bash
# per-GPU throughput on 1/N-height tiles (what rank 0 of N GPUs renders), default knobs
for h in 1080 540 270 135; do
  python bench.py --steps 3 --warmup 1 --spp ${SPP:-8} --height $h --cpu-seconds 0 "$@" 2>&1 | tail -1 > /tmp/line.json
  python3 -c "
import json; d=json.load(open('/tmp/line.json')); print(d['config']['height'], d['config']['pipeline'][:12], 'Mrays/s', round(d['value']), 'unique', round(d['unique_mrays_per_s']), 'ms/step', round(d['ms_per_step'],1))"
done
