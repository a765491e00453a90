#!/bin/bash
# per-GPU efficiency on a 1/8-height tile as a function of resident blocks per CU
for b in "$@"; do
  python bench.py --steps 3 --warmup 1 --spp 8 --height 136 --cpu-seconds 0 --tune blocks_per_cu=$b 2>&1 | tail -1 > /tmp/line.json
  python3 -c "
import json; d=json.load(open('/tmp/line.json')); print('blocks_per_cu=$b', d['config']['height'], round(d['value']), round(d['unique_mrays_per_s']), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],3))"
done
