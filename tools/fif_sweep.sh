#!/bin/bash
# bench.py with knob sets, one line each: tools/fif_sweep.sh "steps warmup fif tune" ...
for spec in "$@"; do
  set -- $spec
  python bench.py --steps $1 --warmup $2 --frames-in-flight $3 --cpu-seconds 0 --no-in-flight-check --per-step-dispatches 0 --tune $4 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('steps $1 fif $3 $4:',round(d['value']),'Mrays/s',round(d['ms_per_step'],2),'ms/step',d['config']['pipeline'][:12])"
done
