#!/bin/bash
# ms per step of the driver's bench command under rt_set_tuning knobs: tools/knob_sweep.sh "mask_identity=1" "mk_w_leaf=24" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for t in "" "$@"; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-in-flight-check --per-step-dispatches 0 ${t:+--tune $t} > gpurun_out/knob.json 2> gpurun_out/knob.err || { echo "$t failed"; tail -3 gpurun_out/knob.err; continue; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/knob.json') if l.startswith('{')][-1]); r=d['roofline']
print(f\"{'$t' or 'default':32s} {d['ms_per_step']:8.2f} ms/step  trace {r['avg_launch_ms']:.3f} ms x {r['launches']:.0f}\", flush=True)"
done
