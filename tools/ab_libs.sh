#!/bin/bash
# A/B of several builds of the library on one box, interleaved, two passes: ms per step of the driver's bench command.
# usage: tools/ab_libs.sh "<names: base pre ...>" [bench.py arguments]     (librt_amd_<name>.so from tools/build_variant.sh; "head" = the tree's)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
names=$1; shift
args=${*:---steps 20 --warmup 5}
mkdir -p gpurun_out
for pass in 1 2; do
  for n in $names; do
    lib=$PWD/ray_tracer_amd/librt_amd_$n.so; [ $n = head ] && lib=$PWD/ray_tracer_amd/librt_amd.so
    RT_AMD_LIB=$lib timeout -k 10 300 python3 bench.py $args --cpu-seconds 0 --no-in-flight-check --per-step-dispatches 0 > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n failed"; tail -3 gpurun_out/ab_$n.err; exit 1; }
    python3 - "$n" "$pass" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/ab_{sys.argv[1]}.json") if l.startswith("{")][-1])
r = d["roofline"]
print(f"pass {sys.argv[2]} {sys.argv[1]:10s} {d['ms_per_step']:8.2f} ms/step  {d['value']:8.1f} Mrays/s  trace {r['avg_launch_ms']:.3f} ms x {r['launches']:.0f}", flush=True)
PY
  done
done
