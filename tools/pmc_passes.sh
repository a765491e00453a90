#!/bin/bash
# PMC passes over one short bench run each (<= 3 counters of a block per pass; every pass under its own timeout).
# usage: tools/pmc_passes.sh <out-prefix> "<counters pass 1>" "<counters pass 2>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
pfx=$1; shift
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${pfx}_$i -- python3 bench.py --no-build --per-step-dispatches 0 --steps 1 --warmup 0 --spp 2 --cpu-seconds 0 --no-profile ${BENCH_ARGS:-} > gpurun_out/${pfx}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/${pfx}_$i.log; exit 1; }
  python tools/pmc_sum.py gpurun_out/${pfx}_$i ${KERNEL:-k_trace_pw} | tail -n +3
done
