#!/bin/bash
# PMC passes of the bench command reduced for one kernel (substring): tools/pmc_kernel.sh <tag> <kernel substring>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; kern=$2
out=gpurun_out/$tag; mkdir -p $out/pmc
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1   # never under rocprofv3
CMD="python3 bench.py --no-build --per-step-dispatches 0 --steps 10 --warmup 10 --spp 8 --tune probe=0,lanes=1 --cpu-seconds 0 --no-in-flight-check"
i=0
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc/pass_$i -- $CMD > $out/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/pmc_$i.log; }
  [ $i = 1 ] && grep '^{' $out/pmc_$i.log | tail -1 > $out/pmc/bench_pass.json
done
python3 tools/pmc_roofline.py $out/pmc "$kern" $out/counters_$kern.json "$CMD" > $out/pmc_roofline.log 2>&1
rm -rf $out/pmc
grep -v "^ *\"[A-Z_a-z]*\": [0-9]*,$" $out/counters_$kern.json | head -40
