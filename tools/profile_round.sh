#!/bin/bash
# The judged evidence of a round, in one go (on the GPU box): bench line, rocprofv3 kernel stats of the same command, and
# the PMC passes (HBM-side traffic, TA / VALU / L1 counters) reduced to one JSON that bench.py quotes in `roofline`.
# usage: tools/profile_round.sh <tag> [extra bench.py arguments]      results under gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1   # before any profiled run: no compiler or make process is ever started under rocprofv3 (--no-build below)
CMD="python3 bench.py --no-build --steps 10 --warmup 10 --spp 8 --tune probe=0 --per-step-dispatches 0 $*"   # one warm-up group and one timed group of ten steps in flight (what the default bench.py run does twice); no ray-cost probe (its small launches would sit in the per-kernel averages; the warm-up measures the ray cost instead)
timeout -k 10 400 ${CMD/--per-step-dispatches 0/} > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
grep '^{' $out/bench.log | tail -1 > $out/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD --cpu-seconds 0 --no-in-flight-check > $out/bench_under_rocprof.log 2>&1 || { tail -5 $out/bench_under_rocprof.log; exit 1; }
grep '^{' $out/bench_under_rocprof.log | tail -1 > $out/bench_under_rocprof.json
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
kern=$(python3 -c "import json; print('k_render_fused' if 'fused' in json.load(open('$out/bench.json'))['config']['pipeline'] else 'k_trace_pw')")
# one pass per counter group: TCC (FETCH_SIZE alone, WRITE_SIZE alone), SQ, TA/TCP; GRBM rides along
i=0
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  # counter passes: the dispatch in ONE part ("lanes" 1). rocprofv3 serialises kernels while it collects counters, so the three
  # overlapping parts of a normal dispatch would each be measured alone on their half-size grids; one part is the kernel alone
  # on the whole GPU, which is what its unit utilisations are meant to say. (Look-ups and traffic per ray do not depend on it.)
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc/pass_$i -- ${CMD/probe=0/probe=0,lanes=1} --cpu-seconds 0 --no-in-flight-check > $out/pmc_$i.log 2>&1 || { echo "pmc pass $i ($set) failed"; tail -5 $out/pmc_$i.log; }
  [ $i = 1 ] && grep '^{' $out/pmc_$i.log | tail -1 > $out/pmc/bench_pass.json
done
python3 tools/pmc_roofline.py $out/pmc $kern $out/counters_$kern.json "$CMD (PMC passes: --cpu-seconds 0)" > $out/pmc_roofline.log 2>&1 || tail -5 $out/pmc_roofline.log
rm -rf $out/stats $out/pmc
cut -c1-400 $out/bench.json; head -4 $out/kernel_stats.csv; cat $out/counters_$kern.json
