#!/bin/bash
# The judged evidence of a round, in one go: bench line, rocprofv3 kernel stats of the same command, and the two PMC
# passes for HBM-side traffic (FETCH_SIZE, WRITE_SIZE; separate runs). usage: tools/profile_round.sh <tag>   (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1
out=gpurun_out/$tag
mkdir -p $out
CMD="python3 bench.py --steps 3 --warmup 1 --spp 8"
timeout -k 10 400 $CMD > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
tail -1 $out/bench.log > $out/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD --cpu-seconds 0 > $out/bench_under_rocprof.log 2>&1 || { tail -5 $out/bench_under_rocprof.log; exit 1; }
grep '^{' $out/bench_under_rocprof.log | tail -1 > $out/bench_under_rocprof.json
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $CMD --cpu-seconds 0 --no-profile > $out/fetch.log 2>&1 || { tail -5 $out/fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $CMD --cpu-seconds 0 --no-profile > $out/write.log 2>&1 || { tail -5 $out/write.log; exit 1; }
kern=$(python3 -c "import json; print('k_render_fused' if 'fused' in json.load(open('$out/bench.json'))['config']['pipeline'] else 'k_trace_pw')")
python tools/pmc_traffic.py $(find $out/fetch -name '*counter_collection.csv' | head -1) $(find $out/write -name '*counter_collection.csv' | head -1) $kern $out/traffic_$kern.json "python3 bench.py --steps 3 --warmup 1 --spp 8 (Sponza 1920x1080)"
rm -rf $out/stats $out/fetch $out/write
cat $out/bench.json | cut -c1-300; head -5 $out/kernel_stats.csv; cat $out/traffic_$kern.json
