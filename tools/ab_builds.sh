#!/bin/bash
# A/B of two builds of librt_amd on one box: ray_tracer_amd/librt_amd_old.so (built from another commit) vs the tree's.
# usage: tools/ab_builds.sh <ab_trace.py arguments...>
for lib in ray_tracer_amd/librt_amd_old.so ray_tracer_amd/librt_amd.so ray_tracer_amd/librt_amd_old.so ray_tracer_amd/librt_amd.so; do
  echo "== $lib"
  RT_AMD_LIB=$PWD/$lib timeout -k 10 250 python tools/ab_trace.py "$@" 2>&1 | tail -n +2 || exit 1
done
