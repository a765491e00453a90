#!/bin/bash
# Builds of the library with the compiler's optional passes cut off after pass execution N (hipcc -mllvm -opt-bisect-limit=N),
# four at a time: tools/bisect_build.sh N1 N2 ...  ->  ray_tracer_amd/librt_amd_bis<N>.so   (the search for a miscompiling pass)
cd "$(dirname "$0")/.." || exit 1
flags=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIPFLAGS))")
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $flags -mllvm -opt-bisect-limit=$n ray_tracer_amd/csrc/scene.cpp ray_tracer_amd/csrc/rt_device.hip -o ray_tracer_amd/librt_amd_bis$n.so > /tmp/bis$n.log 2>&1; echo "built $n: $(grep -c 'BISECT: running' /tmp/bis$n.log) passes run, $(grep -c 'NOT running' /tmp/bis$n.log) skipped" ) &
done
wait
