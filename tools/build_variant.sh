#!/bin/bash
# A second build of the library with extra compiler flags, for A/B runs on one box (tools/ab_libs.sh):
#   tools/build_variant.sh <name> [-DRT_PRELOAD_TOP=0 ...]   ->  ray_tracer_amd/librt_amd_<name>.so
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
flags=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIPFLAGS))")
exec /opt/rocm/bin/hipcc $flags "$@" ray_tracer_amd/csrc/scene.cpp ray_tracer_amd/csrc/rt_device.hip -o ray_tracer_amd/librt_amd_$name.so
