#!/bin/bash
# Round 3's wrong binary, reproduced: the library built from THIS tree with the AMDGPU backend pass si-opt-vgpr-liverange left on
# (everything else as __graft_entry__.HIPFLAGS) renders k_render_fused<24, true, false, false> wrong — C5 flattened (Sponza + 16
# flattened 871 414-triangle dragons), 3840x2160, every 16th row, 8 spp, fused pipeline: ~190 000 of 518 400 pixels with alpha != 1
# and ~187 000 whose colours differ from the multi-kernel pipeline's (which equals the oracle) — while the same sources with
# -mllvm -amdgpu-opt-vgpr-liverange=0 render it right. Source state: the commit that introduced this file (ROCm 7.2.0, clang 22).
#   on this container:  tools/miscompile_repro.sh build      (two libraries: ..._lr_on.so, ..._lr_off.so)
#   on the GPU box:     tools/miscompile_repro.sh run        (tools/alpha_check.py with each; also the instantiation-matrix test)
# The pass was found with tools/bisect_build.sh (hipcc -mllvm -opt-bisect-limit=N): N = 59 617 right, N = 59 618 wrong, and pass
# execution 59 618 of the device compile of rt_device.hip is "si-opt-vgpr-liverange on k_render_fused<24, true, false, false>".
cd "$(dirname "$0")/.." || exit 1
flags=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIPFLAGS))")
case "$1" in
  build)
    /opt/rocm/bin/hipcc ${flags/-mllvm -amdgpu-opt-vgpr-liverange=0/} ray_tracer_amd/csrc/scene.cpp ray_tracer_amd/csrc/rt_device.hip -o ray_tracer_amd/librt_amd_lr_on.so &
    /opt/rocm/bin/hipcc $flags ray_tracer_amd/csrc/scene.cpp ray_tracer_amd/csrc/rt_device.hip -o ray_tracer_amd/librt_amd_lr_off.so &
    wait ;;
  run)
    for v in lr_on lr_off; do
      echo "== $v"; QUICK=1 RT_AMD_LIB=$PWD/ray_tracer_amd/librt_amd_$v.so timeout -k 10 300 python3 tools/alpha_check.py 2>&1 | grep "^pipeline"
      RT_AMD_LIB=$PWD/ray_tracer_amd/librt_amd_$v.so timeout -k 10 600 python3 -m pytest tests/test_instantiations.py -x -q -m gpu 2>&1 | grep -v phase_stats | tail -3
    done ;;
  *) echo "usage: $0 build|run" ;;
esac
