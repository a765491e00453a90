#!/bin/bash
# Which of the two AMDGPU backend passes that were switched off in rounds 2 and 3 is at fault? The three known-bad trees, re-made
# from this repository's history, each built three ways — A: both passes on, B: -amdgpu-opt-exec-mask-pre-ra=0, C:
# -amdgpu-opt-vgpr-liverange=0 — and run through the tests that failed at the time (ROCm 7.2.0, clang 22):
#   r2_heatmap   6eae51d with its work-around undone (heat-map instantiations of k_render_fused at five work-groups per CU again)
#   r2_ldstable  3b0e3f9 with the objects' LDS table in the CULL instantiations as well (the state that rendered wrong frames)
#   r3           02af0f7, the commit that found round 3's wrong k_render_fused<24, true, false, false>
# Result (profiles/README.md, "r03 one pass, not two"): A wrong on all three; B right on the first two, WRONG on the third; C right
# on all three. si-opt-vgpr-liverange is the common factor; the library is built with that one pass off.
#   here:            tools/pass_attribution.sh build     (trees under _passtrees/, nine libraries, ~6 minutes)
#   on the GPU box:  tools/pass_attribution.sh run       (~15 GPU-minutes)
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
F="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Iinclude -Iray_tracer_amd/csrc"
S="ray_tracer_amd/csrc/scene.cpp ray_tracer_amd/csrc/rt_device.hip"
K=ray_tracer_amd/csrc/rt_kernels.hip.h
case "$1" in
  build)
    for t in r2_heatmap:6eae51d r2_ldstable:3b0e3f9 r3:02af0f7; do
      d=$ROOT/_passtrees/${t%%:*}
      mkdir -p "$d" && git archive "${t##*:}" | tar -x -C "$d" || exit 1
      case ${t%%:*} in
        r2_heatmap) sed -i 's/__launch_bounds__(RT_BLOCK, PIX ? 4 : 5) void k_render_fused/__launch_bounds__(RT_BLOCK, 5) void k_render_fused/' "$d/$K" ;;
        r2_ldstable) sed -i -e 's/if (!CULL \&\& obj < RT_META_LDS)/if (obj < RT_META_LDS)/' -e '/^    if (CULL) return;$/d' -e 's/s_meta\[CULL ? 1 : RT_META_LDS\]/s_meta[RT_META_LDS]/' "$d/$K" ;;
      esac
      ( cd "$d" || exit 1
        /opt/rocm/bin/hipcc $F $S -o ray_tracer_amd/librt_amd_A.so &
        /opt/rocm/bin/hipcc $F -mllvm -amdgpu-opt-exec-mask-pre-ra=0 $S -o ray_tracer_amd/librt_amd_B.so &
        /opt/rocm/bin/hipcc $F -mllvm -amdgpu-opt-vgpr-liverange=0 $S -o ray_tracer_amd/librt_amd_C.so &
        wait
        make -s -C oracle && cp ray_tracer_amd/librt_amd_C.so ray_tracer_amd/librt_amd.so )
    done ;;
  run)
    for t in r2_heatmap r2_ldstable r3; do
      for v in A B C; do
        echo "== $t $v"
        ( cd "$ROOT/_passtrees/$t" || exit 1
          export RT_AMD_LIB=$PWD/ray_tracer_amd/librt_amd_$v.so
          if [ $t = r3 ]; then
            QUICK=1 timeout -k 10 300 python3 tools/alpha_check.py 2>&1 | grep "^pipeline"
            timeout -k 10 500 python3 -m pytest tests/test_instantiations.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v phase_stats | tail -2
          else
            timeout -k 10 400 python3 -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -14 | grep -E "FAILED|passed|failed"
          fi )
      done
    done ;;
  *) echo "usage: $0 build|run" ;;
esac
