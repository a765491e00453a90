#!/usr/bin/env python3
"""N2: the object hierarchy (DevScene::objTree) on and off, on tools/heuristics_table.py's 256 bunny instances.
usage: BUNNY_SCALE=0.12 tools/object_tree_ab.py <object_tree_min>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heuristics_table as h
from ray_tracer_amd import engine
tm = int(sys.argv[1])
r = engine.Renderer(0)
r.set_tuning("object_tree_min", tm)
s, cam = h.SCENES["bunnies256"]()
r.upload_scene(s)
pc = cam(1920, 1080, raysPerPixel=8, progressive=1, singleRender=0)
r.render(pc, 1920, 1080); r.render(pc, 1920, 1080); r.reset_counters()
out = []
for pipe in (0, 1):
    r.set_tuning("pipeline", pipe)
    r.render_frames(pc, 1920, 1080, 4)
    r.set_profiling(True)
    t = time.perf_counter(); r.render_frames(pc, 1920, 1080, 4); out.append((time.perf_counter() - t) / 4 * 1e3)
    if pipe == 0:
        busy = r.trace_busy_ms() / 4
    r.set_profiling(False)
if "--phase-stats" in sys.argv:
    r.set_tuning("pipeline", 0); r.set_tuning("phase_stats", 1); r.reset_counters(); r.render_frames(pc, 1920, 1080, 2); r.counters()
    r.set_tuning("phase_stats", 0)
c = r.counters()
print(f"bunny scale {h.BUNNY_SCALE} object_tree_min {tm}: multi-kernel {out[0]:.1f} ms (a k_trace_pw launch running for {busy:.1f} ms of it), fused {out[1]:.1f} ms per 8-spp 1080p frame (4 in flight); "
      f"{c['boxTests'] / c['raysTraced']:.0f} box tests per ray (reference count)", flush=True)
