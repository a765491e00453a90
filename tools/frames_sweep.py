#!/usr/bin/env python3
"""ms per 8-spp 1080p frame by frames per dispatch and pipeline: frames_sweep.py scene [frames ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heuristics_table as h
from ray_tracer_amd import engine
name = sys.argv[1]
frames_list = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 6, 8, 10]
scene, cam = h.SCENES[name]()
r = engine.Renderer(0); r.upload_scene(scene)
W, H = 1920, 1080
pc = cam(W, H, raysPerPixel=8, progressive=1, singleRender=0)
r.render(pc, W, H); r.render(pc, W, H)
print(f"| {name}: frames per dispatch | fused | multi-kernel | auto |"); print("|---|---|---|---|")
for frames in frames_list:
    out = []
    for pipe in (1, 0, -1):
        r.set_tuning("pipeline", pipe)
        best = 1e9
        for rep in range(2):
            pc.frameCount = 0
            r.sync(); t = time.perf_counter()
            r.render_frames(pc, W, H, frames, sync=False) if frames > 1 else r.render(pc, W, H, sync=False)
            r.sync(); best = min(best, (time.perf_counter() - t) / frames * 1e3)
        out.append(best)
    print(f"| {frames} | {out[0]:.2f} | {out[1]:.2f} | {out[2]:.2f} ({['multi-kernel', 'fused'][r.last_pipeline()]}) |", flush=True)
