#!/usr/bin/env python3
"""The multi-kernel pipeline in one part or in several overlapping parts ("lanes", "lane_grid_pct"): ms per 8-spp frame by scene.
usage: lanes_table.py [scene:width:height:frames ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heuristics_table as h
from ray_tracer_amd import engine
specs = [a for a in sys.argv[1:]] or ["sponza:1920:1080:10", "sponza_dragons:1920:1080:10", "sponza_dragons:3840:2160:2", "sponza_dragons_flat:3840:2160:2",
                                      "klein8:1920:1080:10", "bunny:1920:1080:10", "dragon:1920:1080:10"]
r = engine.Renderer(0)
settings = [(1, 100), (2, 60), (3, 50), (3, 40)]
print("| scene | frames per dispatch | kernel | " + " | ".join(f"lanes {l}, {p} %" for l, p in settings) + " |")
print("|---|---|---|" + "---|" * len(settings))
for spec in specs:
    name, W, H, frames = spec.split(":"); W, H, frames = int(W), int(H), int(frames)
    scene, cam = h.SCENES[name]()
    r.upload_scene(scene)
    pc = cam(W, H, raysPerPixel=8, progressive=1, singleRender=0)
    r.set_tuning("pipeline", 0)
    r.render_frames(pc, W, H, frames)
    out = []
    for lanes, pct in settings:
        r.set_tuning("lanes", lanes); r.set_tuning("lane_grid_pct", pct)
        best = 1e9
        for rep in range(2):
            pc.frameCount = 0
            r.sync(); t = time.perf_counter(); r.render_frames(pc, W, H, frames, sync=False); r.sync()
            best = min(best, (time.perf_counter() - t) / frames * 1e3)
        out.append(best)
    print(f"| {name} {W}x{H} | {frames} | {r.last_kernel()} | " + " | ".join(f"{x:.2f}" for x in out) + " |", flush=True)
