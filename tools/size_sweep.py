#!/usr/bin/env python3
"""Which pipeline for how many paths? Sponza (or --scene), 1920 wide, 8 spp, one dispatch of `rows` rows (rank 0's rows of 1080 / rows
GPUs) per measurement: ms per dispatch by pipeline and by the smallest dispatch that is split into parts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heuristics_table as h
from ray_tracer_amd import engine
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
scene, cam = h.SCENES[name]()
r = engine.Renderer(0)
r.upload_scene(scene)
W, H = 1920, 1080
pc = cam(W, H, raysPerPixel=8, progressive=1, singleRender=0)
r.render(pc, W, H); r.render(pc, W, H)
print("| rows (paths) | fused | multi-kernel, one part | multi-kernel, parts from 128 k paths | auto |")
print("|---|---|---|---|---|")
for stride in (1, 2, 4, 8, 16):
    rows = H // stride
    out = []
    for tune in ({"pipeline": 1}, {"pipeline": 0, "lanes": 1}, {"pipeline": 0, "lanes": 3, "lanes_min_kslots": 128}, {"pipeline": -1, "lanes": 3, "lanes_min_kslots": 1024}):
        for k, v in tune.items():
            r.set_tuning(k, v)
        best = 1e9
        for rep in range(3):
            pc.frameCount = 0
            r.sync(); t = time.perf_counter(); r.render(pc, W, H, row0=0, rowStride=stride, sync=False); r.sync()
            best = min(best, (time.perf_counter() - t) * 1e3)
        out.append(best)
    print(f"| {rows} ({rows * W / 1e6:.2f} M) | " + " | ".join(f"{x:.2f}" for x in out) + f" ({['multi-kernel', 'fused'][r.last_pipeline()]}) |", flush=True)
