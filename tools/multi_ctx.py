#!/usr/bin/env python3
"""Does running G independent contexts (row-interleaved sub-tiles, one host thread and stream each) on ONE GPU
hide the per-round tails of a small tile? Emulates rank 0 of an 8-GPU run: rows 0, 8, 16, ... of 1080."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes

W, H, WORLD, SPP, STEPS = 1920, 1080, int(os.environ.get('WORLD', 8)), 8, 3
PIPE = int(os.environ.get('PIPE', -1))
scene, label = scenes.sponza(0)
for G in (1, 2, 3, 4, 6):
    rs = [engine.Renderer(0) for _ in range(G)]
    for r in rs:
        r.upload_scene(scene)
        r.set_tuning('pipeline', PIPE)
    pcs = [scenes.sponza_camera(W, H, raysPerPixel=SPP, progressive=1) for _ in range(G)]
    def work(g, steps):
        r, pc = rs[g], pcs[g]
        rows = range(g * WORLD, H, WORLD * G)   # rank 0's rows, dealt to G contexts
        for i in range(steps):
            pc.frameCount = i
            r.render(pc, W, H, row0=g * WORLD, rowStride=WORLD * G, nRows=len(rows))
    def run(steps):
        ts = [threading.Thread(target=work, args=(g, steps)) for g in range(G)]
        t = time.perf_counter()
        [x.start() for x in ts]; [x.join() for x in ts]
        return time.perf_counter() - t
    run(1)
    for r in rs: r.reset_counters()
    dt = run(STEPS)
    rays = sum(r.counters()["raysTraced"] for r in rs)
    print(f"G={G}: {dt/STEPS*1e3:7.1f} ms per step, {rays/dt/1e6:7.1f} M executed rays/s")
    for r in rs: r.close()
