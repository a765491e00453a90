#!/usr/bin/env python3
"""Would two halves of a dispatch on two streams overlap (one half's k_shade and launch tails under the other's k_trace_pw)?
G contexts on one GPU, each with its own stream, each rendering every G-th of the tile's rows with `frames` frames in flight,
all submitted asynchronously from one host thread, then synchronised. WORLD = 1: the bench frame; WORLD = 8: rank 0's rows of 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes
W, H, SPP = 1920, 1080, 8
WORLD = int(os.environ.get("WORLD", 1)); FRAMES = int(os.environ.get("FRAMES", 10)); BLOCKS = os.environ.get("BLOCKS", "")
scene, label = scenes.sponza(0)
for G in (1, 2, 3):
    for blocks in ([0] + [int(b) for b in BLOCKS.split(",") if b]):
        rs = [engine.Renderer(0) for _ in range(G)]
        for r in rs:
            r.upload_scene(scene)
            r.set_tuning("pipeline", 0)
            if blocks: r.set_tuning("blocks_per_cu", blocks)
        pcs = [scenes.sponza_camera(W, H, raysPerPixel=SPP, progressive=1) for _ in range(G)]
        def run():
            t = time.perf_counter()
            for g, r in enumerate(rs):
                pcs[g].frameCount = 0
                r.render_frames(pcs[g], W, H, FRAMES, row0=g * WORLD, rowStride=WORLD * G, sync=False)
            for r in rs: r.sync()
            return time.perf_counter() - t
        run(); run()
        dt = min(run(), run())
        print(f"WORLD={WORLD} frames={FRAMES} G={G} blocks_per_cu={blocks or 'auto'}: {dt / FRAMES * 1e3:7.2f} ms per step", flush=True)
        for r in rs: r.close()
