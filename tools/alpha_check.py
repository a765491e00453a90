#!/usr/bin/env python3
"""Fused against multi-kernel on rows of a configuration (no oracle): pixels must be equal bit for bit, alpha must be 1."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import engine, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza_dragons_flat"); ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--spp", type=int, default=8); ap.add_argument("--stride", type=int, default=16); ap.add_argument("--tune", default="")
a = ap.parse_args()
scene, label = scenes.CONFIGS[a.scene]()
cam = scenes.sponza_camera if a.scene.startswith("sponza") else engine.push_constants
W, H = a.width, a.height
pc = cam(W, H, raysPerPixel=a.spp, progressive=1, singleRender=0)
tile = dict(row0=0, rowStride=a.stride, nRows=(H + a.stride - 1) // a.stride)
r = engine.Renderer(0)
r.upload_scene(scene)
imgs = []
for pipe, extra in ((0, {}), (1, {})) if os.environ.get("QUICK") else ((0, {}), (1, {}), (1, {"pixel_refill": 64}), (1, {"pixel_refill": 8}), (1, {"lds_stack": 16})):
    r.set_tuning("pipeline", pipe)
    for k, v in {"pixel_refill": 0, "lds_stack": 24, **extra}.items():
        r.set_tuning(k, v)
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("="); r.set_tuning(k, int(v))
    img = r.render(pc, W, H, **tile); imgs.append(img)
    print(f"pipeline {pipe} {extra} {r.last_kernel()}: alpha != 1 in {(img[..., 3] != 1).sum()} pixels; rgb differs from multi-kernel in {(img[..., :3].view(np.uint32) != imgs[0][..., :3].view(np.uint32)).any(axis=2).sum()}", flush=True)
